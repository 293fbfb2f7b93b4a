#!/bin/bash
# GPU box: the two-launch RO_PRECISION_F64 form, one workgroup per tile (f64_pair_kernel) against the persistent software-pipelined
# kernel (ro_f64stream.hip), interleaved on one device (-DRO_DIAG=1 build: RO_F64_STREAM from the environment).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export RO_STFT_LIB=$ROOT/build/ab/libro_stft_diag.so RO_F64_FUSED=0
P="python3 $ROOT/tools/r5/f64_sweep.py"
for i in 1 2 3; do
  RO_F64_STREAM=0 timeout -k 10 120 $P 10 16384 || exit 1
  RO_F64_STREAM=1 timeout -k 10 120 $P 10 16384 || exit 1
done
for SHAPE in "8192 6144" "65536 49152" "4096 2048" "1048576 0"; do
  set -- $SHAPE
  R=$((268435456 / $1)); [ $R -gt 16384 ] && R=16384
  RO_F64_STREAM=0 timeout -k 10 120 $P 5 $R $1 $2 || exit 1
  RO_F64_STREAM=1 timeout -k 10 120 $P 5 $R $1 $2 || exit 1
done
