#!/bin/bash
# GPU box: the one-launch RO_PRECISION_F64 form against the two-launch form at the C3 shape, 16384 rows per step
# (-DRO_DIAG=1 build in build/ab: the knobs come from the environment).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export RO_STFT_LIB=$ROOT/build/ab/libro_stft_diag.so
P="python3 $ROOT/tools/r5/f64_sweep.py"
RO_F64_FUSED=0 timeout -k 10 120 $P 5 16384 || exit 1
for W in 1 2; do
  for RING in 2 3 4 6 8; do
    RO_F64_FUSED=1 RO_F64_WGS=$W RO_F64_RING_ROWS=$RING timeout -k 10 120 $P 5 16384 || exit 1
  done
done
