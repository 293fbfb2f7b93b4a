#!/bin/bash
# GPU box: sweep of the fused RO_PRECISION_F64 form's knobs against the two-launch form (-DRO_DIAG=1 build in build/ab).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export RO_STFT_LIB=$ROOT/build/ab/libro_stft_diag.so
P="python3 $ROOT/tools/r5/f64_sweep.py"
RO_F64_FUSED=0 $P 5 16384 || exit 1
for W in 2 1; do
  for RING in 4 6 8 12 16; do
    RO_F64_FUSED=1 RO_F64_WGS=$W RO_F64_RING_ROWS=$RING timeout -k 10 120 $P 5 16384 || exit 1
  done
done
