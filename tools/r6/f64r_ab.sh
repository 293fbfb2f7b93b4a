#!/bin/bash
# GPU box: A/B of builds of the library on the FP64 register kernel's rates (tools/r6/f64r_check.py --rate-only), the
# variants interleaved ROUNDS times on one device.  usage: tools/r6/f64r_ab.sh ROUNDS SIZES name1 name2 ...   (names of
# build/ab/libro_stft_<name>.so; "product" = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=$1; SIZES=$2; shift 2
for r in $(seq $ROUNDS); do
  for n in "$@"; do
    if [ "$n" = product ]; then unset RO_STFT_LIB; else export RO_STFT_LIB=$ROOT/build/ab/libro_stft_$n.so; fi
    timeout -k 10 200 python3 $ROOT/tools/r6/f64r_check.py --rate-only --sizes $SIZES 2>&1 | grep "rows/s" | sed "s/^/round $r $n: /" || exit 1
  done
done
