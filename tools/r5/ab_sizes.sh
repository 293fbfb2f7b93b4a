#!/bin/bash
# GPU box: the variants in build/ab/ at several shapes, interleaved.  usage: ab_sizes.sh ROUNDS "BINS OVERLAP ROWS" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=$1; shift
NAMES=$(ls $ROOT/build/ab/ | sed 's/libro_stft_//;s/.so//')
for B in "$@"; do
  set -- $B
  for i in $(seq $ROUNDS); do
    for N in $NAMES; do
      RO_STFT_LIB=$ROOT/build/ab/libro_stft_$N.so python3 $ROOT/bench.py --bins $1 --overlap $2 --rows $3 --steps 8 --warmup 3 --no-cpu-baseline --no-strict --no-streaming --no-large --no-legs --no-parity 2>/dev/null | tail -1 | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bins=$1 overlap=$2 round $i $N', 'rows/s=%.4g' % d['value'], 'frac=%.3f' % d['roofline']['frac'])"
    done
  done
done
