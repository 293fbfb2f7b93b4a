#!/bin/bash
# GPU box: A/B the variants in build/ab/ on ONE device, interleaved ROUNDS times (device-to-device spread is ~6 %,
# so only same-box numbers compare).  Also checks parity of each variant once.  usage: [AB_ARGS='--bins 4096 --overlap 2048'] ab_run.sh [ROUNDS] [names...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUNDS=${1:-3}; shift
NAMES=${@:-$(ls $ROOT/build/ab/ | sed 's/libro_stft_//;s/.so//')}
for N in $NAMES; do
  RO_STFT_LIB=$ROOT/build/ab/libro_stft_$N.so python3 $ROOT/bench.py $AB_ARGS --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$N', 'parity', d.get('parity'))"
done
for i in $(seq $ROUNDS); do
  for N in $NAMES; do
    RO_STFT_LIB=$ROOT/build/ab/libro_stft_$N.so python3 $ROOT/bench.py $AB_ARGS --steps 8 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('round $i', '$N', 'kernel_ms=%.4f' % d['roofline']['kernel_ms'], 'rows/s=%.4g' % d['value'])"
  done
done
