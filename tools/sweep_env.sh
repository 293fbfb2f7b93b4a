#!/bin/bash
# GPU box: bench.py kernel ms for several values of a run-time knob of a -DRO_DIAG=1 build (RO_STFT_LIB).
# usage: [SWEEP_ARGS='--bins 16384 --overlap 12288'] sweep_env.sh VAR v1 v2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$1; shift
for V in "$@"; do
  env $VAR=$V python3 $ROOT/bench.py $SWEEP_ARGS --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-strict --no-streaming 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$SWEEP_ARGS $VAR=$V', 'kernel_ms=%.4f' % d['roofline']['kernel_ms'], 'rows/s=%.4g' % d['value'], 'frac=%.3f' % d['roofline']['frac'])"
done
