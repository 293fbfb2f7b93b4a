#!/bin/bash
# GPU box: bench.py kernel ms for several values of an env knob.  usage: sweep_env.sh VAR v1 v2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$1; shift
for V in "$@"; do
  env $VAR=$V python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$V', 'kernel_ms=%.4f' % d['roofline']['kernel_ms'], 'rows/s=%.4g' % d['value'])"
done
