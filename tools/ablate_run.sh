#!/bin/bash
# on the GPU box: time every diagnostic variant with bench.py (kernel ms from HIP events)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for L in $ROOT/build/ablate/libro_stft_*.so; do
  RO_STFT_LIB=$L python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $L)', 'kernel_ms=%.4f' % d['roofline']['kernel_ms'], 'rows/s=%.3g' % d['value'])"
done
