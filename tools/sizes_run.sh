#!/bin/bash
# GPU box: bench line summary for several (bins overlap) shapes.  usage: sizes_run.sh "4096 2048" "32768 24576" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for B in "$@"; do
  set -- $B
  python3 $ROOT/bench.py --bins $1 --overlap $2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bins=$1 overlap=$2', 'rows/s=%.4g' % d['value'], 'frac=%.3f' % d['roofline']['frac'], 'kernel_ms=%.4f' % d['roofline']['kernel_ms'], 'parity=%.2e' % d['parity']['max_err_rel_to_row_max'])"
done
