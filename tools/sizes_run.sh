#!/bin/bash
# GPU box: bench line summary for several shapes.  usage: sizes_run.sh "BINS OVERLAP [ROWS]" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for B in "$@"; do
  set -- $B
  R=${3:+--rows $3}
  python3 $ROOT/bench.py --bins $1 --overlap $2 $R --steps 8 --warmup 3 --no-cpu-baseline --no-strict --no-streaming 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bins=$1 overlap=$2 rows=%d' % d['config']['rows_per_step_per_gpu'], 'rows/s=%.4g' % d['value'], 'frac=%.3f' % d['roofline']['frac'], 'kernel_ms=%.4f' % d['roofline']['kernel_ms'], 'parity=%.2e' % d['parity']['max_err_rel_to_row_max'])"
done
