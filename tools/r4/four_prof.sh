#!/bin/bash
# GPU box: per-kernel times of the four-step form at Ionozor's shape, for several scratch chunk sizes
# (diagnostic library: RO_FOUR_SCRATCH_MB).  usage: four_prof.sh OUTDIR [MB ...]
O=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for MB in "$@"; do
  export RO_FOUR_SCRATCH_MB=$MB RO_BIG_FORM=${FORM:-four} RO_STFT_LIB=$R/build/ab/libro_stft_${LIBV:-diag}.so
  rm -rf /tmp/fp_$MB
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_$MB -- python3 $R/bench.py --bins ${BINS:-524288} --overlap ${OVERLAP:-262144} --rows ${ROWS:-1024} --steps 8 --warmup 3 --no-cpu-baseline --no-strict --no-streaming > /tmp/fp_$MB.log 2>&1 < /dev/null
  echo "== RO_FOUR_SCRATCH_MB=$MB RO_BIG_FORM=$RO_BIG_FORM lib=${LIBV:-diag}" >> $R/$O/four_prof.txt
  grep '^{' /tmp/fp_$MB.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('rows/s=%.4g frac=%.3f kernel_ms=%.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']))" >> $R/$O/four_prof.txt
  F=$(find /tmp/fp_$MB -name "*kernel_stats.csv" | head -1)
  if [ -n "$F" ]; then head -4 "$F" | cut -c1-160 >> $R/$O/four_prof.txt; else echo "no kernel_stats.csv" >> $R/$O/four_prof.txt; fi
done
