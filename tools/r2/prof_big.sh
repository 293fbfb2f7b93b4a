#!/bin/bash
# GPU box: kernel-trace stats of the large-transform path.  usage: [RO_BIG_FORM=.. RO_STFT_LIB=..] prof_big.sh BINS [ROWS]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
B=${1:-65536}
R=${2:-$((268435456/B))}
OUT=$ROOT/gpurun_out/prof_big_${B}_${RO_BIG_FORM:-default}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --bins $B --overlap $((B*3/4)) --rows $R --no-cpu-baseline --no-strict --no-parity --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/trace.log
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "ro::" in r["Name"]: print(r["Name"][:100], "calls", r["Calls"], "avg ns", r["AverageNs"], "pct", r["Percentage"])
PY
