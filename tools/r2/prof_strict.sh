#!/bin/bash
# GPU box: kernel-trace stats of the RO_PRECISION_F64 leg of bench.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_strict
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $OUT/bench.json 2> $OUT/trace.log
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "f64" in r["Name"]: print(r["Name"][:100], "calls", r["Calls"], "avg ns", r["AverageNs"])
PY
