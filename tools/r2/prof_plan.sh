#!/bin/bash
# GPU box: kernel-trace stats + SQ counter passes of one shape.  usage: prof_plan.sh TAG BINS OVERLAP ROWS [extra bench flags]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; B=$2; O=$3; R=$4; shift 4
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--bins $B --overlap $O --rows $R --no-cpu-baseline --no-strict --no-parity $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/trace.log
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/bench.py $ARGS --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_$N.log || echo "pmc $C failed"
done
{
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
PY
python3 $ROOT/tools/pmc_summary.py $OUT
cat $OUT/bench.json
} > $OUT/SUMMARY.txt 2>&1
cat $OUT/SUMMARY.txt
