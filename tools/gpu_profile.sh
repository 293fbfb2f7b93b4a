#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py.
# usage: tools/gpu_profile.sh <tag>     -> gpurun_out/prof_<tag>/
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-parity --no-streaming"      # the default run: 25 pre-warm + 30 warm-up + 50 timed steps
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.log
echo "trace done"
SHORT="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-streaming --no-strict --no-large"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- $SHORT > /dev/null 2> $OUT/pmc_$N.log || echo "pmc $C failed"
  echo "pmc $C done"
done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/calib_$C -- python3 $ROOT/tools/calib_run.py > $OUT/calib_$C.log 2>&1 || echo "calib $C failed"
  echo "calib $C done"
done
python3 $ROOT/tools/profile_report.py $OUT > $OUT/SUMMARY.txt 2>&1
cat $OUT/SUMMARY.txt
# the strict-precision (RO_PRECISION_F64) path on its own: which kernels it runs and for how long
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/strict -- python3 $ROOT/tools/r3/strict_run.py > $OUT/strict.log 2>&1 || echo "strict trace failed"
python3 - <<PY >> $OUT/SUMMARY.txt
import csv, glob
print("== kernel-trace stats of the strict-precision path (rocprofv3 --kernel-trace --stats -- python3 tools/r3/strict_run.py: 2048 rows x 20 launches, N = 32768, 75 % overlap)")
for f in glob.glob("$OUT/strict/*/*_kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        if "ro::" in row["Name"]:
            print("  %-70s calls %4s  avg %10.1f us  min %10.1f  max %10.1f" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
tail -12 $OUT/SUMMARY.txt
