#!/bin/bash
# GPU box: kernel trace, SQ counters and HBM-side bytes of the register-resident FP64 kernel at one shape (separate
# --pmc passes, no tracing alongside).  usage: tools/r6/f64r_pmc.sh TAG BINS OVERLAP ROWS -> gpurun_out/f64r_TAG/SUMMARY.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-c3}; BINS=${2:-32768}; OVL=${3:-24576}; ROWS=${4:-16384}
OUT=$ROOT/gpurun_out/f64r_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python3 $ROOT/tools/r6/f64r_run.py $BINS $OVL $ROWS 6"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $RUN > $OUT/trace.log 2>&1 || { echo "trace failed"; exit 1; }
i=0
# TRAFFIC_ONLY=1: the HBM-side byte counters alone (and the kernel trace)
if [ -n "$TRAFFIC_ONLY" ]; then SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"); else SETS=(
"SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC" \
         "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"); fi
for C in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- $RUN > $OUT/p$i.log 2>&1 || { echo "pmc pass $i failed"; exit 1; }
done
python3 - $OUT $BINS $OVL $ROWS <<'PY' > $OUT/SUMMARY.txt 2>&1
import csv, glob, os, sys, collections
out, bins, ovl, rows = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
launches = 6
print("# f64r_kernel, bins %d overlap %d, %d rows per launch, %d launches; rocprofv3, one counter set per run" % (bins, ovl, rows, launches))
for f in glob.glob(os.path.join(out, "trace", "*", "*_kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        if "ro::" in row["Name"]:
            print("kernel %s: calls %s avg %.1f us min %.1f max %.1f" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
            avg_ns = float(row["AverageNs"])
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "p*", "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if "f64r" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
v = {k: sum(x) / len(x) for k, x in acc.items()}
for k in sorted(v):
    print("%-26s %.6g per launch" % (k, v[k]))
D = {32768: 2, 65536: 4}.get(bins, 1)
M = max(bins // D, 4096)                          # (below 4096 bins: 4096 / bins rows share the M = 4096 workgroup)
waves_per_sub = M // 16 // 64
subrows = rows * D * bins // (D * M)
if "SQ_WAVE_CYCLES" in v:
    wc = v["SQ_WAVE_CYCLES"]
    print("shares of SQ_WAVE_CYCLES: WAIT_ANY %.3f  WAIT_INST_ANY %.3f  ACTIVE_INST_ANY %.3f  (ACTIVE VALU %.3f  LDS %.3f  SCA %.3f  VMEM %.3f)"
          % (v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc, v["SQ_ACTIVE_INST_ANY"] / wc, v["SQ_ACTIVE_INST_VALU"] / wc,
             v["SQ_ACTIVE_INST_LDS"] / wc, v["SQ_ACTIVE_INST_SCA"] / wc, v.get("SQ_ACTIVE_INST_VMEM", 0) / wc))
if "SQ_INSTS_VALU" in v:
    per = subrows * waves_per_sub
    print("per wave and sub-row: VALU %.0f  SALU %.0f  LDS %.0f  VMEM %.0f  SMEM %.0f  all %.0f"
          % (v["SQ_INSTS_VALU"] / per, v["SQ_INSTS_SALU"] / per, v["SQ_INSTS_LDS"] / per, v["SQ_INSTS_VMEM"] / per, v["SQ_INSTS_SMEM"] / per, v["SQ_INSTS"] / per))
if "SQ_LDS_IDX_ACTIVE" in v:
    print("LDS: bank conflict cycles / active cycles = %.4f" % (v["SQ_LDS_BANK_CONFLICT"] / max(1.0, v["SQ_LDS_IDX_ACTIVE"])))
if "GRBM_GUI_ACTIVE" in v:
    print("effective clock: %.0f MHz (GRBM_GUI_ACTIVE / 8 / kernel time)" % (v["GRBM_GUI_ACTIVE"] / 8 / avg_ns * 1e3))
alg = ((bins - ovl) * 8 + bins * 4) * rows
if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
    fetch = v["FETCH_SIZE"] * 1024 * 1.994       # gfx950: half of a wide read, x 1.994 as calibrated on stft32k_kernel
    write = v["WRITE_SIZE"] * 1024
    print("HBM side: fetch %.4g B (x 1.994) + write %.4g B = %.4g B per launch = %.3f x algorithmic (%.4g B); TCC hit %.3f"
          % (fetch, write, fetch + write, (fetch + write) / alg, alg, v.get("TCC_HIT_sum", 0) / max(1.0, v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0))))
    print("achieved: %.1f GB/s algorithmic = %.3f of 8 TB/s" % (alg / avg_ns, alg / avg_ns / 8000))
    import json
    hit = v.get("TCC_HIT_sum", 0) / max(1.0, v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0))
    json.dump({"round": 6, "kernel": "f64r_kernel (csrc/ro_f64reg.hip: RO_PRECISION_F64, the complex-double row in a CU's registers)",
               "workload": "bins %d, overlap %d, %d rows per launch" % (bins, ovl, rows),
               "fetch_size_raw_bytes": v["FETCH_SIZE"] * 1024, "fetch_calibration": 1.994, "write_size_bytes": write,
               "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": float(alg), "ratio": (fetch + write) / alg,
               "tcc_hit_rate": hit, "kernel_avg_us_under_the_profiler": avg_ns / 1e3,
               "method": "rocprofv3 --pmc, one counter set per run of tools/r6/f64r_run.py (tools/r6/f64r_pmc.sh); FETCH_SIZE x 1.994 as "
                         "calibrated on stft32k_kernel; with nt row stores (no line fills) this kernel's FETCH_SIZE x 1.994 is its input bytes within 1.1 %: profiles/r06_f64r_store_policy.txt"},
              open(os.path.join(out, "traffic.json"), "w"), indent=1)
PY
cat $OUT/SUMMARY.txt
