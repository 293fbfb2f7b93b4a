#!/bin/bash
# GPU box: bytes across the L2 <-> fabric boundary of the RO_PRECISION_F64 path at the C3 shape, two-launch form against
# the one-launch form (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; -DRO_DIAG=1 build in build/ab).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/f64_pmc
ROWS=${ROWS:-4096}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RO_STFT_LIB=$ROOT/build/ab/libro_stft_diag.so
for FORM in 0 1; do
  export RO_F64_FUSED=$FORM
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    T=$(echo $C | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/f${FORM}_$T -- python3 $ROOT/tools/r5/f64_sweep.py 3 $ROWS > $OUT/f${FORM}_$T.log 2>&1 || { echo "pmc $FORM $C failed"; exit 1; }
  done
done
python3 - $OUT $ROWS <<'PY'
import csv, glob, os, sys, collections
out, rows = sys.argv[1], int(sys.argv[2])
launches = 6                                  # 3 warm-up + 3 timed steps per run
print("# RO_PRECISION_F64, C3 shape (bins 32768, overlap 24576), %d rows per step: counters per STEP, rocprofv3 --pmc, one counter set per run" % rows)
print("# FETCH_SIZE is in KiB and reads half of a wide streaming read on gfx950 (x 1.994 as calibrated on stft32k_kernel, profiles/r04_stft_c3_summary.txt); WRITE_SIZE in KiB")
for form in (0, 1):
    tot = {}
    for cset in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(out, "f%d_%s" % (form, cset), "*", "*_counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if "ro::" in row["Kernel_Name"]:
                    acc[row["Counter_Name"]][row["Kernel_Name"].split("(")[0][:70]].append(float(row["Counter_Value"]))
        for c, per_k in sorted(acc.items()):
            for k, v in sorted(per_k.items()):
                x = sum(v) / launches
                if c == "FETCH_SIZE": x *= 1024.0 * 1.994
                if c == "WRITE_SIZE": x *= 1024.0
                tot[c] = tot.get(c, 0.0) + x
                print("  form %d  %-12s %-70s launches %4d  %.4g per step" % (form, c, k, len(v), x))
    pts = rows * 32768.0
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        print("  form %d (%s): fetch %.4g B + write %.4g B per step = %.2f + %.2f = %.2f B per point (algorithmic 6.0); TCC hit rate %.3f"
              % (form, "one launch, intermediate in L2" if form else "two launches", tot["FETCH_SIZE"], tot["WRITE_SIZE"],
                 tot["FETCH_SIZE"] / pts, tot["WRITE_SIZE"] / pts, (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / pts,
                 tot.get("TCC_HIT_sum", 0) / max(1.0, tot.get("TCC_HIT_sum", 0) + tot.get("TCC_MISS_sum", 0))))
PY
