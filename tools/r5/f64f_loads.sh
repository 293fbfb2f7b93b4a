#!/bin/bash
# GPU box: how the consumer's loads of the in-launch intermediate have to be flavoured for the XCD's L2 to serve them.
# The v1 fused kernel alone (tools/r5/f64f_debug.cpp), four builds that differ in those loads only, FETCH_SIZE / WRITE_SIZE per launch.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/f64f_loads
ROWS=${ROWS:-4096}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in sc1 nt inv sc0; do
  timeout -k 5 60 $ROOT/build/f64f_ld_$V $ROWS 6 1 3 | grep -E "^rows|spot" > $OUT/$V.run.txt 2>&1 || { echo "run $V failed"; exit 1; }
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 120 rocprofv3 --pmc $C --output-format csv -d $OUT/${V}_$C -- $ROOT/build/f64f_ld_$V $ROWS 6 1 3 > $OUT/${V}_$C.log 2>&1 || { echo "pmc $V $C failed"; exit 1; }
  done
done
python3 - $OUT $ROWS <<'PY'
import csv, glob, os, sys
out, rows = sys.argv[1], int(sys.argv[2])
pts = rows * 32768.0
print("# v1 fused FP64 kernel, bins 32768 / hop 8192, %d rows per launch, ring 6 rows per XCD, 1 workgroup per CU: bytes across the L2 <-> fabric boundary per point" % rows)
print("# (FETCH_SIZE x 1024 x 1.994: the gfx950 half-count of wide reads; WRITE_SIZE x 1024); algorithmic 6 B per point; the two-launch form moves 38.7")
for v, what in (("sc1", "buffer_load_dwordx4 sc1"), ("nt", "buffer_load_dwordx4 nt"), ("inv", "buffer_inv sc1, then plain buffer_load_dwordx4"), ("sc0", "buffer_load_dwordx4 sc0")):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(os.path.join(out, "%s_%s" % (v, c), "*", "*_counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == c and "f64_fused" in row["Kernel_Name"]:
                    vals.append(float(row["Counter_Value"]))
        tot[c] = (sum(vals) / max(1, len(vals))) * 1024.0 * (1.994 if c == "FETCH_SIZE" else 1.0)
    run = open(os.path.join(out, v + ".run.txt")).read().strip().replace("\n", " | ")
    print("%-48s fetch %5.2f + write %5.2f = %5.2f B per point   [%s]" % (what, tot["FETCH_SIZE"] / pts, tot["WRITE_SIZE"] / pts,
          (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / pts, run))
PY
