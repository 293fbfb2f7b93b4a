#!/bin/bash
# GPU box: does a smaller ring (less than an L2) change what the consumer's loads fetch across the fabric?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/f64f_ring
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in sc1 inv; do
 for RING in 2 3 4; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 120 rocprofv3 --pmc $C --output-format csv -d $OUT/${V}_${RING}_$C -- $ROOT/build/f64f_ld_$V 2048 $RING 1 3 > $OUT/${V}_${RING}_$C.log 2>&1 || { echo "pmc $V $RING $C failed"; exit 1; }
  done
 done
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
pts = 2048 * 32768.0
for v in ("sc1", "inv"):
  for ring in (2, 3, 4):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(os.path.join(out, "%s_%d_%s" % (v, ring, c), "*", "*_counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == c and "f64_fused" in row["Kernel_Name"]:
                    vals.append(float(row["Counter_Value"]))
        tot[c] = (sum(vals) / max(1, len(vals))) * 1024.0 * (1.994 if c == "FETCH_SIZE" else 1.0)
    print("loads %-4s ring %d rows per XCD (%.1f MiB): fetch %5.2f + write %5.2f B per point" % (v, ring, ring * 0.5, tot["FETCH_SIZE"] / pts, tot["WRITE_SIZE"] / pts))
PY
