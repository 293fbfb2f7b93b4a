#!/bin/bash
# GPU box: HBM-side bytes of the four-step form at Ionozor's shape (bins 524288, overlap 262144, 1024 rows per step):
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes.  usage: four_pmc.sh OUTDIR
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in ${PMCS:-FETCH_SIZE WRITE_SIZE}; do
  rm -rf /tmp/fpmc_$C
  rocprofv3 --pmc $C --output-format csv -d /tmp/fpmc_$C -- python3 $ROOT/bench.py --bins 524288 --overlap 262144 --rows 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-streaming > /tmp/fpmc_$C.log 2>&1 < /dev/null || echo "pmc $C failed"
done
python3 - > $OUT/four_pmc.txt <<'PY'
import csv, glob, collections
print("# bins 524288, overlap 262144, 1024 rows per launch of the transform: bytes per STREAM ROW from rocprofv3 --pmc, per kernel")
print("# FETCH_SIZE x 1.994 (the gfx950 correction calibrated on stft32k_kernel, profiles/r04_stft_c3_summary.txt) and WRITE_SIZE, KiB units")
tot = 0.0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob("/tmp/fpmc_%s/*/*_counter_collection.csv" % c):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c and ("four" in row["Kernel_Name"] or "scan" in row["Kernel_Name"]):
                acc[row["Kernel_Name"][:48]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        rows_per_launch = 256.0 if "four" in k else 1024.0
        b = sum(v) / len(v) * 1024.0 * (1.994 if c == "FETCH_SIZE" else 1.0) / rows_per_launch
        tot += b
        print("  %-10s %-48s launches %4d  %.4g MiB per stream row" % (c, k, len(v), b / 2**20))
alg = 262144 * 8 + 524288 * 4
print("  total %.4g MiB per stream row = %.2f x algorithmic (%.4g MiB); the model 8 hop + 20 bins is %.4g MiB" % (tot / 2**20, tot / alg, alg / 2**20, (262144 * 8 + 20 * 524288) / 2**20))
PY
cat $OUT/four_pmc.txt
