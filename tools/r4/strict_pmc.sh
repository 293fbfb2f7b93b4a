#!/bin/bash
# GPU box: HBM bytes of the RO_PRECISION_F64 path at the C3 shape (2048 rows per step) for two sizes of its scratch chunk,
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (-DRO_DIAG=1 build in build/ab/libro_stft_diag.so).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/strict_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RO_STFT_LIB=$ROOT/build/ab/libro_stft_diag.so
for MB in 512 128; do
  export RO_F64_SCRATCH_MB=$MB
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/mb${MB}_$C -- python3 $ROOT/tools/r4/strict_sweep.py 3 > $OUT/mb${MB}_$C.log 2>&1 || echo "pmc $MB $C failed"
  done
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
print("# RO_PRECISION_F64, C3 shape, 2048 rows per step (8 steps per run: 5 warm-up + 3): HBM bytes per STEP from rocprofv3 --pmc")
print("# FETCH_SIZE x 1.994 (the gfx950 correction calibrated on stft32k_kernel, profiles/r04_stft_c3_summary.txt) and WRITE_SIZE, in KiB units of the counter")
for mb in (512, 128):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(out, "mb%d_%s" % (mb, c), "*", "*_counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == c and "ro::" in row["Kernel_Name"]:
                    acc[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
        per_step = 0.0
        for k, v in sorted(acc.items()):
            b = sum(v) * 1024.0 * (1.994 if c == "FETCH_SIZE" else 1.0) / 8.0       # 8 steps
            per_step += b
            print("  chunk %4d MiB  %-10s %-60s launches %4d  %.4g B per step" % (mb, c, k, len(v), b))
        tot[c] = per_step
    alg = 196608 * 2048
    print("  chunk %4d MiB: fetch %.4g + write %.4g = %.4g B per step = %.2f x algorithmic (%.4g); the two-trip model (44 B per point) is %.4g"
          % (mb, tot["FETCH_SIZE"], tot["WRITE_SIZE"], tot["FETCH_SIZE"] + tot["WRITE_SIZE"], (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / alg, alg, 44.0 * 32768 * 2048))
PY
