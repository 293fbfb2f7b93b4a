#!/bin/bash
# GPU box: counters under the one-kernel large form stft_kernel<Plan32768, ., 3> at Bolidozor's shape (bins 65536,
# overlap 49152, 8192 rows per launch; /root/reference/Bolidozor.json:45-46) and at 131072 / 75 %: FETCH_SIZE, WRITE_SIZE,
# L2 hits / misses and L2 read / write requests, one counter set per run.  usage: big_pmc.sh OUTDIR
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SHAPE in "65536 49152 8192" "131072 98304 4096"; do
  set -- $SHAPE
  B=$1; O=$2; R=$3
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_READ_sum TCC_WRITE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    T=$(echo $C | tr ' ' '_')
    rm -rf /tmp/bpmc_${B}_$T
    timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d /tmp/bpmc_${B}_$T -- python3 $ROOT/bench.py --bins $B --overlap $O --rows $R --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-streaming --no-large > /tmp/bpmc_${B}_$T.log 2>&1 < /dev/null || echo "pmc $B $C failed (counter set not available?)"
  done
done
python3 - $OUT <<'PY'
import csv, glob, collections, json, os, sys
out = sys.argv[1]
lines = []
for bins, overlap, rows in ((65536, 49152, 8192), (131072, 98304, 4096)):
    hop = bins - overlap
    alg = hop * 8 + bins * 4
    vals = {}
    for cset in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum", "TCC_READ_sum_TCC_WRITE_sum", "TCC_EA0_WRREQ_sum_TCC_EA0_WRREQ_64B_sum"):
        acc = collections.defaultdict(list)
        for f in glob.glob("/tmp/bpmc_%d_%s/*/*_counter_collection.csv" % (bins, cset)):
            for row in csv.DictReader(open(f)):
                if "stft_kernel" in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for c, v in acc.items():
            vals[c] = sum(v) / len(v)                      # per launch
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        lines.append("bins %d: counters missing (%s)" % (bins, sorted(vals)))
        continue
    fetch = vals["FETCH_SIZE"] * 1024.0 * 1.994
    write = vals["WRITE_SIZE"] * 1024.0
    per_row = (fetch + write) / rows
    rec = {"round": 5, "kernel": "stft_kernel<Plan32768, F32, 3> (csrc/ro_kernels.hip: one kernel, dec = %d workgroups per stream row)" % (bins // 32768),
           "workload": "bins %d, overlap %d, %d rows per launch" % (bins, overlap, rows),
           "fetch_size_raw_bytes": vals["FETCH_SIZE"] * 1024.0, "fetch_calibration": 1.994, "write_size_bytes": write,
           "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": float(alg) * rows,
           "ratio": per_row / alg,
           "tcc_hit_rate": (vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])) if "TCC_HIT_sum" in vals else None,
           "tcc_read_requests_per_row": vals.get("TCC_READ_sum", 0) / rows if "TCC_READ_sum" in vals else None,
           "tcc_write_requests_per_row": vals.get("TCC_WRITE_sum", 0) / rows if "TCC_WRITE_sum" in vals else None,
           "ea_write_requests_per_row": vals.get("TCC_EA0_WRREQ_sum", 0) / rows if "TCC_EA0_WRREQ_sum" in vals else None,
           "ea_write_requests_64B_per_row": vals.get("TCC_EA0_WRREQ_64B_sum", 0) / rows if "TCC_EA0_WRREQ_64B_sum" in vals else None,
           "method": "rocprofv3 --pmc, one counter set per run of bench.py --bins %d --overlap %d --rows %d (tools/r5/big_pmc.sh); FETCH_SIZE x 1.994 as calibrated on stft32k_kernel" % (bins, overlap, rows)}
    json.dump(rec, open(os.path.join(out, "traffic_%d.json" % bins), "w"), indent=1)
    lines.append("bins %6d overlap %6d: fetch %.4g + write %.4g B per launch = %.4g B per stream row = %.3f x algorithmic (%d); TCC hit %s; "
                 "L2 requests per row: read %s write %s; fabric write requests per row %s (64-byte: %s) for %d 128-byte lines of row"
                 % (bins, overlap, fetch, write, per_row, per_row / alg, alg,
                    "%.3f" % rec["tcc_hit_rate"] if rec["tcc_hit_rate"] is not None else "n/a",
                    "%.0f" % rec["tcc_read_requests_per_row"] if rec["tcc_read_requests_per_row"] is not None else "n/a",
                    "%.0f" % rec["tcc_write_requests_per_row"] if rec["tcc_write_requests_per_row"] is not None else "n/a",
                    "%.0f" % rec["ea_write_requests_per_row"] if rec["ea_write_requests_per_row"] is not None else "n/a",
                    "%.0f" % rec["ea_write_requests_64B_per_row"] if rec["ea_write_requests_64B_per_row"] is not None else "n/a",
                    bins * 4 // 128))
open(os.path.join(out, "big_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
