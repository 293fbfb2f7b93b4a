#!/bin/bash
# GPU box: the round's record run -- the default bench line, the driver's form of it, the C5 stream on one rank and on
# two ranks started by bench.py itself (gloo-host rehearsal: both ranks on this one GPU), then the rocprofv3 profile.
O=${1:-gpurun_out/r4j}
mkdir -p $O
python3 bench.py > $O/bench_full.json 2> $O/bench_full.err &&
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverlike.json 2>/dev/null &&
python3 bench.py --workload c5 --c5-seconds 600 --no-cpu-baseline --no-strict > $O/c5_1rank.json 2> $O/c5_1rank.err &&
timeout -k 10 300 python3 bench.py --gpus 2 --comm gloo-host --workload c5 --c5-seconds 600 --no-cpu-baseline > $O/c5_2rank_launcher.json 2> $O/c5_2rank_launcher.err
echo "launcher rc=$?"
timeout -k 10 700 bash tools/gpu_profile.sh r04 > $O/profile.log 2>&1
echo "profile rc=$?"
python3 - $O <<'PY'
import json, sys
O = sys.argv[1]
for f in ("bench_full", "bench_driverlike", "c5_1rank", "c5_2rank_launcher"):
    try:
        d = json.loads(open("%s/%s.json" % (O, f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no line", e)
        continue
    print(f, "n_gpus", d["n_gpus"], "value %.4g" % d["value"], "frac %.4f" % d["roofline"]["frac"],
          d["config"].get("c5_hash_of_stitched_band_and_records"), (d.get("clock_power") or {}).get("timed_region"))
    for k in ("streaming", "streaming_batch256"):
        if k in d:
            print("  ", k, {kk: d[k][kk] for kk in ("value", "frac_of_pcie", "pcie_h2d_GBs", "pcie_d2h_GBs", "pcie_bound_rows_per_s",
                                                   "ms_per_process_call_mean", "rows_by_dma_into_the_row_ring", "stages") if kk in d[k]})
    if "strict_precision" in d:
        print("   strict", d["strict_precision"]["value"])
PY
tail -5 $O/c5_2rank_launcher.err | cut -c1-200
