#!/bin/bash
# GPU box: the round's record run, part 1 -- the default bench line, the driver's form of it, the phase stamps of the FP64
# register kernel (needs build/ab/libro_stft_f64rstamps.so) and its full counter passes at C3 and C2.  Part 2 is
# tools/gpu_profile.sh r06 (the headline kernel's profile).  usage: tools/r6/round_run.sh OUTDIR
O=${1:-gpurun_out/r6_record}
mkdir -p $O
python3 bench.py > $O/bench_full.json 2> $O/bench_full.err || { echo "bench failed"; tail -5 $O/bench_full.err; exit 1; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverlike.json 2> /dev/null || { echo "driver-form bench failed"; exit 1; }
for s in "32768 24576 16384" "4096 2048 65536" "65536 49152 8192"; do
  RO_STFT_LIB=$PWD/build/ab/libro_stft_f64rstamps.so timeout -k 10 120 python3 tools/r6/f64r_stamps.py $s >> $O/stamps.txt 2>&1 || { echo "stamps failed"; exit 1; }
done
timeout -k 10 400 tools/r6/f64r_pmc.sh c3 32768 24576 16384 > $O/pmc_c3.log 2>&1 || { echo "pmc c3 failed"; exit 1; }
timeout -k 10 400 tools/r6/f64r_pmc.sh c2 4096 2048 65536 > $O/pmc_c2.log 2>&1 || { echo "pmc c2 failed"; exit 1; }
python3 - $O <<'PY'
import json, sys
O = sys.argv[1]
for f in ("bench_full", "bench_driverlike"):
    d = json.loads([l for l in open("%s/%s.json" % (O, f)) if l.startswith("{")][-1])
    print(f, "value %.4g" % d["value"], "ms/step %.4f" % d["ms_per_step"], "frac %.4f" % d["roofline"]["frac"], d["roofline"].get("limiter"))
    sp = d.get("strict_precision")
    if sp:
        print("   strict c3 %.4g (%.3f, traffic x %.3f)  c2 %.4g (%.3f)" % (sp["value"], sp["roofline"]["frac"], sp["roofline"].get("traffic_over_algorithmic") or 0,
              sp.get("c2", {}).get("value", 0), sp.get("c2", {}).get("roofline", {}).get("frac", 0)))
    if "bolidozor" in d:
        print("   bolidozor f32 %.4g (%.3f)  f64 %.4g (%.3f)" % (d["bolidozor"]["value"], d["bolidozor"]["roofline"]["frac"],
              d["bolidozor"].get("f64", {}).get("value", 0), d["bolidozor"].get("f64", {}).get("roofline", {}).get("frac", 0)))
    for k in ("streaming", "streaming_batch256"):
        if k in d:
            print("  ", k, "%.4g" % d[k]["value"], d[k].get("frac_of_pcie"))
PY
