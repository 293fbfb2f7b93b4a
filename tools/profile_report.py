#!/usr/bin/env python3
"""Digest a gpurun_out/prof_<tag>/ directory (tools/gpu_profile.sh) into the summary that is
committed under profiles/: kernel-trace stats, PMC means per dispatch, HBM traffic per launch
with the FETCH_SIZE calibration measured on the same kernel."""
import csv, glob, json, os, sys, collections
d = sys.argv[1]

def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, sub, "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc

print("== kernel-trace stats (rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-parity)")
for f in glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        if "ro::" in row["Name"]:
            print("  %-70s calls %4s  avg %10.1f us  min %10.1f  max %10.1f" % (
                row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3,
                float(row["MaxNs"]) / 1e3))
# per-launch durations from the trace itself.  bench.py issues, in this order: the untimed launches (pre-warm + warm-up),
# the K timed steps, the soak behind them (clock_power.soak, ~1.5 s), and K launches inside ro_stft_time_resident
# (roofline.kernel_ms_posthoc).  The device needs ~20 launches after idle to reach its steady clocks, so the --stats
# average over ALL launches is not the steady-state figure; the timed steps' own mean is printed next to it.
untimed, timed = 55, 50
bj0 = os.path.join(d, "bench_trace.json")
if os.path.exists(bj0):
    for line in open(bj0):
        if line.startswith("{"):
            j0 = json.loads(line)
            timed = int(j0.get("steps", timed))
            untimed = int(j0.get("roofline", {}).get("untimed_launches_before_the_timed_region", j0.get("warmup", 30) + 25))
for f in glob.glob(os.path.join(d, "trace", "*", "*_kernel_trace.csv")):
    rows_ = list(csv.DictReader(open(f)))
    # the headline kernel only: the side legs of the default run (bolidozor: stft_kernel<Plan32768, ., 3>) are other kernels
    head = "stft32k_kernel" if any("stft32k_kernel" in r["Kernel_Name"] for r in rows_) else "stft_kernel"
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows_ if head in r["Kernel_Name"]]
    if len(dur) >= untimed + 2 * timed:
        mean = lambda x: sum(x) / max(len(x), 1)
        t = dur[untimed:untimed + timed]
        soak = dur[untimed + timed:len(dur) - timed]
        print("  stft_kernel launches in issue order, us: first 12 = %s" % [round(x) for x in dur[:12]])
        print("  %d launches: %d untimed (mean %.1f us), the %d TIMED steps: %.1f us, %d of the soak behind them: %.1f us, "
              "the last %d (ro_stft_time_resident): %.1f us" % (len(dur), untimed, mean(dur[:untimed]), timed, mean(t),
                                                                 len(soak), mean(soak), timed, mean(dur[-timed:])))
bj = os.path.join(d, "bench_trace.json")
if os.path.exists(bj):
    for line in open(bj):
        if line.startswith("{"):
            j = json.loads(line)
            print("  bench line under the profiler: value %.4g rows/s, kernel_ms %.4f (HIP events), frac %.4f" % (
                j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"]))

print("== FETCH_SIZE / WRITE_SIZE calibration on stft_kernel with overlap = 0 (every input byte read once)")
cal = {}
for c, known in (("FETCH_SIZE", 32768 * 8192 * 8), ("WRITE_SIZE", 32768 * 8192 * 4)):
    acc = counters("calib_" + c)
    for k, v in acc.items():
        if ("stft_kernel" in k or "stft32k_kernel" in k) and c in v:
            rep = sum(v[c]) / len(v[c]) * 1024.0
            cal[c] = known / rep
            print("  %s: reported %.4g B per launch, known %.4g B  -> multiply the counter by %.3f" % (c, rep, known, cal[c]))

print("== PMC means per dispatch, bench.py workload (R = 16384 rows, N = 32768, 75 % overlap)")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in glob.glob(os.path.join(d, "pmc_*")):
    if os.path.isdir(sub):
        for k, v in counters(os.path.basename(sub)).items():
            for c, vals in v.items():
                acc[k][c] += vals
for k in sorted(acc):
    if "ro::" not in k:
        continue
    print("  " + k[:90])
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("     %-26s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
    if ("stft_kernel" in k or "stft32k_kernel" in k) and "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
        f = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"]) * 1024.0
        w = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"]) * 1024.0
        fc, wc = f * cal.get("FETCH_SIZE", 1.0), w * cal.get("WRITE_SIZE", 1.0)
        alg = 196608.0 * 16384
        print("     -> HBM traffic per launch: fetch %.4g B (raw %.4g) + write %.4g B = %.4g B; algorithmic %.4g B; ratio %.3f"
              % (fc, f, wc, fc + wc, alg, (fc + wc) / alg))
        print("     TRAFFIC_BYTES_PER_LAUNCH %d" % int(fc + wc))
