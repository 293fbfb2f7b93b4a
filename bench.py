#!/usr/bin/env python3
"""bench.py -- spectra/s of the MI355X STFT / waterfall / bolid-scan hot path.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as
      python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the env), RCCL = backend "nccl".  Started from a plain
  shell (`python3 bench.py --gpus N`, no WORLD_SIZE in the env) it starts that very job as a child process before
  it has imported torch or made any HIP call, and exits with the child's code (launch_ranks).

Workload (BASELINE.json configs[2..3], "C3/C4"): synthetic IQ resident in HBM, 48 kHz,
FFT bins = 32768, overlap = 24576 (75 %), Nuttall window, R = 16384 rows per step per GPU
(T = 32768 + 8192*(R-1) samples = 1.07 GB float32 I/Q in, 2.15 GB of float32 magnitude
rows out), sigma=1 Gaussian noise + CW carrier 30 sigma at +10.6 kHz; every step also runs
BolidRecorder's per-row noise/peak/average scan with radio-observer.json's bands.

A "step" = one pass of the hot path over that batch: the fused window->FFT->|X|->shift
kernel with the band scan in its epilogue, inputs already in HBM.  With N > 1 every rank owns
one time chunk of R rows (weak scaling, no collective inside the transform) and each step ends
with the RCCL all-gather of that step's waterfall band tile + scan records -- the stitch
north_star names (--gather root: to rank 0 only, the one consumer the reference has;
--gather none: compute only; a run reports all three) -- overlapped with the next step's
compute on a side stream.  --workload c5 is BASELINE config 5 itself: ONE fixed 8-hour stream
(168 747 rows) split over the ranks by ro_shard_rows (strong scaling), every rank generating
its own slice from a counter-based generator, rank 0 hashing the stitched band + records so
that an N-rank run can be compared with the 1-rank run bit for bit.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BINS, OVERLAP, FS = 32768, 24576, 48000
HOP = BINS - OVERLAP
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
ALG_BYTES_PER_ROW = HOP * 8 + BINS * 4     # hop*8 in + bins*4 out = 196608 B
# radio-observer.json:62-87
JSON_SNAPSHOT = (10100.0, 11000.0)
JSON_BOLID = dict(min_detect=10300.0, max_detect=10900.0, min_noise=9000.0, max_noise=9600.0,
                  avg_freq_range=40.0)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=30)
    p.add_argument("--prewarm", type=int, default=25,
                   help="untimed launches before the W warm-up steps: the device needs ~20 launches after idle to "
                        "settle its clocks (first launches run 1.06, 1.24, 1.45, 1.35 ... ms), whatever W is")
    p.add_argument("--rows", type=int, default=16384, help="rows per step per GPU")
    p.add_argument("--bins", type=int, default=32768, help="FFT size (default = the headline C3/C4 workload)")
    p.add_argument("--overlap", type=int, default=None, help="overlap in samples (default 75 %% of bins)")
    p.add_argument("--window", choices=["nuttall", "hann"], default="nuttall",
                   help="nuttall = what the reference runs (src/FFTBackend.cpp:165-184); hann = BASELINE's C2 wording")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-parity", action="store_true")
    p.add_argument("--comm", choices=["nccl", "gloo-host"], default="nccl",
                   help="gloo-host: rehearsal mode for 1-GPU boxes -- every rank uses cuda:0 and the gather is "
                        "staged through host memory over gloo (exercises the N>1 control flow, not xGMI)")
    p.add_argument("--pattern", choices=["nccl", "direct"], default="nccl",
                   help="how --gather all moves the rows: one ncclAllGather (whatever algorithm RCCL picks) or the DIRECT "
                        "exchange SURVEY 8(e) prescribes for xGMI's point-to-point links -- every rank sends its block to each "
                        "peer and receives each peer's block at its stitched place, one group of sends / receives "
                        "(ro_allgather_rows_direct with --exchange capi); a run with N > 1 reports both legs")
    p.add_argument("--gather", choices=["all", "root", "none"], default="all",
                   help="N > 1: what ends a step inside the timed region of `value`.  all (default) = the RCCL all-gather "
                        "of band tile + scan records north_star and BASELINE's config 5 name (every rank stitches); root = "
                        "gathered to rank 0 only, the one process that stitches in the reference "
                        "(src/WaterfallBackend.cpp:141-211, src/BolidRecorder.cpp:171-273): 7 direct transfers of 40 MB "
                        "over 7 xGMI links, where a ring all-gather moves 283 MB through every link; none = compute "
                        "only.  The other two are measured after the timed region (config.exchange_legs_rows_per_s).")
    p.add_argument("--exchange", choices=["torch", "capi"], default="torch",
                   help="who moves the bytes: torch.distributed (default) or the product's own C-ABI exchange "
                        "(ro_allgather_rows / ro_gather_rows on an ncclComm_t created before the timed loop)")
    p.add_argument("--workload", choices=["c3", "c5"], default="c3",
                   help="c3 (default): R rows per GPU and step, weak scaling; c5: BASELINE config 5, one fixed stream of "
                        "--c5-seconds split over the ranks by ro_shard_rows, strong scaling, hash of the stitched result")
    p.add_argument("--c5-seconds", type=float, default=8 * 3600.0, help="length of the c5 stream (default 8 h)")
    p.add_argument("--no-streaming", action="store_true", help="skip the drop-in (Backend::process) streaming leg")
    p.add_argument("--stream-seconds", type=float, default=2.5, help="timed length of the streaming leg")
    p.add_argument("--no-legs", action="store_true",
                   help="N > 1: skip the extra untimed-for-`value` runs with the other --gather modes (config.exchange_legs_rows_per_s)")
    p.add_argument("--no-strict", action="store_true", help="skip the RO_PRECISION_F64 side measurement")
    p.add_argument("--no-large", action="store_true", help="skip the side measurements at the station configs' shapes (bins 524288, 65536)")
    p.add_argument("--soak-seconds", type=float, default=1.5,
                   help="N = 1: behind the timed region, keep launching the same step for this long while the clock / "
                        "power sampler runs (clock_power.soak): the package power the hwmon file reports is an average "
                        "over about a second, longer than the timed region of the default run (0 = skip)")
    p.add_argument("--pmc-traffic", type=float, default=None,
                   help="HBM bytes per launch from a separate rocprofv3 --pmc run (default: profiles/*_traffic.json)")
    return p.parse_args()


def make_bands(ro):
    f2b = lambda f: ro.frequency_to_bin(BINS, FS, f)
    lo_d, hi_d = sorted((f2b(JSON_BOLID["min_detect"]), f2b(JSON_BOLID["max_detect"])))
    lo_n, hi_n = sorted((f2b(JSON_BOLID["min_noise"]), f2b(JSON_BOLID["max_noise"])))
    avg = max(f2b(JSON_BOLID["avg_freq_range"]) - f2b(0.0), 1)
    return ro.Bands(low_noise=lo_n, noise_width=hi_n - lo_n, low_detect=lo_d, detect_width=hi_d - lo_d,
                    avg_bins=avg)


def synth_iq(torch, samples, seed, device):
    """sigma=1 complex noise + 30 sigma carrier at +10.6 kHz, float32 interleaved [samples, 2]."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    iq = torch.randn((samples, 2), generator=g, device=device, dtype=torch.float32)
    # carrier phase computed modulo the period so float32 stays exact enough at 1e8 samples
    t = torch.arange(samples, device=device, dtype=torch.int64)
    period_num = 10600 * t % FS                       # phase = 2 pi * (10600 t mod 48000) / 48000
    ph = period_num.to(torch.float64) * (2.0 * np.pi / FS)
    iq[:, 0] += (30.0 * torch.cos(ph)).to(torch.float32)
    iq[:, 1] += (30.0 * torch.sin(ph)).to(torch.float32)
    del t, period_num, ph
    return iq


def oracle_module(lib_path=None):
    """the oracle's ctypes loader; lib_path selects another build of the same source (the -O0 twin)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    if lib_path is None:
        import ro_oracle as O
        O.lib()
        return O
    import importlib.util
    spec = importlib.util.spec_from_file_location("ro_oracle_alt", os.path.join(ROOT, "oracle", "ro_oracle.py"))
    O = importlib.util.module_from_spec(spec)
    os.environ["RO_ORACLE_LIB"] = lib_path
    try:
        spec.loader.exec_module(O)
        O.lib()
    finally:
        os.environ.pop("RO_ORACLE_LIB", None)
    return O


def cpu_baseline(iq_host, w, bands, budget_s, O=None):
    """Oracle (oracle/ro_oracle.c, one thread) on a bounded prefix of the same input; the FP64 transform runs on
    whatever engine the oracle has been switched to (O.fft_engine())."""
    O = O or oracle_module()
    max_rows = O.row_count(iq_host.shape[0], BINS, OVERLAP)
    O.fft_prepare(BINS)
    done, chunk = 0, 64
    t0 = time.perf_counter()
    while done < max_rows and time.perf_counter() - t0 < budget_s:
        n = min(chunk, max_rows - done)
        rows = O.stft(iq_host, BINS, OVERLAP, w=w, first_row=done, max_rows=n)
        O.scan_rows(rows, bands.low_noise, bands.noise_width, bands.low_detect, bands.detect_width,
                    bands.avg_bins)
        done += n
    dt = time.perf_counter() - t0
    return done, dt


def cpu_baseline_threads(iq_host, w, bands, budget_s, threads):
    """The same oracle on `threads` host threads at once (the C calls release the GIL), every thread walking its own
    interleaved set of 64-row chunks of the same input until the budget is spent."""
    import threading
    O = oracle_module()
    O.fft_prepare(BINS)                                      # plans / tables before the threads share them
    max_rows = O.row_count(iq_host.shape[0], BINS, OVERLAP)
    chunk = 64
    done = [0] * threads
    t0 = time.perf_counter()

    def work(t):
        first = t * chunk
        while time.perf_counter() - t0 < budget_s:
            if first >= max_rows:
                first = t * chunk                            # the sample is bounded; walk it again
            n = min(chunk, max_rows - first)
            rows = O.stft(iq_host, BINS, OVERLAP, w=w, first_row=first, max_rows=n)
            O.scan_rows(rows, bands.low_noise, bands.noise_width, bands.low_detect, bands.detect_width,
                        bands.avg_bins)
            done[t] += n
            first += threads * chunk

    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return sum(done), time.perf_counter() - t0


def host_cores():
    """cores this process may use: the affinity mask, cut down to the cgroup CPU quota when there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            if f.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def pmc_traffic(rows):
    """HBM bytes per launch measured with rocprofv3 PMC counters in their own passes (the counters
    cannot be read from inside this process); the newest profiles/rNN_traffic.json, scaled to `rows`."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        try:
            best = json.load(open(f))
        except Exception:
            pass
    if not best:
        return None
    if (BINS, OVERLAP) != (32768, 24576):
        return None                                          # the PMC passes were taken on the headline shape
    per_row = best["traffic_bytes_per_launch"] / (best["algorithmic_bytes_per_launch"] / ALG_BYTES_PER_ROW)
    return per_row * rows


def pcie_copy_rates(torch, dev, mib=256):
    """pinned host <-> device copy rates of this box, both directions at once on two streams (the streaming path
    uploads one batch while it downloads another): GB/s host-to-device, device-to-host"""
    n = mib << 20
    h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
    h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
    d_in = torch.empty(n, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for timed in (False, True):
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(s1):
            ev[0].record(s1)
            for _ in range(4):
                d_in.copy_(h_in, non_blocking=True)
            ev[1].record(s1)
        with torch.cuda.stream(s2):
            ev[2].record(s2)
            for _ in range(4):
                h_out.copy_(d_out, non_blocking=True)
            ev[3].record(s2)
        torch.cuda.synchronize(dev)
    both = (4 * n / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9, 4 * n / (ev[2].elapsed_time(ev[3]) * 1e-3) / 1e9)
    # ... and each direction on its own: what the link gives a direction at best.  The path's uploads are half the size of
    # its downloads and neither runs all the time, so the one-direction rates are the bound that can never be beaten
    # (the both-at-once rates were: a box's simultaneous 256 MiB copies can be slower than the path's own traffic)
    # (best of four timed rounds per direction: a direction's rate is bimodal on this pool -- 30 or 57 GB/s device to host
    # from one run to the next -- and a bound is the best the link does)
    alone = []
    for src, dst, st in ((h_in, d_in, s1), (d_out, h_out, s2)):
        best = 0.0
        for trial in range(5):
            torch.cuda.synchronize(dev)
            with torch.cuda.stream(st):
                ev[0].record(st)
                for _ in range(4):
                    dst.copy_(src, non_blocking=True)
                ev[1].record(st)
            torch.cuda.synchronize(dev)
            if trial > 0:
                best = max(best, 4 * n / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9)
        # ... and the same direction as three streams' worth of copies at once, as the path issues them (a slot's stream
        # each): on a box where ONE stream's device-to-host copies run at 30 GB/s the path itself moved 40 -- a copy
        # engine's rate, not the link's -- and a bound has to be what the link does at best
        streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
        third = n // 3
        t0, t1 = torch.cuda.Event(enable_timing=True), [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for trial in range(4):
            torch.cuda.synchronize(dev)
            t0.record(torch.cuda.current_stream(dev))
            for k, sk in enumerate(streams):
                sk.wait_event(t0)
                with torch.cuda.stream(sk):
                    for _ in range(4):
                        dst[k * third:(k + 1) * third].copy_(src[k * third:(k + 1) * third], non_blocking=True)
                    t1[k].record(sk)
            torch.cuda.synchronize(dev)
            if trial > 0:
                best = max(best, 4 * 3 * third / (max(t0.elapsed_time(e) for e in t1) * 1e-3) / 1e9)
        alone.append(best)
    return both[0], both[1], alone[0], alone[1]


def strict_leg(torch, ro, dev, local_rank, parity, bins, overlap, rows, fs, bands, window, steps, iq=None, rows_buf=None,
               recs_buf=None):
    """RO_PRECISION_F64 on one shape, inputs resident: rows/s and the fraction of the HBM peak its algorithmic bytes make,
    every kernel of the mode included (the transform's; the band scan's when `bands`), HIP events by ro_stft_time_resident."""
    hop = bins - overlap
    samples = bins + hop * (rows - 1)
    own = iq is None
    if own:
        iq = synth_iq(torch, samples, 0xC2 if bins == 4096 else 0x64, dev)
        rows_buf = torch.empty((rows, bins), dtype=torch.float32, device=dev)
    if bands is not None and recs_buf is None:
        recs_buf = torch.zeros((rows, 3), dtype=torch.float32, device=dev)
    sptr = torch.cuda.current_stream(dev).cuda_stream
    alg = hop * 8 + bins * 4
    reg = 256 <= bins <= 65536
    with ro.Stft(bins=bins, overlap=overlap, sample_rate=fs, device=local_rank, bands=bands, window=window,
                 precision=ro.RO_PRECISION_F64) as st64:
        n64 = steps + 2
        ms64, _, _ = st64.time_resident(iq, ro.RO_IQ_F32, samples, 0, rows, rows_buf, n64,
                                        d_records=recs_buf if bands is not None else None, stream=sptr)
        torch.cuda.synchronize(dev)
        ms_strict = float(np.mean(ms64[2:]))                  # (the first two launches warm the tables and the clocks)
        entry = {"mode": "RO_PRECISION_F64 (double window multiply, double transform, double sqrt, one narrowing: the "
                         "reference's arithmetic type)%s" % ("; the complex-double row stays in a CU's registers, no HBM scratch"
                                                            if reg else "; passes through HBM scratch"),
                 "workload": "bins %d, overlap %d, %d rows per launch%s" % (bins, overlap, rows, ", band scan included" if bands is not None else ""),
                 "value": rows / (ms_strict * 1e-3), "unit": "rows/s", "rows_per_step": rows, "steps": n64 - 2,
                 "ms_per_step": ms_strict, "dtype": "f64",
                 "roofline": {"bound": "hbm", "limiter": "FP64 issue + LDS exchanges (one workgroup per CU at M = 16384, four at 4096: csrc/ro_f64reg.hip) under the package power cap (profiles/r06_f64r_power.txt: 1382 W at C3, 1386 W / 2005 MHz at C2)" if reg
                                                        else "HBM scratch between the passes",
                              "unit": "GB/s", "peak": HBM_PEAK_GBS,
                              "achieved": alg * rows / (ms_strict * 1e-3) / 1e9,
                              "frac": alg * rows / (ms_strict * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "algorithmic_bytes_per_row": alg,
                              "kernel": ("f64r_kernel (csrc/ro_f64reg.hip)" if reg else "f64_pair_kernel (two launches per chunk)")
                                        + (" + scan_kernel" if bands is not None else ""),
                              "traffic": None}}
        # what the limiter names, as a number: the transform's flops against the FP64 vector peak (half the FP32 vector
        # peak of MI355X_MICROARCH.md: 157.3 / 2 TFLOP/s).  5 N log2 N is the customary count for an FFT; the kernel issues
        # 3 N log2 N FMAs (six-FMA butterflies) plus the fold and the square roots.
        flops = 5.0 * bins * float(np.log2(bins))
        entry["roofline"]["fp64_valu"] = {"unit": "TFLOP/s", "peak": 78.65, "flops_per_row": flops,
                                          "achieved": flops * rows / (ms_strict * 1e-3) / 1e12,
                                          "frac": flops * rows / (ms_strict * 1e-3) / 1e12 / 78.65}
        # bytes per launch by FETCH_SIZE / WRITE_SIZE from a committed rocprofv3 --pmc record of this shape (tools/r6/
        # f64r_pmc.sh), scaled to this launch's rows: counters cannot be read from inside the run
        rec_name = "r06_traffic_f64_%d.json" % bins
        try:
            with open(os.path.join(ROOT, "profiles", rec_name)) as fh:
                rec = json.load(fh)
            per_row = rec["traffic_bytes_per_launch"] / (rec["algorithmic_bytes_per_launch"] / alg)
            entry["roofline"]["traffic"] = per_row * rows
            entry["roofline"]["traffic_over_algorithmic"] = per_row / alg
            entry["roofline"]["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this shape)" % rec_name
        except Exception as e:
            entry["roofline"]["traffic_source"] = "profiles/%s unreadable: %s" % (rec_name, str(e)[:80])
        if parity:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import ro_oracle as O
            worst = 0.0
            for r in (0, rows - 1):
                seg = iq[r * hop:r * hop + bins].cpu().numpy()
                want = O.stft(seg, bins, overlap, w=st64.window)[0].astype(np.float64)
                got = rows_buf[r].cpu().numpy().astype(np.float64)
                worst = max(worst, float((np.abs(got - want) / np.maximum(want, 1e-300)).max()))
            entry["parity"] = {"rows_checked": [0, rows - 1], "max_err_per_bin_relative": worst, "tolerance_per_bin": 1e-5}
    if own:
        del iq, rows_buf
        torch.cuda.empty_cache()
    return entry


def large_leg(torch, ro, dev, local_rank, parity, bins, overlap, rows, cite, kernels, traffic):
    """A station config's shape above 32768 bins on the resident path: rows/s and the fraction of the HBM peak its
    algorithmic bytes (hop 8 + bins 4 per row) make, all kernels of the size included (the transform's + the band
    scan), HIP events by ro_stft_time_resident."""
    fs = 96000
    hop = bins - overlap
    samples = bins + hop * (rows - 1)
    iq = synth_iq(torch, samples, 0x10, dev)
    out_rows = torch.empty((rows, bins), dtype=torch.float32, device=dev)
    recs = torch.zeros((rows, 3), dtype=torch.float32, device=dev)
    f2b = lambda f: ro.frequency_to_bin(bins, fs, f)
    lo_d, hi_d = sorted((f2b(JSON_BOLID["min_detect"]), f2b(JSON_BOLID["max_detect"])))
    lo_n, hi_n = sorted((f2b(JSON_BOLID["min_noise"]), f2b(JSON_BOLID["max_noise"])))
    bands = ro.Bands(low_noise=lo_n, noise_width=hi_n - lo_n, low_detect=lo_d, detect_width=hi_d - lo_d,
                     avg_bins=max(f2b(JSON_BOLID["avg_freq_range"]) - f2b(0.0), 1))
    sptr = torch.cuda.current_stream(dev).cuda_stream
    with ro.Stft(bins=bins, overlap=overlap, sample_rate=fs, device=local_rank, bands=bands) as st:
        # (the device has idled through the previous leg's host work: 40-60 ms of launches, the second half timed)
        iters = 24
        ms, _, _ = st.time_resident(iq, ro.RO_IQ_F32, samples, 0, rows, out_rows, iters, d_records=recs, stream=sptr)
        torch.cuda.synchronize(dev)
        step_ms = float(np.mean(ms[iters // 2:]))
        alg = hop * 8 + bins * 4
        entry = {"workload": "bins %d, overlap %d (%s), %d rows per launch, band scan included" % (bins, overlap, cite, rows),
                 "value": rows / (step_ms * 1e-3), "unit": "rows/s", "ms_per_step": step_ms, "dtype": "f32",
                 "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                              "achieved": alg * rows / (step_ms * 1e-3) / 1e9,
                              "frac": alg * rows / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "kernel": kernels, "algorithmic_bytes_per_row": alg, "traffic": traffic}}
        if isinstance(traffic, str) and traffic.endswith(".json"):
            # bytes per launch by FETCH_SIZE / WRITE_SIZE from a committed rocprofv3 --pmc record of this shape, scaled to
            # this launch's rows (counters cannot be read from inside the run)
            try:
                with open(os.path.join(ROOT, "profiles", traffic)) as fh:
                    rec = json.load(fh)
                per_row = rec["traffic_bytes_per_launch"] / (rec["algorithmic_bytes_per_launch"] / alg)
                entry["roofline"]["traffic"] = per_row * rows
                entry["roofline"]["traffic_over_algorithmic"] = per_row / alg
                entry["roofline"]["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; L2 hit rate %s)" % (
                    traffic, "%.3f" % rec["tcc_hit_rate"] if rec.get("tcc_hit_rate") is not None else "n/a")
            except Exception as e:
                entry["roofline"]["traffic"] = None
                entry["roofline"]["traffic_source"] = "profiles/%s unreadable: %s" % (traffic, str(e)[:80])
        if parity:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import ro_oracle as O
            worst, picks = 0.0, [0, rows // 2, rows - 1]
            for r in picks:
                seg = iq[r * hop:r * hop + bins].cpu().numpy()
                want = O.stft(seg, bins, overlap, w=st.window)[0]
                got = out_rows[r].cpu().numpy()
                worst = max(worst, float(np.abs(got.astype(np.float64) - want).max() / want.max()))
            entry["parity"] = {"rows_checked": picks, "tolerance": 1e-5, "max_err_rel_to_row_max": worst}
    del iq, out_rows, recs
    torch.cuda.empty_cache()
    return entry


def streaming_leg(seconds, max_batch_rows, pcie=None):
    """rows/s and GB/s of the C++ host mirror's Backend::process path (tests/harness drives it; test infrastructure)"""
    lib = os.path.join(ROOT, "tests", "harness", "libro_host_harness.so")
    if not os.path.exists(lib):
        return {"error": "tests/harness/libro_host_harness.so is missing: run __graft_entry__.build()"}
    H = ctypes.CDLL(lib)
    H.ro_host_stream_bench.restype = ctypes.c_int
    H.ro_host_stream_bench.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                       ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    stats = (ctypes.c_double * 16)()
    block = 4096
    rc = H.ro_host_stream_bench(BINS, OVERLAP, FS, block, seconds, max_batch_rows, 2 * (BINS // block) + 64, stats)
    if rc != 0:
        return {"error": "ro_host_stream_bench returned %d" % rc}
    secs, smp, rws = stats[0], stats[1], stats[2]
    out = {"value": rws / secs, "unit": "rows/s", "seconds": secs, "rows": int(rws), "samples": int(smp),
           "samples_per_s": smp / secs, "real_time_factor": smp / secs / FS,
           "GBs_in_as_delivered": smp * 16 / secs / 1e9,           # struct Complex = two doubles (src/Backend.h:26-29)
           "GBs_in_on_the_bus": smp * 8 / secs / 1e9,              # ... narrowed to float32 on their way into the pinned staging buffer
           "GBs_rows_out": rws * BINS * 4 / secs / 1e9,
           "process_calls": int(stats[3]), "samples_per_call": block, "rows_per_kernel_launch": int(stats[4]),
           "ms_per_process_call_mean": stats[5], "ms_per_process_call_max": stats[6], "events_fired": int(stats[7]),
           "rows_by_dma_into_the_row_ring": bool(stats[15]),
           # ro_stft_timing of the stream's handle over the timed region (FFTBackend::logProcessingTimes' counterpart):
           # where a Backend::process call spends its time -- push = narrowing + staging + launching a batch when one is
           # complete; fetch = waiting for a batch's download and handing its rows to the recorders
           "stages": {"push_ms_avg": stats[8], "push_calls": int(stats[12]), "fetch_ms_avg": stats[9],
                      "fetch_calls": int(stats[13]), "batch_gpu_ms_avg": stats[10], "batches": int(stats[14]),
                      "row_gpu_us_avg": stats[11],
                      "host_us_per_row_outside_push_and_fetch":
                          (secs * 1e6 - stats[8] * 1e3 * stats[12] - stats[9] * 1e3 * stats[13]) / max(rws, 1.0)},
           "path": "FrontendDriver::process -> HipWaterfallBackend::process (raw ring: one bulk push per call; ro_stft_push "
                   "RO_IQ_F64 narrowing straight into the pinned staging buffer -> H2D -> kernels -> D2H of full rows INTO "
                   "the slots of the pinned row ring (ro_stft_set_row_sink) -> ro_stft_fetch of the scan records) -> "
                   "RingBuffer2D::push bookkeeping -> BolidRecorder::update per row; startStream (handle creation) and the "
                   "warm-up calls are outside the timed region, endStream inside"}
    if pcie:
        h2d_both, d2h_both, h2d, d2h = pcie
        # the bus's own bound for this path: per row hop samples of 8 bytes up, bins floats down, each direction at the
        # best rate the link gives it
        bound = 1.0 / max(HOP * 8 / (h2d * 1e9), BINS * 4 / (d2h * 1e9))
        out.update({"pcie_h2d_GBs": h2d, "pcie_d2h_GBs": d2h, "pcie_h2d_GBs_both_ways_at_once": h2d_both,
                    "pcie_d2h_GBs_both_ways_at_once": d2h_both, "pcie_bound_rows_per_s": bound,
                    "frac_of_pcie": out["value"] / bound,
                    "pcie_note": "pinned 256 MiB copies measured in this run, each direction alone (the bound: the better of one "
                                 "stream's copies and three streams' at once, as the path issues them) and both at "
                                 "once (for the record); the bound is the slower direction's time per row (hop x 8 bytes "
                                 "up, bins x 4 bytes down) at its one-direction rate"})
    return out


class ClockPowerSampler:
    """shader clock and package power of THIS GPU during the run, from the amdgpu hwmon files in sysfs (plain file reads
    from a thread of this process: no child process, no rocm-smi).  The card is the one whose PCI address is the
    device's (torch: pci_domain_id / pci_bus_id / pci_device_id); without a match nothing is reported."""

    def __init__(self, torch, dev_index):
        import glob
        import threading
        self.files = None
        self.samples = []                      # (perf_counter, sclk MHz, watts)
        self.card = None
        try:
            p = torch.cuda.get_device_properties(dev_index)
            want = "%04x:%02x:%02x." % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
            for d in sorted(glob.glob("/sys/class/drm/card*/device")):
                if os.path.basename(os.path.realpath(d)).startswith(want):
                    hw = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*")))
                    if hw and os.path.exists(os.path.join(hw[0], "freq1_input")):
                        pw = [f for f in ("power1_input", "power1_average") if os.path.exists(os.path.join(hw[0], f))]
                        self.files = (os.path.join(hw[0], "freq1_input"), os.path.join(hw[0], pw[0]) if pw else None)
                        self.card = os.path.basename(os.path.dirname(d)) + " " + os.path.basename(os.path.realpath(d))
                    break
        except Exception:
            self.files = None
        self._stop = threading.Event()
        self._go = threading.Event()           # cleared while the timed region runs: no reads, no GIL, no SMU queries
        self._go.set()
        self._thread = threading.Thread(target=self._run, daemon=True) if self.files else None

    def _read(self, f):
        try:
            with open(f) as fh:
                return float(fh.read().strip())
        except Exception:
            return float("nan")

    def read_once(self):
        if self.files:
            t = time.perf_counter()
            mhz = self._read(self.files[0]) / 1e6
            w = self._read(self.files[1]) / 1e6 if self.files[1] else float("nan")
            self.samples.append((t, mhz, w))

    def _run(self):
        while not self._stop.is_set():
            self._go.wait()
            if self._stop.is_set():
                break
            self.read_once()
            time.sleep(0.002)

    def pause(self):
        """entering the timed region: the thread parks (it neither takes the GIL from the launch loop nor makes the
        firmware answer a power query while the kernel under test runs); ONE read brackets the region on each side"""
        self._go.clear()
        self.read_once()

    def resume(self):
        self.read_once()
        self._go.set()

    def start(self):
        if self._thread:
            self._thread.start()

    def stop(self):
        if self._thread:
            self._stop.set()
            self._go.set()
            self._thread.join()

    def summary(self, t0, t1):
        """mean / min / max inside [t0, t1] (the timed region) and, for context, over the warm-up before it"""
        if not self.files:
            return {"available": False, "note": "no amdgpu hwmon files for this device's PCI address in sysfs"}

        def stats(rows, col):
            x = np.array([r[col] for r in rows], dtype=np.float64)
            x = x[np.isfinite(x)]
            return None if x.size == 0 else {"mean": float(x.mean()), "min": float(x.min()), "max": float(x.max())}
        inside = [r for r in self.samples if t0 <= r[0] <= t1]
        before = [r for r in self.samples if r[0] < t0]
        return {"available": True, "card": self.card,
                "source": "hwmon freq1_input / power1_input: one read per ~2 ms during warm-up and soak; the sampler is parked "
                          "inside the timed region, which is bracketed by one read just before and one just after it",
                "timed_region": {"samples": len(inside), "sclk_mhz": stats(inside, 1), "package_watts": stats(inside, 2)},
                "warmup": {"samples": len(before), "sclk_mhz": stats(before, 1), "package_watts": stats(before, 2)}}


class RcclComm:
    """an ncclComm_t of this job's ranks for the product's own exchange (--exchange capi): created through ctypes on
    librccl, the unique id travelling over the torch.distributed group that already exists"""

    class Uid(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    def __init__(self, dist, world, rank):
        self.rccl = ctypes.CDLL("librccl.so.1")
        uid = RcclComm.Uid()
        if rank == 0:
            assert self.rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
        box = [ctypes.string_at(ctypes.byref(uid), 128) if rank == 0 else None]      # (all 128 bytes, NULs included)
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        ctypes.memmove(ctypes.byref(uid), box[0], 128)
        self.comm = ctypes.c_void_p()
        self.rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, RcclComm.Uid, ctypes.c_int]
        rc = self.rccl.ncclCommInitRank(ctypes.byref(self.comm), world, uid, rank)
        if rc != 0:
            raise RuntimeError("ncclCommInitRank failed with code %d" % rc)

    def close(self):
        self.rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        self.rccl.ncclCommDestroy(self.comm)


def launch_ranks(n):
    """`python3 bench.py --gpus N` from a plain shell: start the N ranks as a CHILD job
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <the same arguments>) and exit with
    its code.  This process has not imported torch and has made no HIP call -- it never touches a GPU, it only waits;
    rank 0 of the child prints the JSON line on the stdout it inherits."""
    import socket
    with socket.socket() as s:                                # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    global BINS, OVERLAP, HOP, ALG_BYTES_PER_ROW
    if a.bins != BINS or a.overlap is not None:             # non-headline shapes (e.g. C2: --bins 4096 --overlap 2048)
        BINS = a.bins
        OVERLAP = a.overlap if a.overlap is not None else (3 * BINS) // 4
        HOP = BINS - OVERLAP
        ALG_BYTES_PER_ROW = HOP * 8 + BINS * 4
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Whatever needs a compiler or a child process happens BEFORE this process touches the GPU (a fork + exec from a
    # process that holds a device is a hazard on this pool): the -O0 twin of the oracle for the cpu_baseline_O0 leg.
    o0_lib, o0_err = os.path.join(ROOT, "oracle", "libro_oracle_O0.so"), None
    if world == 1 and not a.no_cpu_baseline and not os.path.exists(o0_lib):
        try:
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libro_oracle_O0.so"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception as e:                               # no compiler on the box: the leg says so
            o0_err = str(e)[:200]
    # (read by the ROCm runtime when it initialises -- before the first HIP call: dmabuf IPC is what RCCL needs here)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    if a.gpus != world:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d: the two must agree" % (a.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    if a.comm == "gloo-host":
        local_rank = 0
        if a.exchange == "capi" and world > 1:
            sys.exit("--exchange capi needs one GPU per rank (RCCL refuses two ranks on one device): use --comm nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.comm == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    ro = importlib.import_module("radio-observer_amd")
    sh = ro.sharding()
    bands = make_bands(ro)
    snap_lo = ro.frequency_to_bin(BINS, FS, JSON_SNAPSHOT[0])
    snap_hi = ro.frequency_to_bin(BINS, FS, JSON_SNAPSHOT[1])
    c5 = a.workload == "c5"
    tile = (snap_lo, snap_hi - snap_lo) if (world > 1 or c5) else None
    exchanging = world > 1 or (c5 and a.exchange == "capi")       # (a one-rank communicator still runs the C-ABI calls)

    # ---- this rank's input and its place in the job
    if c5:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import util as tutil                                   # the counter-based C5 generator (test infrastructure)
        total_samples = int(a.c5_seconds * FS)
        R_total = ro.row_count(total_samples, BINS, OVERLAP)
        first_row, R = ro.shard_rows(R_total, world, rank)
        s0, samples = ro.shard_samples(first_row, R, BINS, OVERLAP)
        iq = tutil.c5_slice(torch, s0, samples, total_samples=total_samples, device=dev)
        R_max = ro.shard_max_rows(R_total, world)
    else:
        R = a.rows
        R_total, R_max, first_row = R * world, R, R * rank
        samples = BINS + HOP * (R - 1)
        # (every rank its own noise: weak scaling measures rate, not a result; --workload c5 is the comparable one)
        iq = synth_iq(torch, samples, 0xC3 + rank, dev)
    rows = torch.empty((R, BINS), dtype=torch.float32, device=dev)
    recs = [torch.zeros((R, 3), dtype=torch.float32, device=dev) for _ in range(2)]   # ro_scan_record_t = 12 B
    # N > 1: the STFT kernel of the large plans owns every register of the CUs it runs on, so the exchange's
    # kernels (side stream) would only run between two of its launches; one CU per XCD is left to them
    st = ro.Stft(bins=BINS, overlap=OVERLAP, sample_rate=FS, device=local_rank, bands=bands, tile=tile,
                 window=ro.RO_WINDOW_HANN if a.window == "hann" else ro.RO_WINDOW_NUTTALL,
                 spare_cus_per_xcd=1 if world > 1 else 0)
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream
    L = ro.library()

    tiles = g_tiles = g_recs = stage = comm_stream = None
    comm_done = [None, None]
    rccl = None
    if tile:
        tcols = tile[1]
        tiles = [torch.empty((R, tcols), dtype=torch.float32, device=dev) for _ in range(2)]
    if exchanging:
        g_tiles = [torch.empty((world * R_max, tcols), dtype=torch.float32, device=dev) for _ in range(2)]
        g_recs = [torch.empty((world * R_max, 3), dtype=torch.float32, device=dev) for _ in range(2)]
        comm_stream = torch.cuda.Stream(device=dev)
        if a.exchange == "capi":
            rccl = RcclComm(dist, world, rank)
            stage = [torch.empty((R_max, max(tcols, 3)), dtype=torch.float32, device=dev) for _ in range(2)]

    # how a step ends: "all" = ncclAllGather, "direct" = the all-gather as grouped point-to-point transfers, "root" = gather
    # to rank 0, "none" = compute only
    mode = {"now": ("direct" if a.gather == "all" and a.pattern == "direct" else a.gather) if exchanging else "none", "note": None}

    def exchange(out, inp, cols, b, which):
        """one buffer (band tile or records) of this step: all-gather, or gather to rank 0.  `out` = world x R_max rows
        as an equal-block all-gather leaves them (root mode on rank 0 with the C ABI: R_total rows, already in place)"""
        cptr = comm_stream.cuda_stream
        if a.exchange == "capi":
            if mode["now"] == "all":
                rc = L.ro_allgather_rows(rccl.comm, ctypes.c_void_p(inp.data_ptr()), R, R_total, world, rank, cols * 4,
                                         ctypes.c_void_p(stage[which].data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                         ctypes.c_void_p(cptr))
            elif mode["now"] == "direct":
                rc = L.ro_allgather_rows_direct(rccl.comm, ctypes.c_void_p(inp.data_ptr()), R, R_total, world, rank, cols * 4,
                                                ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cptr))
            else:
                rc = L.ro_gather_rows(rccl.comm, ctypes.c_void_p(inp.data_ptr()), R, R_total, world, rank, 0, cols * 4,
                                      ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cptr))
            if rc != 0:
                raise RuntimeError("C-ABI exchange failed: %s" % L.ro_last_error())
            return
        if mode["now"] == "direct":
            # the same schedule as ro_allgather_rows_direct, through torch.distributed's point-to-point operations
            if a.comm == "nccl":
                sh.gather_rows_direct(inp[:R], R_total, out=out[:R_total])
            else:                                               # rehearsal: through host memory
                out[:R_total].copy_(sh.gather_rows_direct(inp[:R].cpu(), R_total))
            return
        send = sh.pad_block(inp, R_total, world) if c5 else inp
        if a.comm == "nccl":
            if mode["now"] == "root":
                parts = list(out.view((world,) + tuple(send.shape)).unbind(0)) if rank == 0 else None
                dist.gather(send, parts, dst=0)
            else:
                dist.all_gather_into_tensor(out, send)
        else:                                                   # rehearsal: the same exchange through host memory
            h_in = send.cpu()
            h_out = torch.empty(out.shape, dtype=out.dtype)
            if mode["now"] == "root":
                parts = list(h_out.view((world,) + tuple(h_in.shape)).unbind(0)) if rank == 0 else None
                dist.gather(h_in, parts, dst=0)
            else:
                dist.all_gather_into_tensor(h_out, h_in)
            out.copy_(h_out)

    def step(i):
        b = i & 1
        if comm_done[b] is not None:
            stream.wait_event(comm_done[b])                 # tile buffer b is free again
        st.run_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, d_tile=tiles[b] if tile else None, d_records=recs[b],
                        stream=sptr)
        if not exchanging or mode["now"] == "none":
            return
        ready = torch.cuda.Event()
        ready.record(stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ready)
            exchange(g_tiles[b], tiles[b], tcols, b, 0)
            exchange(g_recs[b], recs[b], 3, b, 1)
            ev = torch.cuda.Event()
            ev.record(comm_stream)
            comm_done[b] = ev

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev if a.comm == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def min_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev if a.comm == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item())

    def usable(m):
        """probe one way of ending a step OUTSIDE any timed region; every rank learns the same answer"""
        keep = mode["now"]
        mode["now"] = m
        ok = 1.0
        try:
            step(0)
            step(1)
            torch.cuda.synchronize(dev)
        except Exception as e:                                  # e.g. a backend without gather
            ok = 0.0
            mode["note"] = "%s refused by the backend (%s)" % (m, str(e)[:120])
        if world > 1:
            ok = min_over_ranks(ok)
        mode["now"] = keep
        return ok > 0.5

    # the way a step ends is chosen ONCE, before anything is timed, and by all ranks together
    if exchanging and mode["now"] in ("root", "direct") and not usable(mode["now"]):
        mode["now"] = "all"
        mode["note"] = (mode["note"] or "") + ": all-gather instead"
    sampler = ClockPowerSampler(torch, local_rank) if rank == 0 else None
    if sampler:
        sampler.start()
    for i in range(a.prewarm + a.warmup):
        step(i)
    fence()
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_bracket0 = time.perf_counter()
    if sampler:
        sampler.pause()
    t0 = time.perf_counter()
    ev_a.record(stream)
    for i in range(a.steps):
        step(i)
    ev_b.record(stream)
    fence()
    dt = time.perf_counter() - t0
    if sampler:
        sampler.resume()
    t_bracket1 = time.perf_counter()
    clock_power = None
    if sampler:
        # ... and on, outside the timed region: what the package settles at under this kernel (the power reading is a
        # slow average; the timed region of the default run is 50 ms)
        t_soak0 = time.perf_counter()
        soak_steps = 0
        if world == 1 and a.soak_seconds > 0:
            while time.perf_counter() - t_soak0 < a.soak_seconds:
                for i in range(50):
                    step(i)
                torch.cuda.synchronize(dev)
                soak_steps += 50
        t_soak1 = time.perf_counter()
        sampler.stop()
        clock_power = sampler.summary(t_bracket0, t_bracket1)
        if soak_steps and clock_power.get("available"):
            tail = sampler.summary(t_soak0 + 0.6 * (t_soak1 - t_soak0), t_soak1)["timed_region"]      # the last 40 % of it
            clock_power["soak"] = {"seconds": t_soak1 - t_soak0, "steps": soak_steps,
                                   "ms_per_step": (t_soak1 - t_soak0) / soak_steps * 1e3,
                                   "last_40_percent": tail}
    gpu_ms_per_step = ev_a.elapsed_time(ev_b) / a.steps       # HIP events on the launch stream, over the timed region
    if world > 1:
        dt = max_over_ranks(dt)

    # ---- c5: what the job produced, stitched in row order on rank 0, hashed (N ranks == 1 rank, bit for bit)
    # rank 0's own scan records of the last step: its input (seed 0xC3, or the first time chunk of the C5 stream) and
    # therefore its records are the same whatever the number of ranks -- an N-rank run's line can be checked against the
    # one-rank run's, bit for bit, without the C5 workload
    rank0_hash = None
    if rank == 0:
        import hashlib
        rank0_hash = hashlib.sha256(recs[(a.steps - 1) & 1].cpu().numpy().tobytes()).hexdigest()[:32]
    c5_hash = None
    if c5:
        import hashlib
        last = (a.steps - 1) & 1
        if exchanging and mode["now"] != "none":
            if (a.exchange == "capi" and mode["now"] == "root") or mode["now"] == "direct":
                t_all, r_all = g_tiles[last][:R_total], g_recs[last][:R_total]          # rows already in place
            else:
                t_all, r_all = sh.stitch(g_tiles[last], R_total, world), sh.stitch(g_recs[last], R_total, world)
        else:
            t_all, r_all = tiles[last], recs[last]
        if rank == 0 and (world == 1 or mode["now"] != "none"):
            hh = hashlib.sha256()
            hh.update(t_all.cpu().numpy().tobytes())
            hh.update(r_all.cpu().numpy().tobytes())
            c5_hash = hh.hexdigest()[:32]

    # ---- N > 1, OUTSIDE the timed region of `value`: the same K steps with the other ways of ending a step, so that
    # one run separates compute from exchange (all-gather as north_star words it, gather to rank 0, compute only)
    legs = None
    if world > 1 and a.gather != "none" and not a.no_legs:
        main_mode = mode["now"]
        legs = {main_mode: R_total * a.steps / dt}
        for m in ("all", "direct", "root", "none"):
            if m in legs:
                continue
            if m != "none" and not usable(m):
                legs[m] = "unavailable: %s" % mode["note"]
                continue
            mode["now"] = m
            for i in range(max(a.warmup, 2)):
                step(i)
            fence()
            t1 = time.perf_counter()
            for i in range(a.steps):
                step(i)
            fence()
            legs[m] = R_total * a.steps / max_over_ranks(time.perf_counter() - t1)
        mode["now"] = main_mode

    # ---- kernel durations with HIP events on the launch stream (same launches as the timed loop)
    ms_all, k_stft, k_scan = st.time_resident(iq, ro.RO_IQ_F32, samples, 0, R, rows, max(a.steps, 5),
                                              d_records=recs[0], stream=sptr)
    torch.cuda.synchronize(dev)

    out = None
    if rank == 0:
        total_rows = R_total * a.steps
        value = total_rows / dt
        # one step = one kernel (scan and tile are its epilogue) and nothing else on the launch stream: with an exchange
        # every step begins with a wait for the side stream, which would be counted as kernel time
        fused = BINS == 32768 and world == 1 and not exchanging
        # The dominant kernel's average launch duration: HIP events on the launch stream around the K launches of the
        # TIMED region when a step is exactly that one kernel (bins = 32768: scan and tile are its epilogue); for the
        # plans with a separate scan kernel the transform's own share comes from the event pairs of
        # ro_stft_time_resident, taken right behind the timed region.
        kernel_ms = gpu_ms_per_step if fused else k_stft
        achieved = ALG_BYTES_PER_ROW * R / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "FFT rows/sec (spectra/sec), N=32768 75% overlap, incl. per-row bolid scan",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "prewarm": a.prewarm,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if c5 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("C5: one synthetic IQ stream of %.0f s at 48 kHz (%d rows), FFT bins=%d, overlap=%d, "
                                    "noise + a chirp every 30 s, split over the ranks by time chunk" %
                                    (a.c5_seconds, R_total, BINS, OVERLAP)) if c5 else
                                   "%s: synthetic IQ 48 kHz, FFT bins=%d, overlap=%d (%d%%), "
                                   "%s window, waterfall magnitude rows + BolidRecorder scan"
                                   % ("C3/C4" if (BINS, OVERLAP) == (32768, 24576) else
                                      "C2" if (BINS, OVERLAP) == (4096, 2048) else "custom", BINS, OVERLAP,
                                      round(100.0 * OVERLAP / BINS), a.window.capitalize()),
                       "rows_per_step_per_gpu": R, "samples_per_step_per_gpu": samples,
                       "untimed_launches_before_the_timed_region": a.prewarm + a.warmup,
                       "input": "float32 I/Q resident in HBM", "output": "float32 rows in HBM",
                       "parallelism": "time-chunk per GPU" + (("; %s of band tile [%d,+%d) + scan records per step (%s), "
                                                               "overlapped with the next step"
                                                               % (({"all": "all-gather (ncclAllGather)", "direct": "all-gather as direct point-to-point transfers",
                                                                    "root": "gather to rank 0"}[mode["now"]],)
                                                                  + tile + ("C ABI: ro_allgather_rows / ro_allgather_rows_direct / ro_gather_rows"
                                                                            if a.exchange == "capi" else
                                                                            "torch.distributed",)))
                                                              if exchanging and mode["now"] != "none" else
                                                              ("; compute only (--gather none)" if exchanging else "")),
                       **({"gather_note": mode["note"]} if mode["note"] else {}),
                       **({"c5_hash_of_stitched_band_and_records": c5_hash} if c5_hash else {}),
                       **({"rank0_scan_records_hash": rank0_hash} if rank0_hash and not c5 else {}),
                       **({"exchange_legs_rows_per_s": legs} if legs else {})},
            "roofline": {"bound": "hbm",
                         # what holds the kernel below that bound (profiles/r05_stft_c3_summary.txt, r04_ledgers.md): the
                         # package's power cap under 3 N log2 N packed FMAs per row, not HBM (traffic = 1.01 x algorithmic)
                         "limiter": "package power cap (VALU)" if BINS == 32768 else "see DESIGN.md 4",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # the same bytes over the wall-clock step of the timed region (host gaps and exchange included)
                         "frac_step": ALG_BYTES_PER_ROW * R / (dt / a.steps) / 1e9 / HBM_PEAK_GBS,
                         # ... and over the event pairs of a separate loop of launches behind the timed region
                         "frac_posthoc": ALG_BYTES_PER_ROW * R / (k_stft * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": a.pmc_traffic if a.pmc_traffic is not None else pmc_traffic(R),
                         "traffic_source": "--pmc-traffic" if a.pmc_traffic is not None else
                                           "newest profiles/r*_traffic.json (rocprofv3 --pmc passes of this command; "
                                           "counters cannot be read from inside the run)",
                         "kernel": ("stft32k_kernel" if BINS == 32768 else "stft_kernel<%d>" % BINS),
                         "kernel_ms": kernel_ms,
                         "kernel_ms_source": "HIP events on the launch stream around the %d launches of the timed region"
                                             % a.steps if fused else "ro_stft_time_resident event pairs behind the timed region",
                         "kernel_ms_posthoc": k_stft,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_ROW * R,
                         "scan_kernel_ms": k_scan},
            "device": st.device_name,
            # what the device ran at while it was timed ("make the line self-qualifying"): at the package's
            # 1400 W cap the shader clock, not the kernel, differs between boxes
            "clock_power": clock_power,
            "gpu_ms_per_step_events": gpu_ms_per_step, "step_ms_back_to_back": float(np.mean(ms_all)),
            "step_ms_median": float(np.median(ms_all)), "step_ms_min": float(np.min(ms_all)),
        }

        # the steady state under the package's power cap: the same launches for --soak-seconds behind the timed region (the
        # driver's 20-step region ends before the package has reached its cap; this is what a long job runs at)
        if clock_power and isinstance(clock_power.get("soak"), dict):
            soak_ms = clock_power["soak"]["ms_per_step"]
            out["roofline"]["frac_at_power_cap"] = ALG_BYTES_PER_ROW * R / (soak_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["ms_per_step_at_power_cap"] = soak_ms

        # ---- device copy bandwidth for context (float32 copy of the row buffer)
        c = torch.empty_like(rows)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            c.copy_(rows)
        e1.record()
        torch.cuda.synchronize(dev)
        out["device_copy_GBs"] = 2 * rows.numel() * 4 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        out["roofline"]["frac_of_device_copy"] = achieved / out["device_copy_GBs"]      # report both
        del c

        # ---- parity spot check of this run's output (oracle = checker only)
        if not a.no_parity:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import ro_oracle as O
            pick = [0, 1, R // 2, R - 1]
            worst, pb_max, pb_over, pb_bins = 0.0, 0.0, 0, 0
            scan_ok = True
            rec_host = recs[0].cpu().numpy().view(np.uint8).reshape(R, 12)
            rec_host = np.frombuffer(rec_host.tobytes(), dtype=ro.capi.SCAN_DTYPE)
            for r in pick:
                seg = iq[r * HOP:r * HOP + BINS].cpu().numpy()
                want = O.stft(seg, BINS, OVERLAP, w=st.window)[0]
                got = rows[r].cpu().numpy()
                worst = max(worst, float(np.abs(got.astype(np.float64) - want).max() / want.max()))
                pb = np.abs(got.astype(np.float64) - want) / np.maximum(want, 1e-300)
                pb_max = max(pb_max, float(pb.max()))
                pb_over += int((pb > 1e-5).sum())
                pb_bins += pb.size
                n, p, av = O.scan_rows(got[None, :], bands.low_noise, bands.noise_width, bands.low_detect,
                                       bands.detect_width, bands.avg_bins)
                scan_ok &= (n[0] == rec_host["noise"][r] and p[0] == rec_host["peak"][r]
                            and av[0] == rec_host["average"][r])
            out["parity"] = {"rows_checked": pick, "max_err_rel_to_row_max": worst, "tolerance": 1e-5,
                             # the other reading of north_star's "1e-5 relative": every bin against its own magnitude.
                             # The float32 default does not meet it on the weakest bins of a row (reported, and pinned by
                             # tests/test_gpu_strict.py); strict_precision below is the mode that does.
                             "per_bin": {"max_rel": pb_max, "frac_over_1e-5": pb_over / max(1, pb_bins), "met": pb_max <= 1e-5},
                             "scan_records_bit_exact": bool(scan_ok)}

        # ---- the strict-precision mode next to the headline (never the headline): the same input and rows per step, then
        # the C2 shape
        if world == 1 and not a.no_strict:
            out["strict_precision"] = strict_leg(torch, ro, dev, local_rank, not a.no_parity, BINS, OVERLAP, R, FS, bands,
                                                 ro.RO_WINDOW_HANN if a.window == "hann" else ro.RO_WINDOW_NUTTALL,
                                                 max(a.steps, 20), iq=iq, rows_buf=rows, recs_buf=recs[1])
            if (BINS, OVERLAP) == (32768, 24576) and not c5:
                out["strict_precision"]["c2"] = strict_leg(torch, ro, dev, local_rank, not a.no_parity, 4096, 2048, 65536, FS,
                                                           None, ro.RO_WINDOW_NUTTALL, max(a.steps, 20))
                # ... and C1's (the reference's own CPU-runnable case: 1024 bins, 50 %), four rows to a workgroup
                out["strict_precision"]["c1"] = strict_leg(torch, ro, dev, local_rank, not a.no_parity, 1024, 512, 262144, FS,
                                                           None, ro.RO_WINDOW_NUTTALL, max(a.steps, 20))

        # ---- the station configs' own shapes next to the headline (never the headline), inputs resident:
        # Ionozor.json:27-28 on the four-step pair of kernels, Bolidozor.json:45-46 on the one-kernel large form
        if world == 1 and not a.no_large and not c5 and BINS == 32768:
            out["ionozor"] = large_leg(torch, ro, dev, local_rank, not a.no_parity, 524288, 262144, 1024, "Ionozor.json:27-28",
                                       "four_cols_kernel + four_rows_kernel + scan_kernel (csrc/ro_fourstep.hip)",
                                       "r05_traffic_524288.json")
            out["bolidozor"] = large_leg(torch, ro, dev, local_rank, not a.no_parity, 65536, 49152, 8192, "Bolidozor.json:45-46",
                                         "stft_kernel<Plan32768, ., 3> (two workgroups per stream row) + scan_kernel",
                                         "r05_traffic_65536.json")
            if not a.no_strict:                              # ... and the same shape in the reference's arithmetic type
                out["bolidozor"]["f64"] = strict_leg(torch, ro, dev, local_rank, not a.no_parity, 65536, 49152, 8192, 96000,
                                                     None, ro.RO_WINDOW_NUTTALL, 12)

        # ---- the drop-in path at full speed (never the headline): Frontend::process -> HipWaterfallBackend::process
        # with 4096-sample vector<Complex> calls (src/RawStream.cpp:44-66) -> kernels -> full rows back to the host row
        # ring -> BolidRecorder::update per row; PCIe both ways included, handle creation and warm-up calls excluded
        if world == 1 and not a.no_streaming and not c5:
            pcie = pcie_copy_rates(torch, dev)
            out["streaming"] = streaming_leg(a.stream_seconds, 0, pcie)
            # (the Backend's own default bounds a batch by latency -- one second of rows = 6; the same path with 256 rows
            # per launch, what a file replay would ask for through WaterfallConfig::max_batch_rows)
            out["streaming_batch256"] = streaming_leg(a.stream_seconds, 256, pcie)

        # ---- CPU baseline (rank 0, N=1 only): the oracle port on the host cores of this box.  Its FP64 transform runs
        # on libfftw3 itself (fftw_plan_dft_1d(..., FFTW_FORWARD, FFTW_ESTIMATE) + fftw_execute, the reference's call
        # sites: src/FFTBackend.cpp:117-120,236) when this host has libfftw3.so.3, else on a vendor FFT behind the same
        # fftw3 API (MKL's FFTW3 interface), else on the oracle's own radix-2 transform; `fft_engine` says which.
        if world == 1 and not a.no_cpu_baseline:
            n_cpu = min(R, 32768 if BINS >= 16384 else 262144)
            host = iq[:BINS + HOP * (n_cpu - 1)].cpu().numpy()
            O = oracle_module()
            ncores, navail = host_cores(), os.cpu_count() or 0
            done_p, cdt_p = cpu_baseline(host, st.window, bands, a.cpu_seconds / 3, O)
            port = {"value": done_p / cdt_p, "unit": "rows/s", "cores": 1, "kind": "port", "fft_engine": "port",
                    "sample": "first %d rows of the same input (%.1f s): oracle/ro_oracle.c -O2, its own FP64 radix-2 "
                              "FFT + scan, single thread" % (done_p, cdt_p)}
            have_lib = O.use_fftw(True)
            if have_lib:
                engine = O.fft_engine()
                done, cdt = cpu_baseline(host, st.window, bands, a.cpu_seconds * 2 / 3, O)
                out["cpu_baseline"] = {"value": done / cdt, "unit": "rows/s", "cores": 1, "kind": "port",
                                       "fft_engine": engine,
                                       "sample": "first %d rows of the same input (%.1f s): oracle/ro_oracle.c -O2 (framing, "
                                                 "window, magnitude, scan) with the FP64 transform on %s through the fftw3 "
                                                 "API, FFTW_ESTIMATE plan, single thread; %d of the host's %d cores are "
                                                 "available to this process" % (done, cdt, engine, ncores, navail)}
                out["cpu_baseline_port_fft"] = port
            else:
                port["sample"] += "; no libfftw3.so.3 and no fftw3-API vendor library on this host; %d of the host's %d " \
                                  "cores are available to this process" % (ncores, navail)
                out["cpu_baseline"] = port
            # the upper bound of "what this host could do": one independent slice of the stream per core
            cores = min(ncores, 64)
            if cores > 1:
                done_n, cdt_n = cpu_baseline_threads(host, st.window, bands, a.cpu_seconds / 2, cores)
                out["cpu_baseline_all_cores"] = {"value": done_n / cdt_n, "unit": "rows/s", "cores": cores,
                                                 "kind": "port", "fft_engine": O.fft_engine(),
                                                 "sample": "%d rows of the same input in %.1f s, one thread per core"
                                                           % (done_n, cdt_n)}
            O.use_fftw(False)
            # the reference is built -O0 (Makefile:26-28): the same port compiled that way, radix-2 transform
            try:
                if not os.path.exists(o0_lib):                   # (built before the first HIP call, or by build())
                    raise RuntimeError(o0_err or "oracle/libro_oracle_O0.so is missing")
                O0 = oracle_module(o0_lib)
                done0, cdt0 = cpu_baseline(host, st.window, bands, min(4.0, a.cpu_seconds / 3), O0)
                out["cpu_baseline_O0"] = {"value": done0 / cdt0, "unit": "rows/s", "cores": 1, "kind": "port",
                                          "fft_engine": "port",
                                          "sample": "first %d rows (%.1f s): the same source compiled -O0 like the "
                                                    "reference's Makefile:26-28" % (done0, cdt0)}
            except Exception as e:                                   # no compiler on the box: say so, do not fail the bench
                out["cpu_baseline_O0"] = {"error": str(e)[:200]}
        print(json.dumps(out), flush=True)

    st.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
