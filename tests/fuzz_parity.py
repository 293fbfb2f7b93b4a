"""Randomised parity of the resident entry points against the oracle (test infrastructure).

The parametrised tests pin one feature at a time; this draws whole configurations -- size (every power of two and
chirp-z lengths), overlap (0, 1, N-1, odd), row range, sample format, I/Q gain, window, precision mode, scan bands,
band tile, row stride and base offset, CUs left free -- and checks every output the call makes:

  * rows against the oracle's (norm-wise bar of tests/test_gpu_stft.py; the FP64 modes against float rounding),
  * the cells of the row buffer the call must not touch (stride padding, rows outside the range, guard words),
  * the tile against the rows it copies, bit for bit,
  * the scan records against the oracle's scan of those rows, bit for bit,
  * complex spectra (power-of-two sizes, F32) against the oracle's spectrum.

`--kind stream` feeds the same draws call by call (ro_stft_push / flush / fetch, with and without a row sink, pieces and
fetches of random size, float / double / complex / int16 deliveries) and holds every streamed row, tile and record
against the resident call's, bit for bit.

`python tests/fuzz_parity.py --seconds 300 --seed 1 [--kind stream]` on a GPU box; `--case K` replays one case of a seed.
tests/test_gpu_fuzz.py runs a short fixed-seed slice of it in the GPU suite.
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

TOL_F32 = 1e-5          # BASELINE.md's norm-wise bar (tests/test_gpu_stft.py)
TOL_F64 = 1.5e-7        # FP64 arithmetic, float32 rows: the oracle's bits but for the rounding of the stored row where the two
                        # double results straddle a float boundary (13 000 cases on the device: 2.8e-10 at worst --
                        # small bins only --, profiles/r05_fuzz.txt)
TOL_F64_PER_BIN = 2e-7  # ... and bin by bin (tests/test_gpu_strict.py)
TOL_SPEC = 1e-5         # complex spectra, against the largest bin of the row

POW2_SMALL = [256, 512, 1024, 2048, 4096]
POW2_MID = [8192, 16384, 32768]
POW2_LARGE = [65536, 131072, 262144, 524288, 1048576]


def draw_case(rng):
    """one configuration as a plain dict (printable, replayable)"""
    cls = rng.choice(["small", "mid", "large", "czt_small", "czt_mid", "czt_large"],
                     p=[0.30, 0.25, 0.12, 0.15, 0.12, 0.06])
    if cls == "small":
        bins = int(rng.choice(POW2_SMALL))
    elif cls == "mid":
        bins = int(rng.choice(POW2_MID))
    elif cls == "large":
        bins = int(rng.choice(POW2_LARGE, p=[0.35, 0.25, 0.2, 0.12, 0.08]))
    elif cls == "czt_small":
        bins = 2 * int(rng.integers(129, 4096))
    elif cls == "czt_mid":
        bins = 2 * int(rng.integers(4096, 20000))
    else:
        bins = 2 * int(rng.integers(20000, 262143))
    pow2 = bins & (bins - 1) == 0
    ov_kind = rng.choice(["zero", "one", "half", "three_quarters", "max", "any", "beyond"],
                         p=[0.12, 0.05, 0.2, 0.2, 0.1, 0.28, 0.05])
    overlap = {"zero": 0, "one": 1, "half": bins // 2, "three_quarters": bins * 3 // 4, "max": bins - 1,
               "any": int(rng.integers(0, bins)), "beyond": bins + int(rng.integers(0, 100))}[ov_kind]
    eff = min(max(overlap, 0), bins - 1)                     # src/FFTBackend.cpp:108-109
    hop = bins - eff
    budget = 1 << 22 if bins <= 32768 else 3 << 20          # points the oracle transforms per case
    max_rows = max(1, min(40, budget // bins))
    if hop <= 8:                                             # many rows are cheap in samples, not in transforms
        max_rows = min(max_rows, 24)
    total = int(rng.integers(1, max_rows + 1))
    first = int(rng.integers(0, total))
    rows = int(rng.integers(1, total - first + 1))
    tail = int(rng.integers(0, min(hop, 9)))                 # leftover samples that make no row
    fmt = str(rng.choice(["f32", "i16"], p=[0.7, 0.3]))
    gain = float(rng.choice([0.0, 0.0, 0.5, -3.25, 37.5]))
    window = str(rng.choice(["nuttall", "hann", "custom"], p=[0.6, 0.15, 0.25]))
    precision = int(rng.choice([0, 1], p=[0.7, 0.3]))          # (draws the same number of values as ever: the seeds' cases)
    if not pow2:                                             # RO_PRECISION_F64 is for power-of-two bins only
        precision = 0
    if bins > 131072 and precision:                          # keep the FP64 oracle + device time bounded
        precision = int(rng.choice([0, 1], p=[0.7, 0.3]))
    bands = None
    if rng.random() < 0.6:
        avg = int(rng.integers(1, min(101, bins // 8)))
        dw = int(rng.integers(1, max(2, min(bins // 4, 9000))))
        lo = avg // 2
        hi = bins - dw - avg
        if hi > lo:
            ld = int(rng.integers(lo, hi))
            nw = int(rng.integers(1, max(2, min(bins // 4, 9000))))
            ln = int(rng.integers(0, bins - nw + 1))
            bands = (ln, nw, ld, dw, avg)
    tile = None
    if rng.random() < 0.5:
        cols = int(rng.integers(1, min(bins, 4096) + 1))
        tile = (int(rng.integers(0, bins - cols + 1)), cols)
    stride_extra = int(rng.choice([0, 0, 1, 3, 64, 1000]))
    base_off = int(rng.choice([0, 0, 1, 5, 32]))
    spare = int(rng.choice([0, 0, 0, 1, 3]))
    spectra = bool(pow2 and precision == 0 and rng.random() < 0.25)
    ln = bool(tile is not None and rng.random() < 0.35)      # tile_ln: the tile's natural log and its min / max per row
    return dict(ln=ln, bins=bins, overlap=overlap, total=total, first=first, rows=rows, tail=tail, fmt=fmt, gain=gain,
                window=window, precision=precision, bands=bands, tile=tile, stride_extra=stride_extra,
                base_off=base_off, spare=spare, spectra=spectra, data_seed=int(rng.integers(0, 2 ** 31)))


def make_input(c):
    rng = np.random.default_rng(c["data_seed"])
    bins = c["bins"]
    hop = bins - min(max(c["overlap"], 0), bins - 1)
    samples = bins + (c["total"] - 1) * hop + c["tail"]
    if c["fmt"] == "i16":
        iq = rng.integers(-20000, 20000, size=(samples, 2), dtype=np.int16)
    else:
        iq = rng.standard_normal((samples, 2)).astype(np.float32)
        t = np.arange(samples, dtype=np.float64)
        f = float(rng.uniform(-0.45, 0.45))
        amp = float(rng.choice([0.0, 5.0, 200.0]))
        iq[:, 0] += (amp * np.cos(2 * np.pi * f * t)).astype(np.float32)
        iq[:, 1] += (amp * np.sin(2 * np.pi * f * t)).astype(np.float32)
    w = None
    if c["window"] == "custom":
        w = (0.25 + rng.random(bins)).astype(np.float32)
    return iq, w


def run_case(ro, oracle, torch, c):
    """raises AssertionError with the case in the message; returns the rows' error for the record"""
    bins, first, rows = c["bins"], c["first"], c["rows"]
    eff = ro.clamp_overlap(bins, c["overlap"])
    iq, w = make_input(c)
    samples = iq.shape[0]
    assert ro.row_count(samples, bins, eff) == c["total"], c
    kw = dict(bins=bins, overlap=c["overlap"], iq_gain=c["gain"], precision=c["precision"],
              spare_cus_per_xcd=c["spare"])
    if c["window"] == "hann":
        kw["window"] = ro.RO_WINDOW_HANN
    elif c["window"] == "custom":
        kw["window_table"] = w
    if c["bands"]:
        ln, nw, ld, dw, avg = c["bands"]
        kw["bands"] = ro.Bands(low_noise=ln, noise_width=nw, low_detect=ld, detect_width=dw, avg_bins=avg)
    if c["tile"]:
        kw["tile"] = c["tile"]
    fmt = ro.RO_IQ_I16 if c["fmt"] == "i16" else ro.RO_IQ_F32
    # every other FP64-mode case of the register kernel's sizes (256 ... 65536) hands over struct Complex's doubles themselves
    # (src/Backend.h:26-29), with bits no float32 holds (derived from the case, no extra draw: the seeds' cases stay)
    if c["precision"] == 1 and c["fmt"] == "f32" and 256 <= bins <= 65536 and (c["data_seed"] & 1):
        iq = iq.astype(np.float64) * (1.0 + 2.0 ** -29) + 2.0 ** -31
        fmt = ro.RO_IQ_F64
    stride = bins + c["stride_extra"]
    guard = 64
    d_iq = torch.from_numpy(iq).cuda()
    buf = torch.full((c["base_off"] + rows * stride + guard,), float("nan"), dtype=torch.float32, device="cuda")
    d_rows = buf[c["base_off"]:]
    d_tile = d_recs = None
    if c["tile"]:
        d_tile = torch.full((rows * c["tile"][1] + guard,), float("nan"), dtype=torch.float32, device="cuda")
    if c["bands"]:
        d_recs = torch.full((rows * 3 + guard,), float("nan"), dtype=torch.float32, device="cuda")
    spec = d_ln = d_mm = None
    if c.get("ln"):
        kw["tile_ln"] = True
        d_ln = torch.full((rows * c["tile"][1] + guard,), float("nan"), dtype=torch.float32, device="cuda")
        d_mm = torch.full((rows * 2 + guard,), float("nan"), dtype=torch.float32, device="cuda")
    with ro.Stft(**kw) as st:
        assert st.hop == bins - eff, c
        if c.get("ln"):
            st.run_resident_ln(d_iq, fmt, samples, first, rows, d_rows, d_tile, d_ln=d_ln, d_minmax=d_mm,
                               row_stride=stride, d_records=d_recs, stream=torch.cuda.current_stream().cuda_stream)
        else:
            st.run_resident(d_iq, fmt, samples, first, rows, d_rows, row_stride=stride, d_tile=d_tile,
                            d_records=d_recs, stream=torch.cuda.current_stream().cuda_stream)
        # (a quarter of the FP64-mode cases of the register kernel's sizes ask for the complex rows too -- derived from the
        # case, no extra draw)
        spectra64 = c["precision"] == 1 and 256 <= bins <= 65536 and ((c["data_seed"] >> 1) & 3) == 0
        if c["spectra"] or spectra64:
            spec = torch.full((rows, bins, 2), float("nan"), dtype=torch.float32, device="cuda")
            st.spectra_resident(d_iq, fmt, samples, first, rows, spec,
                                stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    host = buf.cpu().numpy()
    assert np.isnan(host[:c["base_off"]]).all(), ("wrote in front of the row buffer", c)
    body = host[c["base_off"]:c["base_off"] + rows * stride].reshape(rows, stride)
    got = body[:, :bins]
    assert np.isfinite(got).all(), ("non-finite row values", c)
    assert np.isnan(body[:, bins:]).all(), ("wrote into the stride padding", c)
    assert np.isnan(host[c["base_off"] + rows * stride:]).all(), ("wrote behind the row buffer", c)
    ow = w if w is not None else oracle.window(bins, "hann" if c["window"] == "hann" else "nuttall")
    src = iq.astype(np.float32) if c["fmt"] == "i16" else iq
    want = oracle.stft(src, bins, eff, w=ow, gain=c["gain"], first_row=first, max_rows=rows)
    assert want.shape == got.shape, (want.shape, got.shape, c)
    err = np.abs(got.astype(np.float64) - want).max(axis=1) / np.maximum(np.abs(want).max(axis=1), 1e-300)
    tol = TOL_F64 if c["precision"] else TOL_F32
    assert err.max() <= tol, ("rows differ from the oracle: %.3g > %.3g" % (err.max(), tol), c)
    if c["precision"]:                                       # the FP64 mode's other bar: every bin of its own (one float32 ulp)
        w64 = want.astype(np.float64)
        big = w64 > 1e-30
        per_bin = (np.abs(got.astype(np.float64) - w64)[big] / w64[big]).max() if big.any() else 0.0
        assert per_bin <= TOL_F64_PER_BIN, ("FP64 mode beyond its per-bin bar: %.3g" % per_bin, c)
    if c["tile"]:
        t = d_tile.cpu().numpy()
        f, n = c["tile"]
        assert np.array_equal(t[:rows * n].reshape(rows, n), got[:, f:f + n]), ("tile differs from its rows", c)
        assert np.isnan(t[rows * n:]).all(), ("wrote behind the tile", c)
        if c.get("ln"):
            # tests/test_gpu_ln_tile.py's bars: logf within 2 ulp of libm's, the rows' min / max of it exact
            image = np.ascontiguousarray(got[:, f:f + n])
            lnv = d_ln.cpu().numpy()
            mm = d_mm.cpu().numpy()
            assert np.isnan(lnv[rows * n:]).all() and np.isnan(mm[rows * 2:]).all(), ("wrote behind the ln tile", c)
            lnv, mm = lnv[:rows * n].reshape(rows, n), mm[:rows * 2].reshape(rows, 2)
            want_ln = oracle.ln_levels(image)[0]
            nz = image != 0
            tol = 2 * np.spacing(np.maximum(np.abs(want_ln[nz]), np.float32(1.0)).astype(np.float32))
            assert np.all(np.abs(lnv[nz] - want_ln[nz]) <= tol), ("ln tile beyond 2 ulp", c)
            if nz.all():
                assert np.array_equal(mm[:, 0], lnv.min(axis=1)) and np.array_equal(mm[:, 1], lnv.max(axis=1)), ("ln min / max", c)
    if c["bands"]:
        r = d_recs.cpu().numpy()
        recs = r[:rows * 3].copy().view(ro.capi.SCAN_DTYPE).reshape(-1)
        ln, nw, ld, dw, avg = c["bands"]
        n_, p_, a_ = oracle.scan_rows(np.ascontiguousarray(got), ln, nw, ld, dw, avg)
        assert np.array_equal(recs["peak"], p_), ("scan peak", c)
        assert np.array_equal(recs["noise"].view(np.uint32), n_.view(np.uint32)), ("scan noise", c)
        assert np.array_equal(recs["average"].view(np.uint32), a_.view(np.uint32)), ("scan average", c)
        assert np.isnan(r[rows * 3:]).all(), ("wrote behind the records", c)
    if spec is not None:
        s = spec.cpu().numpy().astype(np.float64)
        s = s[..., 0] + 1j * s[..., 1]
        hop = bins - eff
        flat = src.astype(np.float64)
        for i in (0, rows - 1):
            seg = flat[(first + i) * hop:(first + i) * hop + bins]
            _, ws = oracle.row_with_spectrum(seg[:, 0] + 1j * seg[:, 1], ow, gain=c["gain"])
            e = np.abs(s[i] - ws).max() / max(np.abs(ws).max(), 1e-300)
            assert e <= TOL_SPEC, ("spectra differ from the oracle: %.3g" % e, c)
            if c["precision"]:                               # the double transform narrowed once: component by component
                scale = np.maximum(np.abs(ws.real), np.abs(ws.imag))
                cerr = np.maximum(np.abs(s[i].real - ws.real), np.abs(s[i].imag - ws.imag))
                big = scale > 1e-30
                cworst = (cerr[big] / scale[big]).max() if big.any() else 0.0
                assert cworst <= TOL_F64_PER_BIN, ("FP64-mode spectra beyond a float32 ulp: %.3g" % cworst, c)
    return float(err.max())



# ---- the call-by-call form: ro_stft_push / flush / fetch (and the row sink) -------------------------------------

def draw_stream_case(rng):
    """a resident configuration plus how the same stream arrives call by call"""
    c = draw_case(rng)
    bins = c["bins"]
    hop = bins - min(max(c["overlap"], 0), bins - 1)
    budget = 1 << 22 if bins <= 32768 else 3 << 20
    c["total"] = int(rng.integers(1, max(2, min(70, budget // bins) + 1)))
    c["first"], c["rows"], c["stride_extra"], c["base_off"], c["spectra"] = 0, c["total"], 0, 0, False
    c["batch"] = int(rng.choice([0, 1, 2, 3, 5, 8], p=[0.08, 0.12, 0.2, 0.2, 0.2, 0.2]))    # 0: the ABI's default (64 MiB of rows)
    c["push_fmt"] = c["fmt"] if c["fmt"] == "i16" else str(rng.choice(["f32", "f64", "c64", "c128"]))
    c["sink"] = bool(c["batch"] > 0 and rng.random() < 0.5)
    if c["sink"]:
        c["slots"] = 2 * c["batch"] + int(rng.integers(0, 2 * c["batch"] + 3))
        c["first_slot"] = int(rng.integers(0, c["slots"]))
        c["ring_pad"] = int(rng.choice([0, 3]))
    span = max(c["batch"], 1) * hop
    c["piece"] = int(rng.choice([max(1, hop // 3), hop, hop + 1, 3 * hop + 7, 4096, span + 7, 4 * span + bins]))
    c["piece"] = max(c["piece"], (bins + c["total"] * hop) // 1500)        # at most a few thousand calls per case
    c["fetch_p"] = float(rng.choice([0.0, 0.3, 1.0]))
    return c


def run_stream_case(ro, oracle, torch, c):
    """the streamed outputs equal the resident call's bit for bit (which run_case holds against the oracle), every row
    once and in order; with a sink, in its slot; refusals leave nothing consumed"""
    bins = c["bins"]
    eff = ro.clamp_overlap(bins, c["overlap"])
    hop = bins - eff
    iq, w = make_input(c)
    samples, total = iq.shape[0], c["total"]
    kw = dict(bins=bins, overlap=c["overlap"], iq_gain=c["gain"], precision=c["precision"],
              spare_cus_per_xcd=c["spare"])
    if c["window"] == "hann":
        kw["window"] = ro.RO_WINDOW_HANN
    elif c["window"] == "custom":
        kw["window_table"] = w
    if c["bands"]:
        ln, nw, ld, dw, avg = c["bands"]
        kw["bands"] = ro.Bands(low_noise=ln, noise_width=nw, low_detect=ld, detect_width=dw, avg_bins=avg)
    if c["tile"]:
        kw["tile"] = c["tile"]
    # the resident call on the same samples: what the stream has to reproduce
    d_iq = torch.from_numpy(iq).cuda()
    d_rows = torch.empty((total, bins), dtype=torch.float32, device="cuda")
    d_recs = torch.zeros((total, 3), dtype=torch.float32, device="cuda") if c["bands"] else None
    with ro.Stft(**kw) as st:
        st.run_resident(d_iq, ro.RO_IQ_I16 if c["fmt"] == "i16" else ro.RO_IQ_F32, samples, 0, total, d_rows,
                        d_records=d_recs, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    ref_rows = d_rows.cpu().numpy()
    ref = ref_rows[:, c["tile"][0]:c["tile"][0] + c["tile"][1]] if c["tile"] else ref_rows
    ref_recs = d_recs.cpu().numpy().view(ro.capi.SCAN_DTYPE).reshape(-1) if c["bands"] else None
    del d_iq, d_rows, d_recs
    # one spot check of the reference itself against the oracle (run_case does this exhaustively)
    ow = w if w is not None else oracle.window(bins, "hann" if c["window"] == "hann" else "nuttall")
    r = total - 1
    src = iq[r * hop:r * hop + bins].astype(np.float32)
    want = oracle.stft(src, bins, 0, w=ow, gain=c["gain"])[0]
    err = float(np.abs(ref_rows[r].astype(np.float64) - want).max() / max(np.abs(want).max(), 1e-300))
    assert err <= (TOL_F64 if c["precision"] else TOL_F32), ("resident row differs from the oracle: %.3g" % err, c)
    del ref_rows

    if c["push_fmt"] == "f64":
        feed = iq.astype(np.float64)
    elif c["push_fmt"] == "c64":
        feed = (iq[:, 0] + 1j * iq[:, 1]).astype(np.complex64)
    elif c["push_fmt"] == "c128":
        feed = iq[:, 0].astype(np.float64) + 1j * iq[:, 1]
    else:
        feed = iq
    cols = ref.shape[1]
    pinned = ring = None
    seen = 0
    prng = np.random.default_rng(c["data_seed"] ^ 0x5EED)

    def check_rows(first, rows, recs):
        assert first == seen, ("rows out of order: first %d, expected %d" % (first, seen), c)
        n = rows.shape[0] if rows is not None else len(recs)
        if rows is not None:
            assert np.array_equal(rows.view(np.uint32), ref[first:first + n].view(np.uint32)), ("streamed rows differ from the resident call", first, c)
        if ref_recs is not None:
            for k in ("noise", "peak", "average"):
                assert np.array_equal(recs[k].view(np.uint32), ref_recs[k][first:first + n].view(np.uint32)), ("streamed record " + k, first, c)

    with ro.Stft(max_batch_rows=c["batch"], **kw) as st:
        if c["sink"]:
            pinned = ro.PinnedArray(c["slots"], cols + c["ring_pad"])
            ring = pinned.array
            ring[:] = np.nan
            st.set_row_sink(ring, c["first_slot"])

        def take(limit):
            nonlocal seen
            while limit > 0:
                ask = min(limit, 1 + int(prng.integers(0, 2 * max(c["batch"], 1) + 40)))
                if c["sink"]:
                    first, got, recs = st.fetch_records(ask)
                    if got == 0:
                        return
                    check_rows(first, None, recs if recs is not None else [0] * got)
                    for r_ in range(first, first + got):
                        slot = ring[(c["first_slot"] + r_) % c["slots"]]
                        assert np.array_equal(slot[:cols].view(np.uint32), ref[r_].view(np.uint32)), ("sink slot of row %d" % r_, c)
                        assert np.isnan(slot[cols:]).all(), ("sink padding written", c)
                else:
                    first, rows, recs = st.fetch(ask)
                    got = rows.shape[0]
                    if got == 0:
                        return
                    check_rows(first, rows, recs)
                seen += got
                limit -= got

        at = 0
        while at < samples:
            n = int(prng.integers(1, c["piece"] + 1))
            piece = feed[at:at + n]
            try:
                st.push(piece)
            except ro.StftError as e:
                assert c["sink"] and e.code == -5 and "nothing was consumed" in str(e), (str(e), c)
                take(10 ** 9)
                room = (c["slots"] - c["batch"]) * hop
                if len(piece) > room:
                    piece = piece[:room]
                st.push(piece)
            at += len(piece)
            if prng.random() < c["fetch_p"]:
                take(st.rows_complete() if prng.random() < 0.5 else int(prng.integers(1, 50)))
        for _ in range(200):
            try:
                st.flush()
                break
            except ro.StftError as e:
                assert c["sink"] and e.code == -5 and "row sink full" in str(e), (str(e), c)
                take(10 ** 9)
        else:
            raise AssertionError(("flush never accepted", c))
        take(10 ** 9)
        stats = st.stats()
        assert seen == total == stats["rows_out"] and stats["samples_in"] == samples, (seen, total, stats, c)
        if c["sink"]:
            st.reset()
            st.set_row_sink(None)
    if pinned is not None:
        del ring
        pinned.close()
    return err


def fuzz(ro, oracle, torch, seed, seconds=None, cases=None, only=None, log=None, kind="resident"):
    """run cases of `seed` (kind: "resident" or "stream") until `seconds` are over or `cases` are done; returns
    (cases run, worst row error by precision mode)"""
    draw, run = (draw_stream_case, run_stream_case) if kind == "stream" else (draw_case, run_case)
    rng = np.random.default_rng(seed)
    t0, k, worst = time.time(), 0, {0: 0.0, 1: 0.0}
    while True:
        if cases is not None and k >= cases:
            break
        if seconds is not None and time.time() - t0 > seconds:
            break
        c = draw(rng)
        if only is None or only == k:
            t1 = time.time()
            e = run(ro, oracle, torch, c)
            worst[c["precision"]] = max(worst[c["precision"]], e)
            if log:
                log("%s seed %d case %d: %.1f s, err %.3g  %s" % (kind, seed, k, time.time() - t1, e, c))
        k += 1
        if only is not None and k > only:
            break
    return k, worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--cases", type=int, default=None)
    ap.add_argument("--case", type=int, default=None, help="replay this case of the seed only")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--kind", choices=["resident", "stream"], default="resident")
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    ro = importlib.import_module("radio-observer_amd")
    import ro_oracle
    ro_oracle.lib()
    last = [time.time()]

    def quiet_log(line):                      # (a heartbeat every half minute: a silent job on a GPU box is taken for hung)
        if time.time() - last[0] > 30:
            last[0] = time.time()
            print(line[:100] + " ...", flush=True)

    log = quiet_log if a.quiet else (lambda s: print(s, flush=True))
    n, worst = fuzz(ro, ro_oracle, torch, a.seed, seconds=None if a.case is not None else a.seconds,
                    cases=a.cases, only=a.case, log=log, kind=a.kind)
    print("fuzz_parity (%s): seed %d, %d cases, all outputs within their bars; worst row error f32 %.3g, f64 %.3g"
          % (a.kind, a.seed, n, worst[0], worst[1]), flush=True)


if __name__ == "__main__":
    main()
