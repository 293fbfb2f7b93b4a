"""GPU: bench.py itself, run in this process (a child process would have to be started from a process that already
holds the GPU): the line carries what the contract asks for, and --workload c5's hash of the stitched band + records
is the hash of the same stream transformed directly through the C ABI."""
import contextlib
import hashlib
import io
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(argv):
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            bench.main()
    finally:
        sys.argv = old
        bench.BINS, bench.OVERLAP, bench.HOP = 32768, 24576, 8192
        bench.ALG_BYTES_PER_ROW = bench.HOP * 8 + bench.BINS * 4
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1, buf.getvalue()[-2000:]
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields(ro, torch_cuda):
    d = run_bench(["--steps", "4", "--warmup", "1", "--prewarm", "2", "--rows", "4096", "--cpu-seconds", "1.5",
                   "--stream-seconds", "0.4"])
    assert d["unit"] == "rows/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["prewarm"] == 2
    assert d["dtype"] == "f32" and d["vs_baseline"] is None and d["scaling"] == "weak"
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["kernel"] == "stft32k_kernel"
    assert rf["achieved"] == pytest.approx(rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9)
    assert rf["frac"] == pytest.approx(rf["achieved"] / 8000.0)
    # the kernel's share cannot be below what the whole step makes of the same bytes, nor far from the post-hoc loop
    assert 0 < rf["frac_step"] <= rf["frac"] * 1.001
    assert abs(rf["frac_posthoc"] - rf["frac"]) < 0.35 * rf["frac"]      # (a cold device: a dozen launches in all)
    assert "timed region" in rf["kernel_ms_source"]
    assert d["parity"]["max_err_rel_to_row_max"] <= 1e-5 and d["parity"]["scan_records_bit_exact"]
    sp = d["strict_precision"]
    assert sp["dtype"] == "f64" and sp["parity"]["max_err_per_bin_relative"] <= 1e-5
    assert sp["roofline"]["frac"] == pytest.approx(sp["roofline"]["achieved"] / 8000.0) and "f64r_kernel" in sp["roofline"]["kernel"]
    assert sp["parity"]["max_err_per_bin_relative"] <= 2e-7 and "4096" in sp["c2"]["workload"]
    assert sp["c2"]["dtype"] == "f64" and sp["c2"]["parity"]["max_err_per_bin_relative"] <= 2e-7 and sp["c2"]["value"] > sp["value"]
    assert "1024" in sp["c1"]["workload"] and sp["c1"]["parity"]["max_err_per_bin_relative"] <= 2e-7 and sp["c1"]["value"] > sp["c2"]["value"]
    assert "f64r_kernel" in sp["c1"]["roofline"]["kernel"] and 0 < sp["roofline"]["fp64_valu"]["frac"] < 1
    assert rf["limiter"].startswith("package power cap")
    io = d["ionozor"]                   # the four-step form at Ionozor.json:27-28, never the headline
    assert io["unit"] == "rows/s" and io["value"] > 0 and io["parity"]["max_err_rel_to_row_max"] <= 1e-5
    assert io["roofline"]["frac"] == pytest.approx(io["roofline"]["achieved"] / 8000.0) and "four_cols_kernel" in io["roofline"]["kernel"]
    bo = d["bolidozor"]                 # Bolidozor.json:45-46 on the one-kernel large form
    assert bo["value"] > io["value"] and bo["parity"]["max_err_rel_to_row_max"] <= 1e-5 and "65536" in bo["workload"]
    assert bo["f64"]["dtype"] == "f64" and bo["f64"]["parity"]["max_err_per_bin_relative"] <= 2e-7 and 0 < bo["f64"]["value"] < bo["value"]
    for k in ("streaming", "streaming_batch256"):
        st = d[k]
        assert "error" not in st, st
        assert st["rows"] > 0 and st["samples_per_call"] == 4096 and st["real_time_factor"] > 10
        assert st["value"] == pytest.approx(st["rows"] / st["seconds"])
    assert len(d["config"]["rank0_scan_records_hash"]) == 32          # what an N-rank run's line is compared with
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert "cpu_baseline_O0" in d and "error" not in d["cpu_baseline_O0"]


@pytest.mark.parametrize("exchange,pattern", [("torch", "nccl"), ("capi", "nccl"), ("capi", "direct")])
def test_bench_c5_hash_is_the_hash_of_the_stream(ro, torch_cuda, exchange, pattern):
    torch = torch_cuda
    seconds = 900.0
    d = run_bench(["--workload", "c5", "--c5-seconds", str(seconds), "--exchange", exchange, "--pattern", pattern, "--steps", "2", "--warmup", "1",
                   "--prewarm", "1", "--no-cpu-baseline", "--no-strict", "--no-parity", "--no-streaming"])
    assert d["scaling"] == "strong"
    got = d["config"]["c5_hash_of_stitched_band_and_records"]
    # the same stream straight through the C ABI
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    import util
    bins, overlap, fs = 32768, 24576, 48000
    total = int(seconds * fs)
    R = ro.row_count(total, bins, overlap)
    assert R == d["config"]["rows_per_step_per_gpu"]
    iq = util.c5_slice(torch, 0, (R - 1) * (bins - overlap) + bins, total_samples=total)
    lo = ro.frequency_to_bin(bins, fs, bench.JSON_SNAPSHOT[0])
    hi = ro.frequency_to_bin(bins, fs, bench.JSON_SNAPSHOT[1])
    rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
    tile = torch.empty((R, hi - lo), dtype=torch.float32, device="cuda")
    recs = torch.zeros((R, 3), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap, bands=bench.make_bands(ro), tile=(lo, hi - lo)) as st:
        st.run_resident(iq, ro.RO_IQ_F32, iq.shape[0], 0, R, rows, d_tile=tile, d_records=recs,
                        stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    hh = hashlib.sha256()
    hh.update(tile.cpu().numpy().tobytes())
    hh.update(recs.cpu().numpy().tobytes())
    assert got == hh.hexdigest()[:32]
    assert np.array_equal(tile.cpu().numpy(), rows[:, lo:hi].cpu().numpy())
