"""A short fixed-seed slice of tests/fuzz_parity.py in the GPU suite: whole random configurations of the resident entry
points (size, overlap, row range, format, gain, window, precision, bands, tile, stride, base offset) against the oracle.
RO_FUZZ_CASES / RO_FUZZ_SECONDS / RO_FUZZ_SEEDS (comma-separated) widen it; profiles/r06_fuzz.txt (round 5: r05_fuzz.txt) holds a long run."""
import os

import pytest

import fuzz_parity

SEEDS = [int(s) for s in os.environ.get("RO_FUZZ_SEEDS", "1,2").split(",") if s.strip()]
# a FIXED number of cases per seed by default (the same draws every run: what the suite checks does not depend on the
# box's speed); RO_FUZZ_SECONDS switches to a time budget
SECONDS = float(os.environ["RO_FUZZ_SECONDS"]) if "RO_FUZZ_SECONDS" in os.environ else None
CASES = None if SECONDS is not None else int(os.environ.get("RO_FUZZ_CASES", "60"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_random_configurations_match_oracle(ro, oracle, torch_cuda, seed):
    n, worst = fuzz_parity.fuzz(ro, oracle, torch_cuda, seed, seconds=SECONDS, cases=CASES)
    print("seed %d: %d cases, worst row error f32 %.3g / f64 %.3g" % (seed, n, worst[0], worst[1]))
    assert n >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_random_streams_equal_the_resident_call(ro, oracle, torch_cuda, seed):
    """the same draws delivered call by call (push / flush / fetch, with and without a row sink): bit for bit the
    resident call's rows, tiles and records, every row once and in order"""
    n, _ = fuzz_parity.fuzz(ro, oracle, torch_cuda, 100 + seed, seconds=SECONDS, cases=CASES, kind="stream")
    print("seed %d: %d streams" % (100 + seed, n))
    assert n >= 1


def test_case_generator_covers_the_edges():
    """the draws themselves (no device): every size class, the overlap edges, both formats, both precisions,
    bands that keep average()'s window inside the row"""
    import numpy as np
    rng = np.random.default_rng(0)
    cases = [fuzz_parity.draw_case(rng) for _ in range(3000)]
    bins = {c["bins"] for c in cases}
    assert {256, 32768, 65536, 524288, 1048576} <= bins
    assert any(b & (b - 1) for b in bins) and all(b % 2 == 0 for b in bins)
    assert any(c["overlap"] == 0 for c in cases) and any(c["overlap"] == c["bins"] - 1 for c in cases)
    assert any(c["overlap"] >= c["bins"] for c in cases)
    assert {c["fmt"] for c in cases} == {"f32", "i16"} and {c["precision"] for c in cases} == {0, 1}
    for c in cases:
        assert 0 <= c["first"] < c["total"] and 1 <= c["rows"] <= c["total"] - c["first"]
        if c["precision"]:
            assert c["bins"] & (c["bins"] - 1) == 0
        if c["bands"]:
            ln, nw, ld, dw, avg = c["bands"]
            assert ld - avg // 2 >= 0 and ld + dw - 1 - avg // 2 + avg <= c["bins"]
            assert 0 <= ln and ln + nw <= c["bins"]
        if c["tile"]:
            assert c["tile"][0] + c["tile"][1] <= c["bins"]
    streams = [fuzz_parity.draw_stream_case(rng) for _ in range(2000)]
    assert {c["push_fmt"] for c in streams} == {"f32", "f64", "c64", "c128", "i16"}
    assert {c["batch"] for c in streams} == {0, 1, 2, 3, 5, 8} and any(c["sink"] for c in streams)
    for c in streams:
        assert c["first"] == 0 and c["rows"] == c["total"] and c["precision"] in (0, 1)
        if c["sink"]:
            assert c["batch"] > 0 and c["slots"] >= 2 * c["batch"] and 0 <= c["first_slot"] < c["slots"]
