"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/ro_stft.h declares, its pure-host helpers agree with the oracle, and every compute
entry point fails loudly (RO_ERR_HIP) when there is no GPU -- there is no CPU fallback."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ro_stft.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ro_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(ro):
    lib = ro.library()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libro_stft.so does not export %s" % n
    # and the binding covers exactly the header
    assert sorted(ro.capi.exported_symbols()) == names
    assert lib.ro_abi_version() == 5
    # nothing undeclared: every ro_* the library exports is in the header (diagnostic hooks live in -DRO_DIAG=1 builds only)
    out = subprocess.run(["nm", "-D", "--defined-only", ro.capi.library_path()], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r" T (ro_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == names, sorted(set(exported) ^ set(names))


def test_struct_layouts(ro):
    assert C.sizeof(ro.ScanRecord) == 12                   # ro_scan_record_t
    assert C.sizeof(ro.Bands) == 20
    assert ro.capi.SCAN_DTYPE.itemsize == 12


def test_host_helpers_match_oracle(ro, oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    for bins in (256, 1024, 4096, 32768, 65536):
        for sr in (48000, 96000, 44100):
            for f in np.concatenate([rng.uniform(-sr, sr, 200), [0, 40, 9000, 10300, 10900, 12000, sr / 2, -sr / 2]]):
                f32 = float(np.float32(f))
                assert ro.frequency_to_bin(bins, sr, f32) == L.ro_oracle_frequency_to_bin(bins, sr, f32)
            for b in rng.integers(0, bins, 100):
                assert ro.bin_to_frequency(bins, sr, int(b)) == L.ro_oracle_bin_to_frequency(bins, sr, int(b))
            for ov in (0, bins // 2, bins - 1, bins + 7, -5):
                assert ro.clamp_overlap(bins, ov) == L.ro_oracle_clamp_overlap(bins, ov)
                assert ro.fft_sample_rate(sr, bins, ov) == L.ro_oracle_fft_sample_rate(sr, bins, ov)
                for T in (0, bins - 1, bins, 5 * bins + 3, 10**9):
                    assert ro.row_count(T, bins, ov) == oracle.row_count(T, bins, ov)
    for t in (0.0, 1.0, 2.0, 5.0, 3600.0, 0.17):
        for rate in (5.859375, 93.75, 23.4375):
            assert ro.time_to_fft_samples(t, rate) == L.ro_oracle_time_to_fft_samples(t, rate)


@pytest.mark.parametrize("bins", [256, 1024, 4096, 32768])
def test_window_tables_bit_identical_to_oracle(ro, oracle, bins):
    assert np.array_equal(ro.window_table(ro.RO_WINDOW_NUTTALL, bins), oracle.window(bins, "nuttall"))
    assert np.array_equal(ro.window_table(ro.RO_WINDOW_HANN, bins), oracle.window(bins, "hann"))
    with pytest.raises(ro.StftError):
        ro.window_table(ro.RO_WINDOW_CUSTOM, bins)


def test_supported_sizes(ro):
    for b in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576):
        assert ro.bins_supported(b)      # 65536 / 524288: Bolidozor.json:45, Ionozor.json:27
    for b in (258, 1000, 32728, 100000, 524286):   # any other even length: chirp-z (src/BolidRecorder.h:35 suggests 32728)
        assert ro.bins_supported(b)
    # odd lengths: the reference's processFFT leaves a column unwritten (src/WaterfallBackend.cpp:489-505)
    for b in (0, 100, 128, 255, 254, 1001, 32727, 524290, 2097152):
        assert not ro.bins_supported(b)


def test_config_is_validated_before_any_device_work(ro):
    """Argument errors come back as RO_ERR_INVALID / RO_ERR_UNSUPPORTED with or without a GPU."""
    for kw, code in ((dict(bins=1001), -2), (dict(bins=1000, precision=1), -2), (dict(bins=1024, iq_phase_shift=1), -2),
                     (dict(bins=1024, sample_rate=0), -1), (dict(bins=1024, spare_cus_per_xcd=17), -1),
                     (dict(bins=1024, spare_cus_per_xcd=-1), -1), (dict(bins=1024, precision=3), -1),
                     (dict(bins=1000, precision=2), -2), (dict(bins=32768, precision=2), -2)):   # 2: the retired one-launch form
        with pytest.raises(ro.StftError) as e:
            ro.Stft(**kw)
        assert e.value.code == code, (kw, str(e.value))
    # the compiled struct and the ctypes mirror agree (ro_stft_create rejects a wrong struct_size)
    cfg = ro.capi.Config()
    cfg.struct_size = C.sizeof(ro.capi.Config) - 4
    cfg.bins = 1024
    h = C.c_void_p()
    assert ro.library().ro_stft_create(C.byref(cfg), C.byref(h)) == -1
    # an ABI-1 caller (struct without `precision`) is accepted up to the point where a device is needed
    cfg.struct_size = ro.capi.Config.precision.offset
    cfg.sample_rate = 48000
    rc = ro.library().ro_stft_create(C.byref(cfg), C.byref(h))
    assert rc in (0, -3), rc                                 # RO_OK on a GPU box, RO_ERR_HIP here
    if rc == 0:
        ro.library().ro_stft_destroy(h)


def test_direct_schedule_is_a_perfect_exchange(ro):
    """ro_direct_schedule (the one schedule ro_allgather_rows_direct, ro_gather_rows and timeshard.gather_rows_direct walk):
    for every world and row count, every rank receives every other rank's rows exactly once at their stitched place, every
    send of a step is matched by a receive of the SAME step, and in every step every rank sends once and receives once."""
    for world in (1, 2, 3, 7, 8):
        for total in (0, 7, 168747):
            shards = [ro.capi.shard_rows(total, world, g) for g in range(world)]
            assert sum(n for _, n in shards) == total
            got = [dict() for _ in range(world)]                 # rank -> {first_row: rows} it receives
            for k in range(0 if world == 1 else 1, world):
                step = [ro.capi.direct_schedule(world, r, total, k) for r in range(world)]
                assert sorted(s[0] for s in step) == list(range(world))       # every rank is sent to once
                assert sorted(s[1] for s in step) == list(range(world))       # ... and sent from once
                for r, (to, frm, f, n) in enumerate(step):
                    assert to == (r + k) % world and frm == (r - k) % world
                    assert step[frm][0] == r                                   # my source sends to me in this very step
                    assert (f, n) == shards[frm]                               # ... and what arrives is its shard, in place
                    if k > 0 and n > 0:
                        assert f not in got[r]                                 # ... once
                        got[r][f] = n
            for r in range(world):
                want = {f: n for g, (f, n) in enumerate(shards) if g != r}
                if world > 1:
                    assert {f: n for f, n in got[r].items() if n} == {f: n for f, n in want.items() if n}
            # the gather to one rank: the root receives every step, rank r sends in the step whose `to` is the root
            for root in range(world):
                for r in range(world):
                    if r != root:
                        ks = [k for k in range(1, world) if ro.capi.direct_schedule(world, r, total, k)[0] == root]
                        assert ks == [(root - r) % world]
                        assert ro.capi.direct_schedule(world, root, total, ks[0])[1] == r
    lib = ro.library()
    z = C.c_int()
    z64 = C.c_int64()
    assert lib.ro_direct_schedule(4, 4, 10, 1, C.byref(z), C.byref(z), C.byref(z64), C.byref(z64)) == -1
    assert lib.ro_direct_schedule(4, 0, 10, 4, C.byref(z), C.byref(z), C.byref(z64), C.byref(z64)) == -1
    assert lib.ro_direct_schedule(4, 0, 10, 1, None, C.byref(z), C.byref(z64), C.byref(z64)) == -1


def test_no_cpu_fallback(ro):
    """Without a HIP device the product must refuse, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ro.StftError) as e:
        ro.Stft(bins=1024, overlap=512)
    assert e.value.code == -3                                # RO_ERR_HIP
    assert "device" in str(e.value).lower() or "hip" in str(e.value).lower()


def test_row_sink_entry_points_without_a_device(ro):
    """ABI 3: the pinned allocator gives nothing without a HIP device (the host mirror's ring then lives on the heap and
    the rows take the copying path), and the sink wants a handle."""
    import torch
    lib = ro.library()
    assert lib.ro_stft_set_row_sink(None, None, 0, 0, 0) == -1           # RO_ERR_INVALID: null handle
    lib.ro_pinned_free(None)                                              # like free(NULL)
    if not torch.cuda.is_available():
        assert not lib.ro_pinned_alloc(0, 4096)
        with pytest.raises(MemoryError):
            ro.PinnedArray(4, 4)


def test_rows_complete_wants_a_handle(ro):
    import ctypes as C2
    n = C2.c_int64(7)
    assert ro.library().ro_stft_rows_complete(None, C2.byref(n)) == -1 and n.value == 7


def test_pinned_check_refuses_heap_memory(ro):
    """ro_pinned_check is what ro_stft_set_row_sink asks before it lets a DMA near the caller's ring: heap memory, NULL
    and empty ranges are not page-locked memory of this process's HIP runtime (with no device nothing is)."""
    import numpy as np
    lib = ro.library()
    heap = np.zeros(1 << 20, np.float32)
    assert lib.ro_pinned_check(heap.ctypes.data, heap.nbytes) == 0
    assert lib.ro_pinned_check(heap.ctypes.data, 4) == 0
    assert lib.ro_pinned_check(None, 4096) == 0 and lib.ro_pinned_check(heap.ctypes.data, 0) == 0


def test_two_hip_runtimes_are_refused():
    """Import order decides how many HIP runtimes a process holds (capi.hip_runtimes): torch first -> libro_stft.so
    reuses torch's copy through the SONAME; the product first -> torch maps its own second copy, and handles are
    refused from then on instead of mixing pointers of two runtimes."""
    code = ("import importlib, sys; sys.path.insert(0, %r)\n"
            "%s\n"
            "ro = importlib.import_module('radio-observer_amd')\n"
            "ro.library()\n"
            "%s\n"
            "n = len(ro.hip_runtimes())\n"
            "try:\n"
            "    ro.require_one_hip_runtime(); ok = True\n"
            "except ro.StftError as e:\n"
            "    ok = False; assert e.code == -5 and 'two HIP runtimes' in str(e)\n"
            "print(n, ok)\n")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="")       # nothing here needs the device
    good = subprocess.check_output([sys.executable, "-c", code % (ROOT, "import torch", "")], env=env, text=True).split()
    assert good == ["1", "True"], good
    bad = subprocess.check_output([sys.executable, "-c", code % (ROOT, "", "import torch")], env=env, text=True).split()
    assert bad == ["2", "False"], bad


def test_product_does_not_touch_the_oracle():
    """Nothing under radio-observer_amd/ may import, link or load oracle/."""
    pkg = os.path.join(ROOT, "radio-observer_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                text = open(os.path.join(d, f), errors="replace").read()
                assert "ro_oracle" not in text and "oracle/" not in text, os.path.join(d, f)


def test_shard_helpers_of_the_c_abi(ro):
    """ro_shard_rows / ro_shard_samples / ro_shard_max_rows / ro_stitch_rows: contiguous cover, sizes within one,
    halo = bins - hop, and the stitch undoes the equal-block padding of an all-gather."""
    for total in (0, 1, 7, 8, 41, 168747):
        for world in (1, 2, 3, 8):
            shards = [ro.shard_rows(total, world, g) for g in range(world)]
            assert shards[0][0] == 0 and sum(r for _, r in shards) == total
            for g in range(1, world):
                assert shards[g][0] == shards[g - 1][0] + shards[g - 1][1]
            assert ro.shard_max_rows(total, world) == max(r for _, r in shards)
            # an all-gather of zero-padded blocks, stitched back
            full = np.arange(total * 3, dtype=np.float32).reshape(total, 3)
            m = ro.shard_max_rows(total, world)
            gathered = np.zeros((world * m, 3), np.float32)
            for g, (first, rows) in enumerate(shards):
                gathered[g * m:g * m + rows] = full[first:first + rows]
            assert np.array_equal(ro.stitch_rows(gathered, total, world), full)
    first, rows = ro.shard_rows(168747, 8, 3)
    s0, ns = ro.shard_samples(first, rows, 32768, 24576)
    assert s0 == first * 8192 and ns == (rows - 1) * 8192 + 32768
    assert ro.shard_samples(5, 0, 4096, 2048) == (5 * 2048, 0)
    with pytest.raises(ro.StftError):
        ro.shard_rows(10, 4, 4)


def test_ln_levels_host_helper_matches_the_viewers_formula(ro, oracle):
    rng = np.random.default_rng(8)
    image = np.abs(rng.standard_normal((40, 77))).astype(np.float32) * 50
    image[3, 5] = 0.0
    ln, want_u8, (mn, mx) = oracle.ln_levels(image)
    assert np.array_equal(ro.ln_levels(ln, mn, mx), want_u8)
    assert not ro.ln_levels(ln, 1.0, 1.0).any()                     # flat range: all zero
