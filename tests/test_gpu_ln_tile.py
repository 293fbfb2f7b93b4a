"""GPU parity of the ln-magnitude band tile and its 8-bit grey image against the oracle's restatement of the offline
viewer (fits2png:46 FN_LOG, :444-445 default_color_fn, :476-477 min/max, :495-497 uint8 store).
Floating point: ln within 2 ulp of the oracle's float32 result (LN_ULPS below); min/max likewise; the grey levels are
checked twice -- bit-exact against the viewer's formula applied to the GPU's own ln/min/max (float32 arithmetic is
IEEE on both sides), and within one level of the oracle's image."""
import numpy as np
import pytest

from util import add_tone, noise_iq

pytestmark = pytest.mark.gpu

LN_ULPS = 2                # device logf and libm logf are each within 1 ulp of the true value


def ln_close(got, want):
    tol = LN_ULPS * np.spacing(np.maximum(np.abs(want), np.float32(1.0)).astype(np.float32))
    return bool(np.all(np.abs(got - want) <= tol))


def ln_gpu(ro, torch, rows, first, cols, want=("ln", "u8", "mm")):
    bins = rows.shape[1]
    R = rows.shape[0]
    d_rows = torch.from_numpy(rows).cuda()
    d_ln = torch.full((R, cols), 7.0, dtype=torch.float32, device="cuda") if "ln" in want else None
    d_u8 = torch.full((R, cols), 77, dtype=torch.uint8, device="cuda") if "u8" in want else None
    d_mm = torch.zeros(2, dtype=torch.float32, device="cuda") if "mm" in want else None
    with ro.Stft(bins=bins, overlap=0) as st:
        st.ln_tile_resident(d_rows, R, first, cols, d_ln=d_ln, d_u8=d_u8, d_minmax=d_mm,
                            stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    f = lambda t: None if t is None else t.cpu().numpy()
    return f(d_ln), f(d_u8), f(d_mm)


def viewer_levels(ln, image, mn, mx):
    with np.errstate(all="ignore"):
        lv = ((ln - np.float32(mn)) / (np.float32(mx) - np.float32(mn)) * np.float32(255)).astype(np.float32)
        out = np.where((image != 0) & (mx > mn), np.nan_to_num(lv, nan=0.0, posinf=0.0, neginf=0.0), 0).astype(np.uint8)
    return out


def check(ro, oracle, torch, rows, first, cols):
    image = np.ascontiguousarray(rows[:, first:first + cols])
    ln, u8, mm = ln_gpu(ro, torch, rows, first, cols)
    want_ln, want_u8, (mn, mx) = oracle.ln_levels(image)
    nz = image != 0
    assert ln_close(ln[nz], want_ln[nz])
    assert np.all(np.isneginf(ln[~nz]))
    assert ln_close(mm, np.array([mn, mx], np.float32))
    assert mm[0] == ln[nz].min() and mm[1] == ln[nz].max()             # the reduction itself is exact
    assert np.array_equal(u8, viewer_levels(ln, image, mm[0], mm[1]))   # the viewer's formula, bit for bit
    d = np.abs(u8.astype(np.int16) - want_u8.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3                      # ulp-level ln differences move few pixels
    return u8


def test_ln_tile_of_stft_rows_c3_band(ro, oracle, torch_cuda):
    """The snapshot band of radio-observer.json (10100-11000 Hz = columns 23278..23893 at N=32768) of real rows."""
    bins, overlap, hop = 32768, 24576, 8192
    rng = np.random.default_rng(0xF4)
    R = 48
    iq = add_tone(noise_iq(rng, bins + (R - 1) * hop), 10600.0, 30.0)
    rows = oracle.stft(iq, bins, overlap)
    lo = oracle.lib().ro_oracle_frequency_to_bin(bins, 48000, 10100.0)
    hi = oracle.lib().ro_oracle_frequency_to_bin(bins, 48000, 11000.0)
    assert (lo, hi) == (23278, 23893)
    u8 = check(ro, oracle, torch_cuda, rows, lo, hi - lo)
    assert u8.max() == 255 and u8.min() == 0
    assert u8[:, oracle.lib().ro_oracle_frequency_to_bin(bins, 48000, 10600.0) - lo].min() > 200    # the carrier


@pytest.mark.parametrize("bins,first,cols,R", [(256, 0, 256, 1), (1024, 1000, 24, 7), (4096, 1, 4094, 33),
                                               (32768, 0, 32768, 5)])
def test_ln_tile_shapes_and_ragged_edges(ro, oracle, torch_cuda, bins, first, cols, R):
    rng = np.random.default_rng(bins + R)
    rows = (np.abs(rng.standard_normal((R, bins))) * 10.0 ** rng.uniform(-6, 6, (R, 1))).astype(np.float32)
    check(ro, oracle, torch_cuda, rows, first, cols)


def test_ln_tile_zero_pixels_and_flat_images(ro, oracle, torch_cuda):
    rng = np.random.default_rng(9)
    rows = rng.random((12, 1024)).astype(np.float32) + 0.5
    rows[3, :] = 0.0                                         # a dropped row (zeros are not part of min/max)
    rows[5, 100:200] = 0.0
    u8 = check(ro, oracle, torch_cuda, rows, 64, 512)
    assert (u8[3] == 0).all()
    flat = np.full((4, 256), 2.5, np.float32)                # max == min: the viewer divides by zero; level 0 here
    ln, u8, mm = ln_gpu(ro, torch_cuda, flat, 0, 256)
    assert (u8 == 0).all() and mm[0] == mm[1] and ln_close(mm[:1], np.log(np.float32([2.5])))
    want = oracle.ln_levels(flat)
    assert (want[1] == 0).all()


def test_ln_tile_optional_outputs_and_errors(ro, oracle, torch_cuda):
    rng = np.random.default_rng(10)
    rows = rng.random((9, 512)).astype(np.float32) + 0.01
    full = ln_gpu(ro, torch_cuda, rows, 10, 300)
    only_u8 = ln_gpu(ro, torch_cuda, rows, 10, 300, want=("u8",))
    only_mm = ln_gpu(ro, torch_cuda, rows, 10, 300, want=("mm",))
    assert np.array_equal(only_u8[1], full[1]) and np.array_equal(only_mm[2], full[2])
    d_rows = torch_cuda.from_numpy(rows).cuda()
    with ro.Stft(bins=512, overlap=0) as st:
        for first, cols in ((-1, 10), (500, 13), (0, 0)):
            with pytest.raises(ro.StftError):
                st.ln_tile_resident(d_rows, 9, first, cols, d_minmax=torch_cuda.zeros(2, device="cuda"))
        with pytest.raises(ro.StftError):
            st.ln_tile_resident(d_rows, 9, 0, 16)            # no output requested


@pytest.mark.parametrize("bins,overlap,tile", [(32768, 24576, (23278, 615)), (32768, 24576, (22528, 2048)),
                                               (4096, 2048, (2900, 300)), (65536, 49152, (40000, 1000))])
def test_fused_ln_tile_of_the_transform(ro, oracle, torch_cuda, bins, overlap, tile):
    """tile_ln = 1: the transform itself hands out the tile, its log and every row's min / max of the log (bins = 32768:
    from the magnitudes still in LDS; other sizes: a small kernel over the tile).  Same bars as the stand-alone ln tile;
    the image range is the min / max over the rows, and ro_ln_levels gives the viewer's grey levels."""
    torch = torch_cuda
    hop = bins - overlap
    R = 37
    rng = np.random.default_rng(bins + tile[1])
    iq = add_tone(noise_iq(rng, bins + (R - 1) * hop), 10600.0, 20.0)
    d_iq = torch.from_numpy(iq).cuda()
    rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
    d_tile = torch.empty((R, tile[1]), dtype=torch.float32, device="cuda")
    d_ln = torch.full((R, tile[1]), 7.0, dtype=torch.float32, device="cuda")
    d_mm = torch.zeros((R, 2), dtype=torch.float32, device="cuda")
    with ro.Stft(bins=bins, overlap=overlap, tile=tile, tile_ln=True) as st:
        st.run_resident_ln(d_iq, ro.RO_IQ_F32, iq.shape[0], 0, R, rows, d_tile, d_ln=d_ln, d_minmax=d_mm,
                           stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    image = rows.cpu().numpy()[:, tile[0]:tile[0] + tile[1]]
    assert np.array_equal(d_tile.cpu().numpy(), image)
    ln, mm = d_ln.cpu().numpy(), d_mm.cpu().numpy()
    want_ln, want_u8, (mn, mx) = oracle.ln_levels(np.ascontiguousarray(image))
    assert ln_close(ln, want_ln)
    assert np.array_equal(mm[:, 0], ln.min(axis=1)) and np.array_equal(mm[:, 1], ln.max(axis=1))   # reductions are exact
    gmn, gmx = mm[:, 0].min(), mm[:, 1].max()
    assert ln_close(np.array([gmn, gmx], np.float32), np.array([mn, mx], np.float32))
    u8 = ro.ln_levels(ln, gmn, gmx)
    assert np.array_equal(u8, viewer_levels(ln, image, gmn, gmx))
    d = np.abs(u8.astype(np.int16) - want_u8.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3


def test_streamed_ln_tile_equals_resident(ro, oracle, torch_cuda):
    torch = torch_cuda
    bins, overlap, hop, tile = 32768, 24576, 8192, (23278, 615)
    rng = np.random.default_rng(21)
    R = 19
    iq = noise_iq(rng, bins + (R - 1) * hop + 100)
    iq[5 * hop:5 * hop + 64] = 0.0
    with ro.Stft(bins=bins, overlap=overlap, tile=tile, tile_ln=True, max_batch_rows=4) as st:
        for i in range(0, len(iq), 4096):
            st.push(iq[i:i + 4096])
        st.flush()
        first, t, ln, mm, _ = st.fetch_ln(1000)
        assert first == 0 and t.shape == ln.shape == (R, 615) and mm.shape == (R, 2)
        d_iq = torch.from_numpy(iq).cuda()
        rows = torch.empty((R, bins), dtype=torch.float32, device="cuda")
        d_tile = torch.empty((R, 615), dtype=torch.float32, device="cuda")
        d_ln = torch.empty((R, 615), dtype=torch.float32, device="cuda")
        d_mm = torch.empty((R, 2), dtype=torch.float32, device="cuda")
        st.run_resident_ln(d_iq, ro.RO_IQ_F32, iq.shape[0], 0, R, rows, d_tile, d_ln=d_ln, d_minmax=d_mm,
                           stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        with pytest.raises(ro.StftError):
            st.run_resident_ln(d_iq, ro.RO_IQ_F32, iq.shape[0], 0, R, rows, d_tile, d_minmax=d_mm)
    assert np.array_equal(t, d_tile.cpu().numpy()) and np.array_equal(ln, d_ln.cpu().numpy())
    assert np.array_equal(mm, d_mm.cpu().numpy())
