// ro_stft_capi.cpp -- the handle of the C ABI in include/ro_stft.h: create / destroy, the transform launches of every plan,
// the resident entry points (rows, spectra, scan, ln tile, timing).  Owns the window / twiddle tables in HBM.  The rest of
// the ABI: ro_abi_helpers.cpp, ro_exchange.cpp, ro_stream.cpp, ro_czt.cpp (ro_host.h lists who has what).
#include "ro_host.h"

using namespace ro::host;

namespace {

// per-stage twiddle tables in the layout apply_twiddles() reads:
//   stage with radix R after sub-transforms of length NS: entry (r-1)*NS + k = exp(-2 pi i r k / (NS R))
// evaluated in long double and rounded once to float.
std::vector<float2> build_twiddles(int bins)
{
    int radix[4];
    std::vector<float2> tw;
    if (!ro::stft_radices(bins, radix)) return tw;
    const long double two_pi = 8.0L * atanl(1.0L);
    long ns = radix[0];
    for (int s = 1; s < 4; ++s) {
        const int R = radix[s];
        if (R <= 1) continue;
        for (int r = 1; r < R; ++r)
            for (long k = 0; k < ns; ++k) {
                const long double ang = -two_pi * (long double)((long long)r * k) / (long double)(ns * R);
                tw.push_back(make_float2((float)cosl(ang), (float)sinl(ang)));
            }
        ns *= R;
    }
    return tw;
}

// exp(-2 pi i m / N) in double, correctly rounded from long double (strict-precision path)
std::vector<double2> build_full_twiddles_f64(int bins)
{
    std::vector<double2> tw((size_t)bins);
    const long double two_pi = 8.0L * atanl(1.0L);
    for (int m = 0; m < bins; ++m) {
        const long double ang = -two_pi * (long double)m / (long double)bins;
        tw[(size_t)m] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    return tw;
}

int check_bands(const ro_stft *h, const ro_bands_t &b)
{
    if (b.noise_width <= 0 || b.detect_width <= 0 || b.avg_bins <= 0)
        return fail(RO_ERR_INVALID, "bands: widths and avg_bins must be positive");
    if (b.low_noise < 0 || b.low_noise + b.noise_width > h->bins)
        return fail(RO_ERR_INVALID, "bands: noise band [%d,+%d) outside [0,%d)", b.low_noise,
                    b.noise_width, h->bins);
    if (b.low_detect < 0 || b.low_detect + b.detect_width > h->bins)
        return fail(RO_ERR_INVALID, "bands: detect band [%d,+%d) outside [0,%d)", b.low_detect,
                    b.detect_width, h->bins);
    return RO_OK;
}

int validate_resident(const ro_stft *h, const void *d_iq, int format, int64_t samples,
                      int64_t first_row, int64_t rows, const float *d_rows, int64_t row_stride,
                      const float *d_tile, const ro_scan_record_t *d_records)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    if (format == RO_IQ_F64 && !h->f64reg)
        return fail(RO_ERR_UNSUPPORTED, "resident RO_IQ_F64 samples are taken by RO_PRECISION_F64 handles of 256 ... 65536 bins only");
    if (format != RO_IQ_F32 && format != RO_IQ_I16 && format != RO_IQ_F64)
        return fail(RO_ERR_INVALID, "resident input must be RO_IQ_F32, RO_IQ_I16 or RO_IQ_F64 (got %d)", format);
    if (rows < 0 || first_row < 0) return fail(RO_ERR_INVALID, "negative row range");
    if (rows == 0) return RO_OK;
    if (!d_iq) return fail(RO_ERR_INVALID, "null input pointer");
    if (!d_rows) return fail(RO_ERR_INVALID, "d_rows is required (the band tile is cut from the rows)");
    if (d_rows && row_stride < h->bins)
        return fail(RO_ERR_INVALID, "row_stride %lld < bins %d", (long long)row_stride, h->bins);
    if (d_tile && h->cfg.tile_cols <= 0) return fail(RO_ERR_INVALID, "tile output requested but tile_cols == 0");
    if (d_records && !h->cfg.enable_scan) return fail(RO_ERR_INVALID, "records requested but enable_scan == 0");
    if (d_records && !d_rows) return fail(RO_ERR_INVALID, "records need full rows (d_rows)");
    // every workgroup reads samples [r*hop, r*hop + bins): the last one must stay inside the buffer
    const int64_t last = (first_row + rows - 1) * (int64_t)h->hop + h->bins;
    if (last > samples)
        return fail(RO_ERR_INVALID, "rows [%lld,+%lld) need %lld samples, buffer holds %lld",
                    (long long)first_row, (long long)rows, (long long)last, (long long)samples);
    if (rows > (int64_t)0x0fffffff) return fail(RO_ERR_INVALID, "too many rows in one launch");
    return RO_OK;
}

ro::TileArgs make_tile_args(const ro_stft *h, const float *d_rows, int64_t row_stride, int64_t rows,
                            float *d_tile)
{
    ro::TileArgs t{};
    t.rows_in = d_rows;
    t.tile_out = d_tile;
    t.rows = rows;
    t.row_stride = row_stride;
    t.first = h->cfg.tile_first_col;
    t.cols = h->cfg.tile_cols;
    return t;
}

ro::ScanArgs make_scan_args(const ro_stft *h, const float *d_rows, int64_t row_stride, int64_t rows,
                            ro_scan_record_t *d_records)
{
    ro::ScanArgs s{};
    s.rows_in = d_rows;
    s.records = d_records;
    s.rows = rows;
    s.row_stride = row_stride;
    s.bins = h->bins;
    s.low_noise = h->cfg.bands.low_noise;
    s.noise_width = h->cfg.bands.noise_width;
    s.low_detect = h->cfg.bands.low_detect;
    s.detect_width = h->cfg.bands.detect_width;
    s.avg_bins = h->cfg.bands.avg_bins;
    return s;
}

// scratch of the large sizes' scratch form (and of their complex spectra): rows per chunk and the blocks
int ensure_big_scratch(ro_stft *h)
{
    if (!h->spec_rows) {
        h->spec_rows = std::max<int64_t>(1, ((int64_t)RO_SPEC_SCRATCH_MB << 20) / ((int64_t)h->bins * 8));
        if (h->spec_rows > 65535) h->spec_rows = 65535;
    }
    if (!h->d_spec) HIP_TRY(hipMalloc(&h->d_spec, (size_t)h->spec_rows * h->bins * sizeof(float2)));
    if (!h->d_spec2) HIP_TRY(hipMalloc(&h->d_spec2, (size_t)h->spec_rows * h->bins * sizeof(float2)));
    return RO_OK;
}

#ifdef RO_DIAG
// Diagnostic builds only (RO_F64_FUSED=1): round 5's one-launch form of the through-HBM FP64 passes, the complex-double
// intermediate of a row handed between workgroups through the L2 of the XCD that makes it (ro_f64fused.hip; measured
// slower than the two launches, profiles/r05_f64_one_launch.txt; not in the product library).  The launch's give-up word
// travels to pinned host memory behind the kernel; a launch that gave up is reported by the NEXT call on the handle.
int launch_transform_f64_fused(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float *d_rows,
                               int64_t row_stride, hipStream_t s)
{
    if (h->h_f64_err && *h->h_f64_err != 0) {
        const unsigned code = *h->h_f64_err;
        *h->h_f64_err = 0;
        return fail(RO_ERR_HIP, "the previous RO_PRECISION_F64 launch gave up waiting for another workgroup (code %u: 1 = row "
                                "map, 2 = ring slot still being read, 3 = row's first half not complete); its rows are incomplete", code);
    }
    if (!h->d_f64_ring) {
        int ring = RO_F64_RING_ROWS * 32768 / h->bins, wgs = RO_F64_WGS_PER_CU;
#ifdef RO_DIAG_KNOBS
        if (const char *e = getenv("RO_F64_RING_ROWS")) ring = atoi(e);
        if (const char *e = getenv("RO_F64_WGS")) wgs = atoi(e);
#endif
        h->f64_ring_rows = std::max(2, std::min(ring, ro::f64_fused_max_ring_rows()));
        h->f64_wgs_per_cu = std::max(1, std::min(wgs, 2));
        HIP_TRY(hipMalloc(&h->d_f64_ring, (size_t)8 * h->f64_ring_rows * h->bins * sizeof(double2)));
        HIP_TRY(hipMalloc(&h->d_f64_ctl, ro::f64_fused_ctl_bytes()));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->h_f64_err), sizeof(unsigned), hipHostMallocDefault));
        *h->h_f64_err = 0;
    }
    ro::BigArgsD b{};
    b.iq = d_iq;
    b.window = h->d_window;
    b.tw = h->d_tw_f64;
    b.first_row = first_row;
    b.rows = rows;
    b.row_stride = row_stride;
    b.hop = h->hop;
    b.n = h->bins;
    b.gain = h->cfg.iq_gain;
    b.rows_out = d_rows;
    HIP_TRY(ro::launch_f64_fused(format, b, h->d_f64_ring, h->d_f64_ctl, h->f64_ring_rows, h->f64_wgs_per_cu, s));
    HIP_TRY(hipMemcpyAsync(h->h_f64_err, h->d_f64_ctl + 1, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    return RO_OK;
}
#endif

// RO_PRECISION_F64: every size as radix-16 passes in double through HBM scratch, in chunks that fit it
int launch_transform_f64(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float *d_rows,
                         int64_t row_stride, hipStream_t s)
{
#ifdef RO_DIAG
    if (const char *e = getenv("RO_F64_FUSED"))
        if (atoi(e) != 0 && ro::f64_fused_supported(h->bins))
            return launch_transform_f64_fused(h, d_iq, format, first_row, rows, d_rows, row_stride, s);
#endif
    if (h->f64reg) {
        ro::F64RegArgs r{};
        r.iq = d_iq;
        r.window_k = h->d_f64r_window;
        r.tw0 = h->d_f64r_tw[0];
        r.tw1 = h->d_f64r_tw[1];
        r.tw2 = h->d_f64r_tw[2];
        r.tw3 = h->d_f64r_tw[3];
        r.rows_out = d_rows;
        r.first_row = first_row;
        r.rows = rows;
        r.row_stride = row_stride;
        r.hop = h->hop;
        r.gain = h->cfg.iq_gain;
        r.stamps = h->d_stamps;
        HIP_TRY(ro::launch_f64reg(h->bins, format, r, s));
        return RO_OK;
    }
    if (!h->d_scratch_d[0]) {
        int64_t mib = RO_F64_SCRATCH_MB;
#ifdef RO_DIAG
        if (const char *e = getenv("RO_F64_SCRATCH_MB")) mib = std::max(1, atoi(e));     // tools/r4/strict_sweep.sh
#endif
        h->scratch_rows_d = std::max<int64_t>(1, (mib << 20) / ((int64_t)h->bins * 16));
        for (int i = 0; i < 2; ++i)
            HIP_TRY(hipMalloc(&h->d_scratch_d[i], (size_t)h->scratch_rows_d * h->bins * sizeof(double2)));
    }
    int radix[8];
    const int passes = ro::f64_radices(h->bins, radix);
    if (passes < 2) return fail(RO_ERR_UNSUPPORTED, "no FP64 pass plan for bins = %d", h->bins);
    for (int64_t done = 0; done < rows; done += h->scratch_rows_d) {
        const int64_t n = std::min(h->scratch_rows_d, rows - done);
        ro::BigArgsD b{};
        b.iq = d_iq;
        b.window = h->d_window;
        b.tw = h->d_tw_f64;
        b.first_row = first_row + done;
        b.rows = n;
        b.row_stride = row_stride;
        b.hop = h->hop;
        b.n = h->bins;
        b.gain = h->cfg.iq_gain;
        // two passes per kernel where the pair fits its LDS tile (radix 16 followed by any radix, bins >= 4096): half
        // the trips through the scratch blocks
        int ns = 1, hop_idx = 0;
        b.rows_out = d_rows + done * row_stride;
        for (int p = 0; p < passes;) {
            const bool pair = h->bins >= 4096 && p + 1 < passes && radix[p] == 16;
            const int last_p = pair ? p + 1 : p;
            b.ns = ns;
            b.in = h->d_scratch_d[(hop_idx + 1) & 1];
            b.out = h->d_scratch_d[hop_idx & 1];
            if (pair) HIP_TRY(ro::launch_f64_pair(radix[p + 1], p == 0, last_p == passes - 1, format, b, s));
            else HIP_TRY(ro::launch_f64_pass(radix[p], p == 0, p == passes - 1, format, b, s));
            for (int q = p; q <= last_p; ++q) ns *= radix[q];
            p = last_p + 1;
            ++hop_idx;
        }
    }
    return RO_OK;
}

}  // namespace

namespace ro {
namespace host {

ro::StftArgs make_stft_args(const ro_stft *h, const void *d_iq, int64_t first_row, int64_t rows,
                            float *d_rows, int64_t row_stride, float *d_tile,
                            ro_scan_record_t *d_records, float *d_ln)
{
    ro::StftArgs a{};
    a.iq = d_iq;
    a.window = h->d_window;
    a.window_k = h->d_window_k;
    a.window_k32 = h->d_window_k32;
    a.twiddles = h->d_twiddles;
    a.twiddles_k = h->d_twiddles_k;
    a.rows_out = d_rows;
    a.first_row = first_row;
    a.rows = rows;
    a.row_stride = row_stride;
    a.hop = h->hop;
    a.gain = (float)h->cfg.iq_gain;
    a.stamps = h->d_stamps;
    a.spare_cus = h->cfg.spare_cus_per_xcd;
    // plans with a fused epilogue scan / tile take them here; for the others the caller launches the separate kernels
    if (!h->f64 && ro::stft_fuses_scan(h->bins)) {
        a.records = d_records;
        a.low_noise = h->cfg.bands.low_noise;
        a.noise_width = h->cfg.bands.noise_width;
        a.low_detect = h->cfg.bands.low_detect;
        a.detect_width = h->cfg.bands.detect_width;
        a.avg_bins = h->cfg.bands.avg_bins;
        a.tile_out = d_tile;
        a.tile_first = h->cfg.tile_first_col;
        a.tile_cols = h->cfg.tile_cols;
        a.ln_out = d_tile ? d_ln : nullptr;
        a.ln_part = h->d_ln_part;
    }
    return a;
}

// d_ln / d_minmax: the tile's log and the rows' min / max of it (tile_ln); need d_tile
int launch_tile_and_scan(ro_stft *h, const float *d_rows, int64_t row_stride, int64_t rows, float *d_tile,
                         ro_scan_record_t *d_records, hipStream_t s, float *d_ln, float *d_minmax)
{
    const bool want_ln = d_tile && (d_ln || d_minmax);
    if (!h->f64 && ro::stft_fuses_scan(h->bins)) {                  // tile, log and records written by the transform
        if (want_ln && d_minmax) HIP_TRY(ro::launch_ln_finish(h->d_ln_part, d_minmax, rows, s));
        return RO_OK;
    }
    if (d_tile) HIP_TRY(ro::launch_tile(make_tile_args(h, d_rows, row_stride, rows, d_tile), s));
    if (want_ln) HIP_TRY(ro::launch_ln_rows(d_tile, d_ln, d_minmax, rows, h->cfg.tile_cols, s));
    if (d_records) HIP_TRY(ro::launch_scan(make_scan_args(h, d_rows, row_stride, rows, d_records), s));
    return RO_OK;
}

// scratch of the fused log: the two tile waves' partial min / max, 4 floats per row of the launch
int ensure_ln_part(ro_stft *h, int64_t rows)
{
    if (rows <= h->ln_part_rows) return RO_OK;
    // sized generously the first time (16 bytes per row) and doubled after that, so that the device-wide wait a
    // regrow needs -- an earlier launch may still be writing the old block -- happens at most a few times per handle
    int64_t want = std::max<int64_t>(rows, std::max<int64_t>(65536, 2 * h->ln_part_rows));
    if (h->d_ln_part) { (void)hipDeviceSynchronize(); (void)hipFree(h->d_ln_part); h->d_ln_part = nullptr; h->ln_part_rows = 0; }
    HIP_TRY(hipMalloc(&h->d_ln_part, (size_t)want * 4 * sizeof(float)));
    h->ln_part_rows = want;
    return RO_OK;
}

// complex spectra of rows [first_row, +rows) of a large size (bins = dec x 32768), bin k at element k of each row:
// fold_kernel, the N = 32768 kernel in spectra mode on its rows, interleave2_kernel
int launch_spectra_big(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float2 *d_out,
                       int64_t out_stride, hipStream_t s)
{
    int rc = ensure_big_scratch(h);
    if (rc != RO_OK) return rc;
    for (int64_t done = 0; done < rows; done += h->spec_rows) {
        const int64_t n = std::min(h->spec_rows, rows - done);
        ro::FoldArgs f{};
        f.iq = d_iq;
        f.window = h->d_window;
        f.rot = h->d_tw_combine;
        f.out = h->d_spec;
        f.first_row = first_row + done;
        f.rows = n;
        f.hop = h->hop;
        f.m = h->sub_bins;
        f.dec = h->dec;
        f.gain = (float)h->cfg.iq_gain;
        HIP_TRY(ro::launch_fold(format, f, s));
        ro::StftArgs a = make_stft_args(h, h->d_spec, 0, n * h->dec, nullptr, 0);
        a.window = h->d_ones;
        a.window_k = h->d_ones;
        a.window_k32 = h->d_ones;
        a.hop = h->sub_bins;
        a.gain = 0.0f;
        a.spec_out = h->d_spec2;
        a.spec_stride = h->sub_bins;
        HIP_TRY(ro::launch_stft(h->sub_bins, RO_FMT_F32, a, s));
        ro::Interleave2Args t{};
        t.in = h->d_spec2;
        t.out = d_out + done * out_stride;
        t.rows = n;
        t.out_stride = out_stride;
        t.m = h->sub_bins;
        t.dec = h->dec;
        HIP_TRY(ro::launch_interleave2(t, s));
    }
    return RO_OK;
}

int launch_transform(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float *d_rows,
                     int64_t row_stride, hipStream_t s, float *d_tile, ro_scan_record_t *d_records, float *d_ln)
{
    if (h->f64) return launch_transform_f64(h, d_iq, format, first_row, rows, d_rows, row_stride, s);
    if (h->czt) return launch_transform_czt(h, d_iq, format, first_row, rows, d_rows, row_stride, s);
    if (!h->big) {
        if (d_tile && d_ln && ro::stft_fuses_scan(h->bins)) {
            int rc = ensure_ln_part(h, rows);
            if (rc != RO_OK) return rc;
        }
        ro::StftArgs a = make_stft_args(h, d_iq, first_row, rows, d_rows, row_stride, d_tile, d_records, d_ln);
        HIP_TRY(ro::launch_stft(h->bins, format, a, s));
        return RO_OK;
    }
    if (h->four) {
        {
            int64_t mib = RO_FOUR_SCRATCH_MB;
#ifdef RO_DIAG_KNOBS
            if (const char *e = getenv("RO_FOUR_SCRATCH_MB")) mib = std::max<int64_t>(4, atoll(e));
#endif
            const int64_t limit = std::max<int64_t>(1, (mib << 20) / ((int64_t)h->bins * 8));
            const int64_t want = std::min(limit, rows);
            if (want > h->four_rows) {
                // The old block may still be in use: by launches queued on the caller's stream, on the handle's own, or on
                // a streaming slot's (captured graphs are not taken for these sizes, run_stream_batch, but a handle may be
                // used both ways).  Everything the handle has queued anywhere is waited for before the block goes.
                if (h->d_four_z) {
                    HIP_TRY(hipStreamSynchronize(s));
                    if (h->stream && h->stream != s) HIP_TRY(hipStreamSynchronize(h->stream));
                    for (ro_stft::Slot &sl : h->slot)
                        if (sl.gstream && sl.gstream != s) HIP_TRY(hipStreamSynchronize(sl.gstream));
                    HIP_TRY(hipFree(h->d_four_z));
                    h->d_four_z = nullptr;
                    h->four_rows = 0;
                }
                HIP_TRY(hipMalloc(&h->d_four_z, (size_t)want * h->bins * 2 * sizeof(float)));
                h->four_rows = want;
            }
        }
        for (int64_t done = 0; done < rows; done += h->four_rows) {
            ro::FourArgs f{};
            f.iq = d_iq;
            f.first_row = first_row + done;
            f.rows = std::min(h->four_rows, rows - done);
            f.hop = h->hop;
            f.gain = (float)h->cfg.iq_gain;
            f.n1 = h->bins / 1024;
            f.window_a = h->d_four_window;
            f.tw_a = h->d_four_tw_a;
            f.tw_b = h->d_four_tw_b;
            f.tw_r = h->d_four_tw_r;
            f.z = h->d_four_z;
            f.rows_out = d_rows + done * row_stride;
            f.row_stride = row_stride;
            f.spare_cus = h->cfg.spare_cus_per_xcd;
            HIP_TRY(ro::launch_fourstep(format, f, s));
        }
        return RO_OK;
    }
    if (h->dec > 1 && h->dif) {
        // one kernel, no scratch: kernel row srow * dec + q makes the bins q + dec k' of stream row srow
        int log2 = 0;
        while ((1 << log2) < h->dec) ++log2;
        ro::StftArgs a = make_stft_args(h, d_iq, first_row, rows * h->dec, d_rows, row_stride);
        a.window = h->d_window;
        a.window_k = h->d_window_dif;
        a.dec = h->dec;
        a.dec_log2 = log2;
        a.dif_tw = h->d_dif_tw;
        a.dif_shift = h->d_dif_shift;
        a.big_form = 1;
        HIP_TRY(ro::launch_stft(h->sub_bins, format, a, s));
        return RO_OK;
    }
    return fail(RO_ERR_STATE, "internal: no transform plan for bins = %d", h->bins);
}

}  // namespace host
}  // namespace ro

extern "C" int ro_bins_supported(int bins)
{
    return (ro::stft_supported(bins) || ro::big_supported(bins) || czt_length(bins) > 0) ? 1 : 0;
}

// ---------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------
extern "C" int ro_stft_create(const ro_stft_config_t *cfg_in, ro_stft_t **out)
{
    if (!cfg_in || !out) return fail(RO_ERR_INVALID, "ro_stft_create: null argument");
    *out = nullptr;
    const ro_stft_config_t *cfg = cfg_in;          // re-pointed at a full-size copy once struct_size is known
    // ABI growth: fields are only ever appended; a caller built against ABI 1 passes the shorter struct and gets the
    // defaults (0) for what it does not know
    const size_t abi1_size = offsetof(ro_stft_config_t, precision);
    if (cfg->struct_size != sizeof(ro_stft_config_t) && cfg->struct_size != abi1_size)
        return fail(RO_ERR_INVALID, "ro_stft_create: struct_size %u is neither %zu (ABI 2) nor %zu (ABI 1)",
                    cfg->struct_size, sizeof(ro_stft_config_t), abi1_size);
    ro_stft_config_t cfg_full{};
    std::memcpy(&cfg_full, cfg_in, cfg_in->struct_size);
    cfg_full.struct_size = sizeof(ro_stft_config_t);
    cfg = &cfg_full;
    if (cfg->precision == 2)                       // ABI 4's RO_PRECISION_F64_ONE_LAUNCH: an experiment, retired from the library
        return fail(RO_ERR_UNSUPPORTED, "precision 2 (the one-launch form of RO_PRECISION_F64) is no longer part of the library: "
                                        "use RO_PRECISION_F64");
    if (cfg->precision != RO_PRECISION_F32 && cfg->precision != RO_PRECISION_F64)
        return fail(RO_ERR_INVALID, "unknown precision %d", cfg->precision);
    if (cfg->tile_ln != 0 && cfg->tile_ln != 1) return fail(RO_ERR_INVALID, "tile_ln must be 0 or 1");
    if (cfg->tile_ln && cfg->tile_cols <= 0) return fail(RO_ERR_INVALID, "tile_ln needs a tile (tile_cols > 0)");
    if (!ro_bins_supported(cfg->bins))
        return fail(RO_ERR_UNSUPPORTED, "bins = %d has no kernel (powers of two 256..1048576, other even lengths "
                                        "258..524286)", cfg->bins);
    if (czt_length(cfg->bins) && cfg->precision != RO_PRECISION_F32)
        return fail(RO_ERR_UNSUPPORTED, "RO_PRECISION_F64 is available for power-of-two bins only");
    if (cfg->iq_phase_shift != 0)
        return fail(RO_ERR_UNSUPPORTED, "iq_phase_shift != 0 is undefined behaviour in the reference "
                                        "(src/FFTBackend.cpp:67-71) and is not supported");
    if (cfg->sample_rate <= 0) return fail(RO_ERR_INVALID, "sample_rate must be positive");
    if (cfg->spare_cus_per_xcd < 0 || cfg->spare_cus_per_xcd > 16)
        return fail(RO_ERR_INVALID, "spare_cus_per_xcd must be in [0, 16]");
    if (cfg->window_kind == RO_WINDOW_CUSTOM && !cfg->window_table)
        return fail(RO_ERR_INVALID, "RO_WINDOW_CUSTOM needs window_table");
    if (cfg->window_kind < RO_WINDOW_NUTTALL || cfg->window_kind > RO_WINDOW_CUSTOM)
        return fail(RO_ERR_INVALID, "unknown window_kind %d", cfg->window_kind);
    if (cfg->tile_cols < 0 || cfg->tile_first_col < 0 || cfg->tile_first_col + cfg->tile_cols > cfg->bins)
        return fail(RO_ERR_INVALID, "tile [%d,+%d) outside [0,%d)", cfg->tile_first_col, cfg->tile_cols,
                    cfg->bins);

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(RO_ERR_HIP, "device %d not present (%d devices)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));

    ro_stft *h = new (std::nothrow) ro_stft();
    if (!h) return fail(RO_ERR_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->cfg.window_table = nullptr;
    h->bins = cfg->bins;
    h->overlap = ro_clamp_overlap(cfg->bins, cfg->overlap);
    h->hop = h->bins - h->overlap;
    h->device = cfg->device;
    if (cfg->enable_scan) {
        int rc = check_bands(h, cfg->bands);
        if (rc != RO_OK) { delete h; return rc; }
    }

    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, cfg->device);
    if (e != hipSuccess) { delete h; return fail(RO_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e)); }
    h->device_name = prop.name;
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        delete h;
        return fail(RO_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 only",
                    cfg->device, prop.gcnArchName);
    }
    // The N = 32768 kernel (and what is built on it) wants 132 KiB of LDS for ONE workgroup and spreads its grid over
    // eight XCDs (blockIdx % 8, round-robin dispatch): say so here rather than with an opaque launch error later.  A
    // partition mode with fewer XCDs per device only loses the placement (rows of one XCD's run no longer share an L2).
    if ((prop.maxSharedMemoryPerMultiProcessor > 0 && (size_t)prop.maxSharedMemoryPerMultiProcessor < (size_t)132 * 1024) ||
        prop.multiProcessorCount < 8) {
        const size_t lds = (size_t)prop.maxSharedMemoryPerMultiProcessor;
        const int cus = prop.multiProcessorCount;
        delete h;
        return fail(RO_ERR_UNSUPPORTED, "device %d offers %zu bytes of LDS per CU and %d CUs; the kernels need 135168 for one workgroup and 8",
                    cfg->device, lds, cus);
    }

    // ... and the persistent kernels hand rows out in eight runs, one per XCD (blockIdx % 8 under round-robin dispatch;
    // ro_stft32k.hip, ro_kernels.hip, ro_fourstep.hip, ro_f64reg.hip).  A partition mode with another number of XCDs
    // computes the same rows and only loses the placement (rows of one run no longer share an L2): said once, not refused.
    {
        int xccs = 0;
        if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, cfg->device) == hipSuccess) {
            static bool said = false;
            if (xccs > 0 && xccs != 8 && !said) {            // (0: a runtime that answers without knowing)
                said = true;
                fprintf(stderr, "libro_stft: device %d reports %d XCDs; the kernels' row placement is laid out for 8 (an MI355X in "
                                "SPX mode): rows are unaffected, the L2 sharing between neighbouring rows is lost\n", cfg->device, xccs);
            }
        } else {
            (void)hipGetLastError();
        }
    }

    h->window.resize(h->bins);
    if (cfg->window_kind == RO_WINDOW_CUSTOM)
        std::memcpy(h->window.data(), cfg->window_table, sizeof(float) * h->bins);
    else
        build_window(cfg->window_kind, h->bins, h->window.data());
    h->big = ro::big_supported(h->bins);
    h->f64 = cfg->precision != RO_PRECISION_F32;
    h->czt_m = czt_length(h->bins);
    h->czt = h->czt_m > 0;
    if (h->big && !h->f64) {
        // bins = dec x 32768, decimation in frequency on the largest single-pass plan:
        //   X[q + dec k'] = sum_m W_N^(m k') { W_bins^(m q) sum_r W_dec^(r q) w[m + N r] x[m + N r] }
        // 65536, 131072: ONE kernel (ro::stft_kernel MODE 3) that sums the row's dec blocks itself -- dec x the row
        // through L2 per output row, nothing extra through HBM (0.27 / 0.19 of the HBM peak; it was 0.11 / 0.06 at
        // 262144 / 524288, where its 4-byte stores `dec` floats apart cost an L2 write request each).
        // 262144 ... 1048576: the four-step pair of kernels (ro_fourstep.hip) for the magnitude rows.
        // Complex spectra of every large size: fold_kernel writes the braces to scratch, the N = 32768 kernel transforms
        // those rows, interleave2_kernel puts the bins in place (launch_spectra_big).  Rounds 2 and 3 made the magnitude
        // rows of dec >= 8 that way too (0.10-0.13 of the peak; profiles/r04_fourstep.txt has the A/B).
        // (Diagnostic builds: RO_BIG_FORM=dif forces the one-kernel form.)
        const int sub = 32768, dec = h->bins / sub;
        bool four = ro::fourstep_supported(h->bins);
#ifdef RO_DIAG_KNOBS
        if (const char *e = getenv("RO_BIG_FORM")) four = std::strcmp(e, "dif") != 0 && four;
#endif
        h->four = four;
        h->dif = !four;
        h->sub_bins = sub;
        h->dec = dec;
    }
    const int plan_bins = h->czt ? 0 : h->big ? h->sub_bins : h->bins;   // whose stage tables this handle needs (0: none)
    std::vector<float2> tw = plan_bins ? build_twiddles(plan_bins) : std::vector<float2>();
    if (plan_bins && (int)tw.size() != ro::stft_twiddle_count(plan_bins)) {
        delete h;
        return fail(RO_ERR_STATE, "internal: twiddle table size mismatch");
    }

#define CREATE_TRY(expr)                                                                   \
    do {                                                                                   \
        hipError_t e2_ = (expr);                                                           \
        if (e2_ != hipSuccess) {                                                           \
            int rc2_ = fail(RO_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e2_));   \
            ro_stft_destroy(h);                                                            \
            return rc2_;                                                                   \
        }                                                                                  \
    } while (0)
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreate(&h->ev0));
    CREATE_TRY(hipEventCreate(&h->ev1));
    CREATE_TRY(hipMalloc(&h->d_window, sizeof(float) * h->bins));
    CREATE_TRY(hipMalloc(&h->d_ln_keys, 16 * 2 * sizeof(unsigned)));
    CREATE_TRY(hipMalloc(&h->d_twiddles, sizeof(float2) * std::max<size_t>(tw.size(), 1)));
    CREATE_TRY(hipMemcpy(h->d_window, h->window.data(), sizeof(float) * h->bins, hipMemcpyHostToDevice));
    if (!h->big && !h->czt) {
        std::vector<float> wk((size_t)h->bins);
        if (!ro::stft_window_layout(h->bins, h->window.data(), wk.data())) {
            ro_stft_destroy(h);
            return fail(RO_ERR_UNSUPPORTED, "no window layout for bins=%d", h->bins);
        }
        CREATE_TRY(hipMalloc(&h->d_window_k, sizeof(float) * h->bins));
        CREATE_TRY(hipMemcpy(h->d_window_k, wk.data(), sizeof(float) * h->bins, hipMemcpyHostToDevice));
        if (h->bins == 32768) {
            ro::stft32k_window_layout(h->window.data(), wk.data());
            CREATE_TRY(hipMalloc(&h->d_window_k32, sizeof(float) * h->bins));
            CREATE_TRY(hipMemcpy(h->d_window_k32, wk.data(), sizeof(float) * h->bins, hipMemcpyHostToDevice));
        }
    }
    if (!tw.empty())
        CREATE_TRY(hipMemcpy(h->d_twiddles, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice));
    if (plan_bins) {
        const int units = ro::stft_packed_twiddle_count(plan_bins);
        std::vector<float4> pk((size_t)std::max(units, 1));
        if (units > 0) ro::stft_pack_twiddles(plan_bins, tw.data(), pk.data());
        CREATE_TRY(hipMalloc(&h->d_twiddles_k, sizeof(float4) * pk.size()));
        CREATE_TRY(hipMemcpy(h->d_twiddles_k, pk.data(), sizeof(float4) * pk.size(), hipMemcpyHostToDevice));
    }
    if (h->f64) {
        std::vector<double2> full = build_full_twiddles_f64(h->bins);
        CREATE_TRY(hipMalloc(&h->d_tw_f64, sizeof(double2) * full.size()));
        CREATE_TRY(hipMemcpy(h->d_tw_f64, full.data(), sizeof(double2) * full.size(), hipMemcpyHostToDevice));
        h->f64reg = ro::f64reg_supported(h->bins);
#ifdef RO_DIAG
        if (const char *e = getenv("RO_F64_HBM")) h->f64reg = h->f64reg && atoi(e) == 0;   // the through-HBM passes, for A/B
#endif
        if (h->f64reg) {
            ro::F64RegTables t;
            ro::f64reg_tables(h->bins, h->window.data(), t);
            CREATE_TRY(hipMalloc(&h->d_f64r_window, sizeof(float) * t.window_k.size()));
            CREATE_TRY(hipMemcpy(h->d_f64r_window, t.window_k.data(), sizeof(float) * t.window_k.size(), hipMemcpyHostToDevice));
            const std::vector<double2> *tabs[4] = {&t.tw0, &t.tw1, &t.tw2, &t.tw3};
            for (int i = 0; i < 4; ++i) {
                if (tabs[i]->empty()) continue;
                CREATE_TRY(hipMalloc(&h->d_f64r_tw[i], sizeof(double2) * tabs[i]->size()));
                CREATE_TRY(hipMemcpy(h->d_f64r_tw[i], tabs[i]->data(), sizeof(double2) * tabs[i]->size(), hipMemcpyHostToDevice));
            }
        }
    }
    if (h->big && h->dec > 1) {
        // the rotations W_bins^(q m), [dec][sub_bins], each rounded once from long double
        std::vector<float2> tc((size_t)h->bins);
        const long double two_pi = 8.0L * atanl(1.0L);
        for (int r = 0; r < h->dec; ++r)
            for (int k = 0; k < h->sub_bins; ++k) {
                const long double ang = -two_pi * (long double)((long long)r * k) / (long double)h->bins;
                tc[(size_t)r * h->sub_bins + k] = make_float2((float)cosl(ang), (float)sinl(ang));
            }
        CREATE_TRY(hipMalloc(&h->d_tw_combine, sizeof(float2) * tc.size()));
        CREATE_TRY(hipMemcpy(h->d_tw_combine, tc.data(), sizeof(float2) * tc.size(), hipMemcpyHostToDevice));
    }
    if (h->big && h->dec > 1) {
        std::vector<float> ones((size_t)h->sub_bins, 1.0f);
        CREATE_TRY(hipMalloc(&h->d_ones, sizeof(float) * ones.size()));
        CREATE_TRY(hipMemcpy(h->d_ones, ones.data(), sizeof(float) * ones.size(), hipMemcpyHostToDevice));
    }
    if (h->dif) {
        std::vector<float> wk((size_t)h->bins);
        for (int r = 0; r < h->dec; ++r)
            if (!ro::stft_window_layout(h->sub_bins, h->window.data() + (size_t)r * h->sub_bins,
                                        wk.data() + (size_t)r * h->sub_bins)) {
                ro_stft_destroy(h);
                return fail(RO_ERR_UNSUPPORTED, "no window layout for bins=%d", h->sub_bins);
            }
        CREATE_TRY(hipMalloc(&h->d_window_dif, sizeof(float) * wk.size()));
        CREATE_TRY(hipMemcpy(h->d_window_dif, wk.data(), sizeof(float) * wk.size(), hipMemcpyHostToDevice));
        std::vector<float2> td((size_t)h->dec);
        const long double two_pi = 8.0L * atanl(1.0L);
        for (int j = 0; j < h->dec; ++j) {
            // (quarter turns exact: cosl(pi/2) is 6e-20, not 0)
            const long double ang = -two_pi * (long double)j / (long double)h->dec;
            long double c = cosl(ang), sn = sinl(ang);
            if ((4 * j) % h->dec == 0) { c = roundl(c); sn = roundl(sn); }
            td[(size_t)j] = make_float2((float)c, (float)sn);
        }
        CREATE_TRY(hipMalloc(&h->d_dif_tw, sizeof(float2) * td.size()));
        CREATE_TRY(hipMemcpy(h->d_dif_tw, td.data(), sizeof(float2) * td.size(), hipMemcpyHostToDevice));
        // residue q's rotation W_bins^(m q) as a shift of the bin index by q / dec: per stage (32, 32 x 32, 32^3 points
        // behind it) the powers 1, 2, 4, 8, 16 of exp(-2 pi i (q / dec) / M), each rounded once from long double
        std::vector<float2> ts((size_t)h->dec * 16, make_float2(1.0f, 0.0f));
        const long double span[3] = {32.0L, 1024.0L, 32768.0L};
        for (int q = 0; q < h->dec; ++q)
            for (int st = 0; st < 3; ++st)
                for (int i = 0; i < 5; ++i) {
                    const long double ang = -two_pi * ((long double)q / (long double)h->dec) * (long double)(1 << i) / span[st];
                    ts[(size_t)q * 16 + st * 5 + i] = make_float2((float)cosl(ang), (float)sinl(ang));
                }
        CREATE_TRY(hipMalloc(&h->d_dif_shift, sizeof(float2) * ts.size()));
        CREATE_TRY(hipMemcpy(h->d_dif_shift, ts.data(), sizeof(float2) * ts.size(), hipMemcpyHostToDevice));
    }
    if (h->four) {
        std::vector<float> wa;
        std::vector<float2> ta, tb, tr;
        ro::fourstep_tables(h->bins, h->window.data(), wa, ta, tb, tr);
        CREATE_TRY(hipMalloc(&h->d_four_window, sizeof(float) * wa.size()));
        CREATE_TRY(hipMemcpy(h->d_four_window, wa.data(), sizeof(float) * wa.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&h->d_four_tw_a, sizeof(float2) * ta.size()));
        CREATE_TRY(hipMemcpy(h->d_four_tw_a, ta.data(), sizeof(float2) * ta.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&h->d_four_tw_b, sizeof(float2) * tb.size()));
        CREATE_TRY(hipMemcpy(h->d_four_tw_b, tb.data(), sizeof(float2) * tb.size(), hipMemcpyHostToDevice));
        CREATE_TRY(hipMalloc(&h->d_four_tw_r, sizeof(float2) * tr.size()));
        CREATE_TRY(hipMemcpy(h->d_four_tw_r, tr.data(), sizeof(float2) * tr.size(), hipMemcpyHostToDevice));
    }
    if (h->czt) {
        const int rc = czt_setup(h);
        if (rc != RO_OK) { ro_stft_destroy(h); return rc; }
    }
#undef CREATE_TRY

    // what travels to the host in the streaming path: the tile when one is configured, else whole rows
    h->out_first = cfg->tile_cols > 0 ? cfg->tile_first_col : 0;
    h->out_cols = cfg->tile_cols > 0 ? cfg->tile_cols : h->bins;
    // streaming buffers are allocated lazily by the first push
    h->batch_rows = cfg->max_batch_rows > 0 ? cfg->max_batch_rows
                                            : std::max(1, (64 << 20) / (h->bins * 4));   // ~64 MiB of rows
#ifdef RO_DIAG_KNOBS
    if (const char *e = getenv("RO_GRAPH_TIME_EVERY")) h->diag_time_every = atoi(e);       // tools/r5/host_calls_ab.py
    if (const char *e = getenv("RO_GRAPH_TWO_EVENTS")) h->diag_done_only = atoi(e) == 0;
    if (const char *e = getenv("RO_GRAPH_DIRECT")) h->diag_direct = atoi(e);
#endif
    *out = h;
    return RO_OK;
}

extern "C" int ro_stft_destroy(ro_stft_t *h)
{
    if (!h) return RO_OK;
    (void)hipSetDevice(h->device);
    if (h->s_in) (void)hipStreamSynchronize(h->s_in);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->s_out) (void)hipStreamSynchronize(h->s_out);
    for (ro_stft::Slot &sl : h->slot)                       // graphed batches and their downloads run on the slots' own streams
        if (sl.gstream) (void)hipStreamSynchronize(sl.gstream);
    free_stream_slots(h);
    if (h->d_window) (void)hipFree(h->d_window);
    if (h->d_window_k) (void)hipFree(h->d_window_k);
    if (h->d_window_k32) (void)hipFree(h->d_window_k32);
    if (h->d_ln_keys) (void)hipFree(h->d_ln_keys);
    if (h->d_twiddles) (void)hipFree(h->d_twiddles);
    if (h->d_twiddles_k) (void)hipFree(h->d_twiddles_k);
    if (h->d_stamps) (void)hipFree(h->d_stamps);
    while (!h->ready.empty()) {
        destroy_batch(h->ready.front());
        h->ready.pop_front();
    }
    for (Batch *b : h->batch_pool) destroy_batch(b);
    if (h->d_tw_combine) (void)hipFree(h->d_tw_combine);
    if (h->d_spec) (void)hipFree(h->d_spec);
    if (h->d_window_dif) (void)hipFree(h->d_window_dif);
    if (h->inner) (void)ro_stft_destroy(h->inner);
    if (h->d_cw) (void)hipFree(h->d_cw);
    if (h->d_bc) (void)hipFree(h->d_bc);
    if (h->d_czt_a) (void)hipFree(h->d_czt_a);
    if (h->d_czt_A) (void)hipFree(h->d_czt_A);
    if (h->d_czt_mag) (void)hipFree(h->d_czt_mag);
    if (h->d_spec2) (void)hipFree(h->d_spec2);
    if (h->d_four_window) (void)hipFree(h->d_four_window);
    if (h->d_four_tw_a) (void)hipFree(h->d_four_tw_a);
    if (h->d_four_tw_b) (void)hipFree(h->d_four_tw_b);
    if (h->d_four_tw_r) (void)hipFree(h->d_four_tw_r);
    if (h->d_four_z) (void)hipFree(h->d_four_z);
    if (h->d_ones) (void)hipFree(h->d_ones);
    if (h->d_dif_tw) (void)hipFree(h->d_dif_tw);
    if (h->d_dif_shift) (void)hipFree(h->d_dif_shift);
    if (h->d_tw_f64) (void)hipFree(h->d_tw_f64);
    if (h->d_f64r_window) (void)hipFree(h->d_f64r_window);
    for (int i = 0; i < 4; ++i)
        if (h->d_f64r_tw[i]) (void)hipFree(h->d_f64r_tw[i]);
    if (h->d_ln_part) (void)hipFree(h->d_ln_part);
    for (int i = 0; i < 2; ++i)
        if (h->d_scratch_d[i]) (void)hipFree(h->d_scratch_d[i]);
    if (h->d_f64_ring) (void)hipFree(h->d_f64_ring);
    if (h->d_f64_ctl) (void)hipFree(h->d_f64_ctl);
    if (h->h_f64_err) (void)hipHostFree(h->h_f64_err);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return RO_OK;
}

#ifdef RO_DIAG
// Diagnostic builds only (-DRO_DIAG=1; not in include/ro_stft.h, not in the product library): allocate / read the per-wave
// phase stamps that a -DRO_STAMPS=1 / -DRO_STAMPS32K=1 / -DRO_F64R_STAMPS=1 build of the kernels fills.
extern "C" int ro_stft_debug_stamps(ro_stft_t *h, unsigned long long *out, int max_words)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    const int words = 4096 * 16;
    if (!h->d_stamps) {
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMalloc(&h->d_stamps, sizeof(unsigned long long) * words));
        HIP_TRY(hipMemset(h->d_stamps, 0, sizeof(unsigned long long) * words));
        return RO_OK;
    }
    if (out) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, h->d_stamps, sizeof(unsigned long long) * std::min(words, max_words),
                          hipMemcpyDeviceToHost));
    }
    return RO_OK;
}
#endif

extern "C" int ro_stft_get_window(const ro_stft_t *h, float *out)
{
    if (!h || !out) return fail(RO_ERR_INVALID, "null argument");
    std::memcpy(out, h->window.data(), sizeof(float) * h->bins);
    return RO_OK;
}

extern "C" int ro_stft_hop(const ro_stft_t *h) { return h ? h->hop : fail(RO_ERR_INVALID, "null handle"); }
extern "C" int ro_stft_bins(const ro_stft_t *h) { return h ? h->bins : fail(RO_ERR_INVALID, "null handle"); }

extern "C" int ro_stft_device_name(const ro_stft_t *h, char *buf, size_t len)
{
    if (!h || !buf || len == 0) return fail(RO_ERR_INVALID, "null argument");
    std::snprintf(buf, len, "%s", h->device_name.c_str());
    return RO_OK;
}

extern "C" int ro_stft_set_bands(ro_stft_t *h, const ro_bands_t *bands)
{
    if (!h || !bands) return fail(RO_ERR_INVALID, "null argument");
    int rc = check_bands(h, *bands);
    if (rc != RO_OK) return rc;
    h->cfg.bands = *bands;
    h->cfg.enable_scan = 1;
    return RO_OK;
}

// ---------------------------------------------------------------------------
// resident path
// ---------------------------------------------------------------------------
extern "C" int ro_stft_run_resident(ro_stft_t *h, const void *d_iq, int format, int64_t samples,
                                    int64_t first_row, int64_t rows, float *d_rows, int64_t row_stride,
                                    float *d_tile, ro_scan_record_t *d_records, void *stream)
{
    int rc = validate_resident(h, d_iq, format, samples, first_row, rows, d_rows, row_stride, d_tile,
                               d_records);
    if (rc != RO_OK || rows == 0) return rc;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;      // NULL = the default (null) stream, like any HIP launch
    rc = launch_transform(h, d_iq, format, first_row, rows, d_rows, row_stride, s, d_tile, d_records);
    if (rc != RO_OK) return rc;
    rc = launch_tile_and_scan(h, d_rows, row_stride, rows, d_tile, d_records, s);
    if (rc != RO_OK) return rc;
    h->stat_launches += 1;
    h->stat_rows += rows;
    return RO_OK;
}

extern "C" int ro_stft_run_resident_ln(ro_stft_t *h, const void *d_iq, int format, int64_t samples, int64_t first_row,
                                       int64_t rows, float *d_rows, int64_t row_stride, float *d_tile, float *d_ln_tile,
                                       float *d_ln_minmax, ro_scan_record_t *d_records, void *stream)
{
    int rc = validate_resident(h, d_iq, format, samples, first_row, rows, d_rows, row_stride, d_tile, d_records);
    if (rc != RO_OK || rows == 0) return rc;
    if (!h->cfg.tile_ln) return fail(RO_ERR_STATE, "this handle was not created with tile_ln");
    if (!d_tile) return fail(RO_ERR_INVALID, "the log image is cut from the tile: d_tile is required");
    if (d_ln_minmax && !d_ln_tile) return fail(RO_ERR_INVALID, "the range comes with the log image: pass d_ln_tile too");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    // the fused epilogue always writes the log when asked for its range (the partial min / max come with it)
    float *ln = d_ln_tile;
    rc = launch_transform(h, d_iq, format, first_row, rows, d_rows, row_stride, s, d_tile, d_records, ln);
    if (rc != RO_OK) return rc;
    rc = launch_tile_and_scan(h, d_rows, row_stride, rows, d_tile, d_records, s, ln, d_ln_minmax);
    if (rc != RO_OK) return rc;
    h->stat_launches += 1;
    h->stat_rows += rows;
    return RO_OK;
}

extern "C" int ro_stft_spectra_resident(ro_stft_t *h, const void *d_iq, int format, int64_t samples,
                                        int64_t first_row, int64_t rows, float *d_spectra, int64_t stride,
                                        void *stream)
{
    // same argument checks as the magnitude path (d_spectra in the place of d_rows)
    int rc = validate_resident(h, d_iq, format, samples, first_row, rows, d_spectra, stride, nullptr, nullptr);
    if (rc != RO_OK) return rc;
    if (h->czt) return fail(RO_ERR_UNSUPPORTED, "complex spectra are available for power-of-two bins");
    HIP_TRY(hipSetDevice(h->device));
    if (h->f64) {
        // the double transform itself, narrowed once per component (ro_f64reg.hip's SPEC instantiations): the register
        // kernel's sizes only
        if (!h->f64reg)
            return fail(RO_ERR_UNSUPPORTED, "complex spectra of RO_PRECISION_F64 handles are available up to 65536 bins");
        ro::F64RegArgs r{};
        r.iq = d_iq;
        r.window_k = h->d_f64r_window;
        r.tw0 = h->d_f64r_tw[0];
        r.tw1 = h->d_f64r_tw[1];
        r.tw2 = h->d_f64r_tw[2];
        r.tw3 = h->d_f64r_tw[3];
        r.rows_out = d_spectra;
        r.first_row = first_row;
        r.rows = rows;
        r.row_stride = stride;
        r.hop = h->hop;
        r.gain = h->cfg.iq_gain;
        r.spectra = 1;
        HIP_TRY(ro::launch_f64reg(h->bins, format, r, (hipStream_t)stream));
        return RO_OK;
    }
    if (h->big)
        return launch_spectra_big(h, d_iq, format, first_row, rows, reinterpret_cast<float2 *>(d_spectra), stride,
                                  (hipStream_t)stream);
    ro::StftArgs a = make_stft_args(h, d_iq, first_row, rows, nullptr, 0);
    a.spec_out = reinterpret_cast<float2 *>(d_spectra);
    a.spec_stride = stride;
    HIP_TRY(ro::launch_stft(h->bins, format, a, (hipStream_t)stream));
    return RO_OK;
}

extern "C" int ro_stft_scan_resident(ro_stft_t *h, const float *d_rows, int64_t row_stride, int64_t rows,
                                     ro_scan_record_t *d_records, void *stream)
{
    if (!h || !d_rows || !d_records) return fail(RO_ERR_INVALID, "null argument");
    if (!h->cfg.enable_scan) return fail(RO_ERR_STATE, "scan bands not configured");
    if (rows < 0 || row_stride < h->bins) return fail(RO_ERR_INVALID, "bad rows / row_stride");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;      // NULL = the default (null) stream, like any HIP launch
    ro::ScanArgs sc = make_scan_args(h, d_rows, row_stride, rows, d_records);
    HIP_TRY(ro::launch_scan(sc, s));
    return RO_OK;
}

extern "C" int ro_stft_ln_tile_resident(ro_stft_t *h, const float *d_rows, int64_t row_stride, int64_t rows,
                                        int first_col, int cols, float *d_ln, uint8_t *d_u8, float *d_minmax,
                                        void *stream)
{
    if (!h || !d_rows) return fail(RO_ERR_INVALID, "null argument");
    if (!d_ln && !d_u8 && !d_minmax) return fail(RO_ERR_INVALID, "no output requested");
    if (rows < 0 || row_stride < h->bins) return fail(RO_ERR_INVALID, "bad rows / row_stride");
    if (first_col < 0 || cols <= 0 || (int64_t)first_col + cols > h->bins)
        return fail(RO_ERR_INVALID, "columns outside the row");
    HIP_TRY(hipSetDevice(h->device));
    ro::LnArgs a;
    a.rows_in = d_rows;
    a.ln_out = d_ln;
    a.u8_out = d_u8;
    a.keys = h->d_ln_keys + 2 * (h->ln_calls++ & 15);      // a pair of its own for each of 16 calls in flight
    a.minmax = d_minmax;
    a.rows = rows;
    a.row_stride = row_stride;
    a.first = first_col;
    a.cols = cols;
    HIP_TRY(ro::launch_ln_tile(a, (hipStream_t)stream));
    return RO_OK;
}

extern "C" int ro_stft_time_resident(ro_stft_t *h, const void *d_iq, int format, int64_t samples,
                                     int64_t first_row, int64_t rows, float *d_rows, int64_t row_stride,
                                     float *d_tile, ro_scan_record_t *d_records, void *stream, int iters,
                                     float *ms_out, float *kernel_ms_out)
{
    int rc = validate_resident(h, d_iq, format, samples, first_row, rows, d_rows, row_stride, d_tile,
                               d_records);
    if (rc != RO_OK) return rc;
    if (iters <= 0 || !ms_out) return fail(RO_ERR_INVALID, "iters must be positive and ms_out non-null");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;      // NULL = the default (null) stream, like any HIP launch
    std::vector<hipEvent_t> ev((size_t)iters * 3, nullptr);
    for (auto &e : ev) {
        if (hipEventCreate(&e) != hipSuccess) {
            for (auto &d : ev) if (d) (void)hipEventDestroy(d);
            return fail(RO_ERR_HIP, "hipEventCreate failed");
        }
    }
    for (int i = 0; i < iters && rc == RO_OK; ++i) {
        hipError_t e = hipEventRecord(ev[3 * i], s);
        if (e == hipSuccess) {
            rc = launch_transform(h, d_iq, format, first_row, rows, d_rows, row_stride, s, d_tile, d_records);
            if (rc != RO_OK) break;
            e = hipEventRecord(ev[3 * i + 1], s);
        }
        if (e == hipSuccess) {
            rc = launch_tile_and_scan(h, d_rows, row_stride, rows, d_tile, d_records, s);
            if (rc != RO_OK) break;
            e = hipEventRecord(ev[3 * i + 2], s);
        }
        if (e != hipSuccess) rc = fail(RO_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(e));
    }
    if (rc != RO_OK) {                                  // leave no event behind on the error paths
        (void)hipStreamSynchronize(s);
        for (auto &e : ev) (void)hipEventDestroy(e);
        return rc;
    }
    hipError_t es = hipStreamSynchronize(s);
    double k0 = 0.0, k1 = 0.0;
    for (int i = 0; i < iters && es == hipSuccess; ++i) {
        float t = 0.f, t0 = 0.f, t1 = 0.f;
        if ((es = hipEventElapsedTime(&t, ev[3 * i], ev[3 * i + 2])) != hipSuccess) break;
        if ((es = hipEventElapsedTime(&t0, ev[3 * i], ev[3 * i + 1])) != hipSuccess) break;
        if ((es = hipEventElapsedTime(&t1, ev[3 * i + 1], ev[3 * i + 2])) != hipSuccess) break;
        ms_out[i] = t;
        k0 += t0;
        k1 += t1;
    }
    if (es != hipSuccess) {
        for (auto &e : ev) (void)hipEventDestroy(e);
        return fail(RO_ERR_HIP, "timing the resident path failed: %s", hipGetErrorString(es));
    }
    if (kernel_ms_out) {
        kernel_ms_out[0] = (float)(k0 / iters);
        kernel_ms_out[1] = (float)(k1 / iters);
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    h->stat_launches += iters;
    h->stat_rows += rows * (int64_t)iters;
    return RO_OK;
}

extern "C" int ro_stft_timing(ro_stft_t *h, ro_stft_timing_t *out, int reset)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    if (out) {
        *out = h->timing;
        out->push_ms_avg = h->timing.push_calls ? h->push_ms_sum / (double)h->timing.push_calls : 0.0;
        out->batch_gpu_ms_avg = h->timed_batches ? h->batch_ms_sum / (double)h->timed_batches : 0.0;
        out->row_gpu_us_avg = h->timed_rows ? h->batch_ms_sum * 1e3 / (double)h->timed_rows : 0.0;
        out->fetch_ms_avg = h->timing.fetch_calls ? h->fetch_ms_sum / (double)h->timing.fetch_calls : 0.0;
    }
    if (reset) {                                             // FFTBackend::clearProcessingTime, src/FFTBackend.h:231-235
        h->timing = ro_stft_timing_t{};
        h->push_ms_sum = h->batch_ms_sum = h->fetch_ms_sum = 0.0;
        h->timed_batches = h->timed_rows = 0;
    }
    return RO_OK;
}

extern "C" int ro_stft_stats(const ro_stft_t *h, int64_t *samples_in, int64_t *rows_out, int64_t *launches,
                             double *kernel_ms_total)
{
    if (!h) return fail(RO_ERR_INVALID, "null handle");
    if (samples_in) *samples_in = h->stat_samples;
    if (rows_out) *rows_out = h->stat_rows;
    if (launches) *launches = h->stat_launches;
    if (kernel_ms_total) *kernel_ms_total = h->stat_kernel_ms;
    return RO_OK;
}
