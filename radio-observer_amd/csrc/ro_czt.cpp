// ro_czt.cpp -- FFTW takes any length (src/FFTBackend.cpp:120); here an even length that is not a power of two is Bluestein's
// chirp-z form on an inner handle of the power-of-two length M >= 2 bins - 1 (ro::CztArgs in ro_kernels.h has the algebra):
// the tables, the inner handle, and the launch sequence.
#include "ro_host.h"

using namespace ro::host;

namespace {

// in-place forward FFT of a power-of-two length in double (table preparation only: the chirp-z filter)
void host_fft(std::vector<std::complex<double>> &x)
{
    const size_t n = x.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(x[i], x[j]);
    }
    const long double two_pi = 8.0L * atanl(1.0L);
    for (size_t len = 2; len <= n; len <<= 1) {
        std::vector<std::complex<double>> w(len / 2);
        for (size_t k = 0; k < len / 2; ++k) {
            const long double ang = -two_pi * (long double)k / (long double)len;
            w[k] = std::complex<double>((double)cosl(ang), (double)sinl(ang));
        }
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const std::complex<double> u = x[i + k], v = x[i + k + len / 2] * w[k];
                x[i + k] = u + v;
                x[i + k + len / 2] = u - v;
            }
    }
}

}  // namespace

namespace ro {
namespace host {

// lengths that are not a power of two run as a chirp-z transform on the power-of-two length M >= 2 bins - 1 <= 2^20.
// Even lengths only: for an odd size the reference's processFFT leaves the last column of the row unwritten and
// writes one column twice (src/WaterfallBackend.cpp:489-505, halfSize = size / 2) -- there is no defined result to match.
int czt_length(int bins)
{
    if (bins < 256 || bins >= (1 << 19) || (bins & 1) || (bins & (bins - 1)) == 0) return 0;
    int m = 512;
    while (m < 2 * bins - 1) m <<= 1;
    return m;
}

// the inner handle (length M, overlap 0, a window of ones, no bands / tile) and the chirp tables; on failure the caller
// destroys h, which frees whatever exists by then
int czt_setup(ro_stft *h)
{
    const int N = h->bins, M = h->czt_m;
    // the inner handle: length M, overlap 0, a window of ones, no bands / tile
    {
        std::vector<float> ones((size_t)M, 1.0f);
        ro_stft_config_t ic{};
        ic.struct_size = sizeof ic;
        ic.bins = M;
        ic.overlap = 0;
        ic.sample_rate = h->cfg.sample_rate;
        ic.window_kind = RO_WINDOW_CUSTOM;
        ic.window_table = ones.data();
        ic.device = h->device;
        ic.spare_cus_per_xcd = h->cfg.spare_cus_per_xcd;
        int rc = ro_stft_create(&ic, &h->inner);
        if (rc != RO_OK) return rc;
    }
    // chirp c[i] = exp(-pi i i^2 / N), the angle reduced exactly: i^2 mod 2N in integers
    const long double pi = 4.0L * atanl(1.0L);
    auto chirp = [&](int64_t i) {
        const int64_t r = (i * i) % (2 * (int64_t)N);
        const long double ang = -pi * (long double)r / (long double)N;
        return std::complex<double>((double)cosl(ang), (double)sinl(ang));
    };
    std::vector<float2> cw((size_t)N);
    for (int i = 0; i < N; ++i) {
        const std::complex<double> c = chirp(i) * (double)h->window[(size_t)i];
        cw[(size_t)i] = make_float2((float)c.real(), (float)c.imag());
    }
    // B = FFT_M(conj(c) wrapped around M) in double on the host, once; the kernels use conj(B) / M
    std::vector<std::complex<double>> b((size_t)M, std::complex<double>(0.0, 0.0));
    b[0] = std::conj(chirp(0));
    for (int i = 1; i < N; ++i) b[(size_t)i] = b[(size_t)(M - i)] = std::conj(chirp(i));
    host_fft(b);
    std::vector<float2> bc((size_t)M);
    for (int i = 0; i < M; ++i)
        bc[(size_t)i] = make_float2((float)(b[(size_t)i].real() / M), (float)(-b[(size_t)i].imag() / M));
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMalloc(&h->d_cw, sizeof(float2) * cw.size()));
    HIP_TRY(hipMemcpy(h->d_cw, cw.data(), sizeof(float2) * cw.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&h->d_bc, sizeof(float2) * bc.size()));
    HIP_TRY(hipMemcpy(h->d_bc, bc.data(), sizeof(float2) * bc.size(), hipMemcpyHostToDevice));
    return RO_OK;
}

// a length that is not a power of two: chirp-z on the inner handle (see ro::CztArgs), in chunks that fit the scratch
int launch_transform_czt(ro_stft *h, const void *d_iq, int format, int64_t first_row, int64_t rows, float *d_rows,
                         int64_t row_stride, hipStream_t s)
{
    ro_stft *in = h->inner;
    const int M = h->czt_m;
    if (!h->d_czt_mag) {                                            // (each block on its own: a failed call can be retried)
        h->czt_rows = std::min<int64_t>(65535, std::max<int64_t>(1, ((int64_t)1 << 30) / ((int64_t)M * 8)));
        if (!h->d_czt_a) HIP_TRY(hipMalloc(&h->d_czt_a, (size_t)h->czt_rows * M * sizeof(float2)));
        if (!h->d_czt_A) HIP_TRY(hipMalloc(&h->d_czt_A, (size_t)h->czt_rows * M * sizeof(float2)));
        HIP_TRY(hipMalloc(&h->d_czt_mag, (size_t)h->czt_rows * M * sizeof(float)));
    }
    for (int64_t done = 0; done < rows; done += h->czt_rows) {
        const int64_t n = std::min(h->czt_rows, rows - done);
        ro::CztArgs c{};
        c.iq = d_iq;
        c.cw = h->d_cw;
        c.bc = h->d_bc;
        c.a = h->d_czt_a;
        c.first_row = first_row + done;
        c.rows = n;
        c.row_stride = row_stride;
        c.hop = h->hop;
        c.n = h->bins;
        c.m = M;
        c.gain = (float)h->cfg.iq_gain;
        HIP_TRY(ro::launch_czt_pre(format, c, s));
        // A = FFT_M(a): the inner handle's rows are the M-sample blocks of d_czt_a (overlap 0, a window of ones)
        if (!in->big) {
            ro::StftArgs a = make_stft_args(in, h->d_czt_a, 0, n, nullptr, 0);
            a.spec_out = h->d_czt_A;
            a.spec_stride = M;
            HIP_TRY(ro::launch_stft(M, RO_FMT_F32, a, s));
        } else {
            int rc = launch_spectra_big(in, h->d_czt_a, RO_FMT_F32, 0, n, h->d_czt_A, M, s);
            if (rc != RO_OK) return rc;
        }
        c.a = h->d_czt_A;
        HIP_TRY(ro::launch_czt_mul(c, s));                          // conj(A B) / M, in place
        int rc = launch_transform(in, h->d_czt_A, RO_FMT_F32, 0, n, h->d_czt_mag, M, s, nullptr, nullptr, nullptr);
        if (rc != RO_OK) return rc;
        c.mag = h->d_czt_mag;
        c.rows_out = d_rows + done * row_stride;
        HIP_TRY(ro::launch_czt_out(c, s));
    }
    return RO_OK;
}

}  // namespace host
}  // namespace ro
