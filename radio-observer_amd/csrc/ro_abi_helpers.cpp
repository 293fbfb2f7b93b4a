// ro_abi_helpers.cpp -- the part of the C ABI (include/ro_stft.h) that needs no handle: the error text, FFTBackend's public
// arithmetic, the window tables, the shard arithmetic of the time-chunk split and its exchange schedule, pinned memory.
#include "ro_host.h"

namespace ro {
namespace host {

namespace {
thread_local std::string g_error;
}

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

const char *last_error_text() { return g_error.c_str(); }

// window tables: same arithmetic as FFTBackend::startStream (src/FFTBackend.cpp:156-186):
// float coefficients, double pi = 4*atan(1), (float)i and (float)(bins-1) widened to double,
// evaluation in double, one narrowing on store.
void build_window(int kind, int bins, float *w)
{
    const double pi = 4.0 * std::atan(1.0);
    const double denom = (double)(float)(bins - 1);
    if (kind == RO_WINDOW_HANN) {
        for (int i = 0; i < bins; ++i)
            w[i] = (float)(0.5 * (1.0 - std::cos(2.0 * pi * (double)(float)i / denom)));
        return;
    }
    const float a0 = 0.355768f, a1 = 0.487396f, a2 = 0.144232f, a3 = 0.012604f;
    for (int i = 0; i < bins; ++i) {
        const double x = (double)(float)i;
        w[i] = (float)((double)a0 - (double)a1 * std::cos(2.0 * pi * x / denom) +
                       (double)a2 * std::cos(4.0 * pi * x / denom) -
                       (double)a3 * std::cos(6.0 * pi * x / denom));
    }
}

}  // namespace host
}  // namespace ro

using namespace ro::host;

// ---------------------------------------------------------------------------
// library
// ---------------------------------------------------------------------------
extern "C" int ro_abi_version(void) { return RO_ABI_VERSION; }

extern "C" const char *ro_last_error(void) { return last_error_text(); }

extern "C" int ro_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RO_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

// ---------------------------------------------------------------------------
// host helpers (FFTBackend's scalar arithmetic; float/double mixing as in the reference)
// ---------------------------------------------------------------------------
extern "C" int ro_clamp_overlap(int bins, int overlap)
{
    if (overlap < 0) return 0;                       // src/FFTBackend.cpp:108
    if (overlap >= bins) return bins - 1;            // :109
    return overlap;
}

extern "C" float ro_fft_sample_rate(int sample_rate, int bins, int overlap)
{
    return (float)sample_rate / (float)(bins - ro_clamp_overlap(bins, overlap));   // :150-151
}

extern "C" int ro_frequency_to_bin(int bins, int sample_rate, float frequency)
{
    // src/FFTBackend.h:169-175: float quotient, double sum and product, truncation, clamp
    const float sr = (float)sample_rate, n = (float)bins;
    const int bin = (int)((double)n * ((double)(frequency / sr) + 0.5));
    if (bin < 0) return 0;
    if (bin >= bins) return bins - 1;
    return bin;
}

extern "C" float ro_bin_to_frequency(int bins, int sample_rate, int bin)
{
    // src/FFTBackend.h:141-145: float quotient, the rest in double, narrowed on return
    const float b = (float)bin, sr = (float)sample_rate, n = (float)bins;
    return (float)((double)sr * (-0.5 + (double)(b / n)));
}

extern "C" int ro_time_to_fft_samples(double seconds, float fft_sample_rate)
{
    return (int)(seconds * (double)fft_sample_rate);             // src/FFTBackend.h:197-200
}

extern "C" int64_t ro_row_count(int64_t samples, int bins, int overlap)
{
    const int64_t hop = bins - ro_clamp_overlap(bins, overlap);
    if (samples < bins) return 0;
    return (samples - bins) / hop + 1;
}

extern "C" int ro_window_table(int kind, int bins, float *out)
{
    if (!out || bins < 2) return fail(RO_ERR_INVALID, "ro_window_table: bad arguments");
    if (kind != RO_WINDOW_NUTTALL && kind != RO_WINDOW_HANN)
        return fail(RO_ERR_INVALID, "ro_window_table: kind %d has no formula", kind);
    build_window(kind, bins, out);
    return RO_OK;
}

// ---------------------------------------------------------------------------
// time-chunk sharding (host arithmetic; the Python side, timeshard.py, calls these)
// ---------------------------------------------------------------------------
extern "C" int ro_shard_rows(int64_t total_rows, int world, int rank, int64_t *first_row, int64_t *rows)
{
    if (total_rows < 0 || world < 1 || rank < 0 || rank >= world || !first_row || !rows)
        return fail(RO_ERR_INVALID, "ro_shard_rows: bad arguments");
    // 128-bit products: rank * total_rows overflows int64 only for absurd sizes, but costs nothing to rule out
    const int64_t lo = (int64_t)(((__int128)rank * total_rows) / world);
    const int64_t hi = (int64_t)(((__int128)(rank + 1) * total_rows) / world);
    *first_row = lo;
    *rows = hi - lo;
    return RO_OK;
}

extern "C" int ro_shard_samples(int64_t first_row, int64_t rows, int bins, int overlap, int64_t *first_sample,
                                int64_t *samples)
{
    if (first_row < 0 || rows < 0 || bins < 2 || !first_sample || !samples)
        return fail(RO_ERR_INVALID, "ro_shard_samples: bad arguments");
    const int64_t hop = bins - ro_clamp_overlap(bins, overlap);
    *first_sample = first_row * hop;
    *samples = rows > 0 ? (rows - 1) * hop + bins : 0;
    return RO_OK;
}

extern "C" int64_t ro_shard_max_rows(int64_t total_rows, int world)
{
    if (total_rows < 0 || world < 1) return fail(RO_ERR_INVALID, "ro_shard_max_rows: bad arguments");
    return (total_rows + world - 1) / world;       // sizes differ by at most one: the largest is the ceiling
}

extern "C" int ro_direct_schedule(int world, int rank, int64_t total_rows, int k, int *to, int *from, int64_t *recv_first_row,
                                  int64_t *recv_rows)
{
    if (world < 1 || rank < 0 || rank >= world || total_rows < 0 || k < 0 || k >= world || !to || !from || !recv_first_row ||
        !recv_rows)
        return fail(RO_ERR_INVALID, "ro_direct_schedule: bad arguments");
    *to = (rank + k) % world;
    *from = (rank - k + world) % world;
    return ro_shard_rows(total_rows, world, *from, recv_first_row, recv_rows);
}

extern "C" int ro_stitch_rows(const void *gathered, int64_t total_rows, int world, size_t row_bytes, void *out)
{
    if (total_rows < 0 || world < 1 || (total_rows > 0 && (!gathered || !out)))
        return fail(RO_ERR_INVALID, "ro_stitch_rows: bad arguments");
    const int64_t block = ro_shard_max_rows(total_rows, world);
    const char *src = static_cast<const char *>(gathered);
    char *dst = static_cast<char *>(out);
    for (int g = 0; g < world; ++g) {
        int64_t first = 0, rows = 0;
        ro_shard_rows(total_rows, world, g, &first, &rows);
        std::memcpy(dst + (size_t)first * row_bytes, src + (size_t)g * (size_t)block * row_bytes,
                    (size_t)rows * row_bytes);
    }
    return RO_OK;
}

extern "C" int ro_ln_levels(const float *ln, int64_t count, float mn, float mx, uint8_t *levels_out)
{
    if (count < 0 || (count > 0 && (!ln || !levels_out))) return fail(RO_ERR_INVALID, "ro_ln_levels: bad arguments");
    const float span = mx - mn;
    for (int64_t i = 0; i < count; ++i) {
        // float32 throughout, like numpy in the viewer (fits2png:444-445); -inf = a zero pixel (dropped there)
        const float level = (ln[i] - mn) / span * 255.f;
        levels_out[i] = (std::isfinite(ln[i]) && span > 0.f) ? (uint8_t)(int)level : (uint8_t)0;
    }
    return RO_OK;
}

// 1: [p, p + bytes) is host memory page-locked by this process's HIP runtime (ro_pinned_alloc, hipHostMalloc,
// hipHostRegister) -- first and last byte are both known to the runtime as host allocations and lie in ONE mapping
// (equal distance in the runtime's view); 0: it is not (heap, stack, a numpy array, device memory, no device at all).
extern "C" int ro_pinned_check(const void *p, size_t bytes)
{
    if (!p || bytes == 0) return 0;
    const char *lo = static_cast<const char *>(p), *hi = lo + bytes - 1;
    hipPointerAttribute_t a0{}, a1{};
    const hipError_t e0 = hipPointerGetAttributes(&a0, lo);
    const hipError_t e1 = e0 == hipSuccess ? hipPointerGetAttributes(&a1, hi) : e0;
    if (e0 != hipSuccess || e1 != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return a0.type == hipMemoryTypeHost && a1.type == hipMemoryTypeHost && a0.hostPointer && a1.hostPointer &&
                   static_cast<const char *>(a1.hostPointer) - static_cast<const char *>(a0.hostPointer) == hi - lo
               ? 1 : 0;
}

extern "C" void *ro_pinned_alloc(int device, size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipSetDevice(device) != hipSuccess) return nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

extern "C" void ro_pinned_free(void *p)
{
    if (p) (void)hipHostFree(p);
}
