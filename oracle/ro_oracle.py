"""ctypes loader for the CPU oracle (oracle/ro_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.  PARITY UNPINNED:
see oracle/ro_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("RO_ORACLE_LIB") or os.path.join(_HERE, "libro_oracle.so")   # override: sanitizer builds


def build(force=False):
    """Compile oracle/ro_oracle.c with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "ro_oracle.c")
    hdr = os.path.join(_HERE, "ro_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "libro_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class RowInfo(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("time_sec", C.c_int64), ("time_usec", C.c_int64),
                ("raw_mark", C.c_int)]


class Scan(C.Structure):
    _fields_ = [("noise", C.c_float), ("peak", C.c_int), ("average", C.c_float)]


class Bands(C.Structure):
    _fields_ = [("low_detect", C.c_int), ("detect_width", C.c_int),
                ("low_noise", C.c_int), ("noise_width", C.c_int),
                ("advance", C.c_int), ("jitter", C.c_int), ("avg_bins", C.c_int),
                ("noise_metadata_rows", C.c_int)]


class Fsm(C.Structure):
    _fields_ = [("state", C.c_int), ("peak_freq", C.c_float), ("noise", C.c_float),
                ("magnitude", C.c_float), ("duration", C.c_int), ("snap_start", C.c_int),
                ("snap_length", C.c_int), ("advance", C.c_int), ("jitter", C.c_int),
                ("fft_sample_rate", C.c_float), ("sample_rate", C.c_int),
                ("min_detect_fq", C.c_float), ("max_detect_fq", C.c_float)]


class Event(C.Structure):
    _fields_ = [("fired", C.c_int), ("snap_start", C.c_int), ("snap_length", C.c_int),
                ("duration_s", C.c_float), ("noise", C.c_float), ("peak_freq", C.c_float),
                ("magnitude", C.c_float), ("fmin", C.c_float), ("fmax", C.c_float),
                ("raw_length", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    f32p = C.POINTER(C.c_float)
    f64p = C.POINTER(C.c_double)
    i64 = C.c_int64
    sig = {
        "ro_oracle_fft_sample_rate": (C.c_float, [C.c_int, C.c_int, C.c_int]),
        "ro_oracle_clamp_overlap": (C.c_int, [C.c_int, C.c_int]),
        "ro_oracle_frequency_to_bin": (C.c_int, [C.c_int, C.c_int, C.c_float]),
        "ro_oracle_bin_to_frequency": (C.c_float, [C.c_int, C.c_int, C.c_int]),
        "ro_oracle_time_to_fft_samples": (C.c_int, [C.c_double, C.c_float]),
        "ro_oracle_recorder_fft_samples_to_raw": (C.c_int, [C.c_int, C.c_float, C.c_int]),
        "ro_oracle_backend_fft_samples_to_raw": (C.c_int, [C.c_int, C.c_float, C.c_int]),
        "ro_oracle_window_nuttall": (None, [C.c_int, f32p]),
        "ro_oracle_window_hann": (None, [C.c_int, f32p]),
        "ro_oracle_row_count": (i64, [i64, C.c_int, C.c_int]),
        "ro_oracle_fft_f64": (C.c_int, [C.c_int, f64p, f64p]),
        "ro_oracle_use_fftw": (C.c_int, [C.c_int]),
        "ro_oracle_fftw_active": (C.c_int, []),
        "ro_oracle_fft_prepare": (C.c_int, [C.c_int]),
        "ro_oracle_dft_direct": (None, [C.c_int, f64p, f64p]),
        "ro_oracle_row": (C.c_int, [C.c_int, f64p, f32p, C.c_double, f32p, f64p]),
        "ro_oracle_stft": (i64, [f64p, i64, C.c_int, C.c_int, f32p, C.c_double, i64, i64, f32p]),
        "ro_oracle_stft_f32": (i64, [f32p, i64, C.c_int, C.c_int, f32p, C.c_double, i64, i64, f32p]),
        "ro_oracle_wftime_add": (None, [i64, i64, i64, i64, C.POINTER(i64), C.POINTER(i64)]),
        "ro_oracle_wftime_add_samples": (None, [i64, i64, C.c_uint64, C.c_int,
                                                C.POINTER(i64), C.POINTER(i64)]),
        "ro_oracle_stream_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int, i64, i64, C.c_double, C.c_int]),
        "ro_oracle_stream_destroy": (None, [C.c_void_p]),
        "ro_oracle_stream_process": (C.c_int, [C.c_void_p, f64p, C.c_int, f32p,
                                               C.POINTER(RowInfo), C.c_int]),
        "ro_oracle_stream_pending": (C.c_int, [C.c_void_p]),
        "ro_oracle_noise": (C.c_float, [f32p, C.c_int]),
        "ro_oracle_peak": (C.c_int, [f32p, C.c_int]),
        "ro_oracle_average": (C.c_float, [f32p, C.c_int]),
        "ro_oracle_scan_row": (None, [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(Scan)]),
        "ro_oracle_bolid_bands": (None, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                         C.c_float, C.c_float, C.c_double, C.c_double,
                                         C.c_float, C.c_double, C.POINTER(Bands)]),
        "ro_oracle_fsm_init": (None, [C.POINTER(Fsm), C.c_int, C.c_int, C.c_float, C.c_int,
                                      C.c_float, C.c_float]),
        "ro_oracle_fsm_update": (None, [C.POINTER(Fsm), C.c_float, C.c_float, C.c_float, C.c_int,
                                        C.POINTER(Event)]),
        "ro_oracle_ring2d_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int]),
        "ro_oracle_ring2d_destroy": (None, [C.c_void_p]),
        "ro_oracle_ring2d_capacity": (C.c_int, [C.c_void_p]),
        "ro_oracle_ring2d_chunk_rows": (C.c_int, [C.c_void_p]),
        "ro_oracle_ring2d_get_size": (C.c_int, [C.c_void_p]),
        "ro_oracle_ring2d_is_full": (C.c_int, [C.c_void_p]),
        "ro_oracle_ring2d_push": (C.c_int, [C.c_void_p]),
        "ro_oracle_ring2d_mark": (C.c_int, [C.c_void_p]),
        "ro_oracle_ring2d_normalize": (C.c_int, [C.c_void_p, C.c_int]),
        "ro_oracle_ring2d_size_from": (C.c_int, [C.c_void_p, C.c_int]),
        "ro_oracle_ring2d_size_between": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
        "ro_oracle_ring2d_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
        "ro_oracle_ring2d_free_reservation": (C.c_int, [C.c_void_p, C.c_int]),
        "ro_oracle_ring2d_is_dirty": (C.c_int, [C.c_void_p, C.c_int]),
        "ro_oracle_ln_rows": (None, [f32p, i64, f32p]),
        "ro_oracle_ln_levels": (None, [f32p, i64, f32p, C.POINTER(C.c_uint8), f32p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f64(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


# ---- numpy-level conveniences -------------------------------------------------

def use_fftw(on=True):
    """Route the FP64 transform through libfftw3 (the reference's FFT) if this host has it; returns whether it is active."""
    return bool(lib().ro_oracle_use_fftw(1 if on else 0))


def fft_engine():
    """'port' (built-in radix-2), 'libfftw3' or 'mkl-fftw3-interface' -- what ro_oracle_fft_f64 runs on right now"""
    return {0: "port", 1: "libfftw3", 2: "mkl-fftw3-interface"}[lib().ro_oracle_fftw_active()]


def fft_prepare(bins):
    return lib().ro_oracle_fft_prepare(bins)


def window(bins, kind="nuttall"):
    w = np.empty(bins, dtype=np.float32)
    if kind == "nuttall":
        lib().ro_oracle_window_nuttall(bins, _f32(w))
    elif kind == "hann":
        lib().ro_oracle_window_hann(bins, _f32(w))
    else:
        raise ValueError(kind)
    return w


def row_count(samples, bins, overlap):
    return int(lib().ro_oracle_row_count(int(samples), bins, overlap))


def fft(x):
    """x: complex128 1-D, power-of-two length."""
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty_like(x)
    rc = lib().ro_oracle_fft_f64(x.size, _f64(x.view(np.float64)), _f64(out.view(np.float64)))
    if rc != 0:
        raise ValueError("oracle fft: unsupported size %d" % x.size)
    return out


def dft_direct(x):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty_like(x)
    lib().ro_oracle_dft_direct(x.size, _f64(x.view(np.float64)), _f64(out.view(np.float64)))
    return out


def stft(iq, bins, overlap, w=None, gain=0.0, first_row=0, max_rows=None):
    """iq: complex64/complex128 1-D stream, or float32/float64 interleaved [T,2].
    Returns float32 rows [R, bins] exactly as WaterfallBackend::processFFT stores them."""
    if w is None:
        w = window(bins)
    w = np.ascontiguousarray(w, dtype=np.float32)
    iq = np.ascontiguousarray(iq)
    if iq.dtype == np.complex64:
        iq = iq.view(np.float32)
    elif iq.dtype == np.complex128:
        iq = iq.view(np.float64)
    iq = iq.reshape(-1)
    samples = iq.size // 2
    total = row_count(samples, bins, overlap)
    if max_rows is None:
        max_rows = max(total - first_row, 0)
    n = max(min(total - first_row, max_rows), 0)
    rows = np.empty((n, bins), dtype=np.float32)
    if iq.dtype == np.float32:
        got = lib().ro_oracle_stft_f32(_f32(iq), samples, bins, overlap, _f32(w), gain,
                                       first_row, n, _f32(rows))
    elif iq.dtype == np.float64:
        got = lib().ro_oracle_stft(_f64(iq), samples, bins, overlap, _f32(w), gain,
                                   first_row, n, _f32(rows))
    else:
        raise TypeError(iq.dtype)
    if got != n:
        raise RuntimeError("oracle stft returned %d, expected %d" % (got, n))
    return rows


def row_with_spectrum(iq_row, w, gain=0.0):
    """One window of `bins` complex128 samples -> (float32 row, complex128 spectrum)."""
    iq_row = np.ascontiguousarray(iq_row, dtype=np.complex128)
    bins = iq_row.size
    row = np.empty(bins, dtype=np.float32)
    spec = np.empty(bins, dtype=np.complex128)
    w = np.ascontiguousarray(w, dtype=np.float32)
    rc = lib().ro_oracle_row(bins, _f64(iq_row.view(np.float64)), _f32(w), gain, _f32(row),
                             _f64(spec.view(np.float64)))
    if rc != 0:
        raise RuntimeError("oracle row failed")
    return row, spec


def scan_rows(rows, low_noise, noise_width, low_detect, detect_width, avg_bins):
    """rows: float32 [R, bins] -> (noise[R] f32, peak[R] i32, average[R] f32)."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    R = rows.shape[0]
    n = np.empty(R, np.float32)
    p = np.empty(R, np.int32)
    a = np.empty(R, np.float32)
    s = Scan()
    L = lib()
    for r in range(R):
        L.ro_oracle_scan_row(_f32(rows[r]), low_noise, noise_width, low_detect, detect_width,
                             avg_bins, C.byref(s))
        n[r], p[r], a[r] = s.noise, s.peak, s.average
    return n, p, a


def ln_levels(image):
    """image: float32 [R, W] band image -> (ln float32 [R, W], levels uint8 [R, W], (min, max)): the viewer's
    natural-log grey image (fits2png:46, :444-445, :476-502)."""
    image = np.ascontiguousarray(image, dtype=np.float32)
    ln = np.empty_like(image)
    u8 = np.empty(image.shape, np.uint8)
    mm = np.empty(2, np.float32)
    with np.errstate(all="ignore"):
        lib().ro_oracle_ln_levels(_f32(image), image.size, _f32(ln), u8.ctypes.data_as(C.POINTER(C.c_uint8)), _f32(mm))
    return ln, u8, (mm[0], mm[1])


def bolid_bands(bins, sample_rate, overlap, min_detect, max_detect, min_noise, max_noise,
                advance_time, jitter_time, avg_freq_range, noise_metadata_time=3600.0):
    b = Bands()
    rate = lib().ro_oracle_fft_sample_rate(sample_rate, bins, overlap)
    lib().ro_oracle_bolid_bands(bins, sample_rate, rate, min_detect, max_detect, min_noise,
                                max_noise, advance_time, jitter_time, avg_freq_range,
                                noise_metadata_time, C.byref(b))
    return b


class BolidFsm:
    """Drives ro_oracle_fsm_update over (n, a, peak_fq, mark) tuples."""

    def __init__(self, advance, jitter, fft_sample_rate, sample_rate, min_detect, max_detect):
        self.f = Fsm()
        lib().ro_oracle_fsm_init(C.byref(self.f), advance, jitter, fft_sample_rate, sample_rate,
                                 min_detect, max_detect)

    def update(self, n, a, peak_fq, mark):
        ev = Event()
        lib().ro_oracle_fsm_update(C.byref(self.f), float(n), float(a), float(peak_fq), int(mark),
                                   C.byref(ev))
        return ev


class Stream:
    """Frontend::process + FFTBackend::process emulation (chunked calls)."""

    def __init__(self, bins, overlap, sample_rate=48000, start=(0, 0), gain=0.0,
                 raw_capacity_rows=1):
        self.bins = bins
        self.h = lib().ro_oracle_stream_create(bins, overlap, sample_rate, start[0], start[1],
                                               gain, raw_capacity_rows)
        if not self.h:
            raise ValueError("unsupported bins %d" % bins)
        self.hop = bins - lib().ro_oracle_clamp_overlap(bins, overlap)

    def process(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex128)
        n = iq.size
        max_rows = (lib().ro_oracle_stream_pending(self.h) + n) // self.hop + 1
        rows = np.empty((max_rows, self.bins), dtype=np.float32)
        infos = (RowInfo * max_rows)()
        got = lib().ro_oracle_stream_process(self.h, _f64(iq.view(np.float64)), n, _f32(rows),
                                             infos, max_rows)
        if got < 0:
            raise RuntimeError("oracle stream_process failed: %d" % got)
        return rows[:got], [(i.offset, i.time_sec, i.time_usec, i.raw_mark) for i in infos[:got]]

    def close(self):
        if self.h:
            lib().ro_oracle_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
