"""ctypes binding of include/ro_stft.h (the C ABI of libro_stft.so).

Only plumbing lives here: argument marshalling, error-code -> exception.  The
arithmetic is in the HIP kernels.  Device buffers are passed as raw integer
addresses (e.g. torch.Tensor.data_ptr()).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

RO_WINDOW_NUTTALL, RO_WINDOW_HANN, RO_WINDOW_CUSTOM = 0, 1, 2
RO_IQ_F32, RO_IQ_I16, RO_IQ_F64 = 0, 1, 2
RO_PRECISION_F32, RO_PRECISION_F64 = 0, 1

RO_OK = 0
_ERR_NAMES = {-1: "RO_ERR_INVALID", -2: "RO_ERR_UNSUPPORTED", -3: "RO_ERR_HIP", -4: "RO_ERR_NOMEM",
              -5: "RO_ERR_STATE"}


class StftError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s (%d): %s" % (_ERR_NAMES.get(code, "RO_ERR"), code, text))
        self.code = code


class Bands(C.Structure):
    """ro_bands_t -- BolidRecorder::start's bin ranges (src/BolidRecorder.cpp:84-102)."""
    _fields_ = [("low_noise", C.c_int32), ("noise_width", C.c_int32), ("low_detect", C.c_int32),
                ("detect_width", C.c_int32), ("avg_bins", C.c_int32)]


class ScanRecord(C.Structure):
    """ro_scan_record_t -- (n, p, a) of BolidRecorder::update (src/BolidRecorder.cpp:121-132)."""
    _fields_ = [("noise", C.c_float), ("peak", C.c_int32), ("average", C.c_float)]


class Timing(C.Structure):
    """ro_stft_timing_t"""
    _fields_ = [("push_calls", C.c_int64), ("push_ms_avg", C.c_double), ("push_ms_max", C.c_double),
                ("batches", C.c_int64), ("batch_gpu_ms_avg", C.c_double), ("batch_gpu_ms_max", C.c_double),
                ("batch_rows", C.c_int64), ("row_gpu_us_avg", C.c_double),
                ("fetch_calls", C.c_int64), ("fetch_ms_avg", C.c_double), ("fetch_ms_max", C.c_double)]


SCAN_DTYPE = np.dtype([("noise", np.float32), ("peak", np.int32), ("average", np.float32)])


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("bins", C.c_int32), ("overlap", C.c_int32),
                ("sample_rate", C.c_int32), ("window_kind", C.c_int32),
                ("window_table", C.POINTER(C.c_float)), ("iq_gain", C.c_double),
                ("iq_phase_shift", C.c_int32), ("device", C.c_int32), ("max_batch_rows", C.c_int32),
                ("enable_scan", C.c_int32), ("bands", Bands), ("tile_first_col", C.c_int32),
                ("tile_cols", C.c_int32), ("spare_cus_per_xcd", C.c_int32), ("precision", C.c_int32),
                ("tile_ln", C.c_int32)]


_EXPORTS = {
    # name: (restype, argtypes)
    "ro_abi_version": (C.c_int, []),
    "ro_last_error": (C.c_char_p, []),
    "ro_device_count": (C.c_int, []),
    "ro_clamp_overlap": (C.c_int, [C.c_int, C.c_int]),
    "ro_fft_sample_rate": (C.c_float, [C.c_int, C.c_int, C.c_int]),
    "ro_frequency_to_bin": (C.c_int, [C.c_int, C.c_int, C.c_float]),
    "ro_bin_to_frequency": (C.c_float, [C.c_int, C.c_int, C.c_int]),
    "ro_time_to_fft_samples": (C.c_int, [C.c_double, C.c_float]),
    "ro_row_count": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "ro_window_table": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "ro_bins_supported": (C.c_int, [C.c_int]),
    "ro_shard_rows": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ro_shard_samples": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64)]),
    "ro_shard_max_rows": (C.c_int64, [C.c_int64, C.c_int]),
    "ro_stitch_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_size_t, C.c_void_p]),
    "ro_direct_schedule": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ro_allgather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_size_t,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "ro_gather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                 C.c_void_p, C.c_void_p]),
    "ro_allgather_rows_direct": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_size_t,
                                           C.c_void_p, C.c_void_p]),
    "ro_stitch_rows_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ro_stft_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "ro_stft_destroy": (C.c_int, [C.c_void_p]),
    "ro_stft_get_window": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "ro_stft_hop": (C.c_int, [C.c_void_p]),
    "ro_stft_bins": (C.c_int, [C.c_void_p]),
    "ro_stft_device_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "ro_stft_set_bands": (C.c_int, [C.c_void_p, C.POINTER(Bands)]),
    "ro_stft_run_resident": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ro_stft_run_resident_ln": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                          C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "ro_ln_levels": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p]),
    "ro_stft_fetch_ln": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ro_stft_spectra_resident": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                           C.c_void_p, C.c_int64, C.c_void_p]),
    "ro_stft_scan_resident": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                        C.c_void_p]),
    "ro_stft_ln_tile_resident": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ro_stft_time_resident": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                        C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "ro_stft_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_int64)]),
    "ro_stft_flush": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ro_stft_fetch": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_float),
                                C.POINTER(ScanRecord), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ro_stft_reset": (C.c_int, [C.c_void_p]),
    "ro_stft_rows_complete": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ro_stft_set_row_sink": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "ro_pinned_alloc": (C.c_void_p, [C.c_int, C.c_size_t]),
    "ro_pinned_free": (None, [C.c_void_p]),
    "ro_pinned_check": (C.c_int, [C.c_void_p, C.c_size_t]),
    "ro_stft_timing": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "ro_stft_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
}

_lib = None


def library_path():
    return _build.LIB


def exported_symbols():
    return sorted(_EXPORTS)


def hip_runtimes():
    """Paths of every libamdhip64 mapped into this process.  libro_stft.so asks the loader for `libamdhip64.so.7`; a
    copy some other package has already loaded under that SONAME (torch bundles one) is reused, so a process that
    imports torch FIRST holds one runtime.  The other order leaves two -- torch asks for its own copy by file name --
    and then a device pointer, a stream or page-locked memory of one runtime means nothing to the other."""
    found = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1]
                if "libamdhip64" in os.path.basename(path):
                    found.add(os.path.realpath(path))
    except OSError:
        pass
    return sorted(found)


def require_one_hip_runtime():
    """Raise when two HIP runtimes share the process (see hip_runtimes): handles are refused rather than handed
    pointers the other runtime owns."""
    rts = hip_runtimes()
    if len(rts) > 1:
        raise StftError(-5, "two HIP runtimes are mapped into this process (%s): import torch BEFORE radio-observer_amd "
                            "loads libro_stft.so, so that both use the same one" % ", ".join(rts))


def library():
    """Load libro_stft.so (building it first if the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("RO_STFT_LIB")           # diagnostic builds (tools/ablate.sh) only
    if not path:
        path = _build.LIB
        if not os.path.exists(path):
            _build.build()
    lib = C.CDLL(path)
    for name, (res, args) in _EXPORTS.items():
        fn = getattr(lib, name)       # AttributeError here = symbol missing from the .so
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class PinnedArray:
    """a float32 numpy view [rows, cols] of page-locked host memory from ro_pinned_alloc (freed with the object)"""

    def __init__(self, rows, cols, device=0):
        n = rows * cols * 4
        self._p = library().ro_pinned_alloc(device, n)
        if not self._p:
            raise MemoryError("ro_pinned_alloc(%d bytes) failed" % n)
        self.array = np.ctypeslib.as_array((C.c_float * (rows * cols)).from_address(self._p)).reshape(rows, cols)

    def close(self):
        if self._p:
            self.array = None
            library().ro_pinned_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _check(rc):
    if rc != RO_OK:
        raise StftError(rc, (library().ro_last_error() or b"").decode("utf-8", "replace"))


# ---- host helpers -------------------------------------------------------------
def clamp_overlap(bins, overlap):
    return library().ro_clamp_overlap(bins, overlap)


def fft_sample_rate(sample_rate, bins, overlap):
    return library().ro_fft_sample_rate(sample_rate, bins, overlap)


def frequency_to_bin(bins, sample_rate, frequency):
    return library().ro_frequency_to_bin(bins, sample_rate, frequency)


def bin_to_frequency(bins, sample_rate, bin_):
    return library().ro_bin_to_frequency(bins, sample_rate, bin_)


def time_to_fft_samples(seconds, fft_rate):
    return library().ro_time_to_fft_samples(seconds, fft_rate)


def row_count(samples, bins, overlap):
    return int(library().ro_row_count(int(samples), bins, overlap))


def window_table(kind, bins):
    w = np.empty(bins, dtype=np.float32)
    _check(library().ro_window_table(kind, bins, w.ctypes.data_as(C.POINTER(C.c_float))))
    return w


def shard_rows(total_rows, world, rank):
    """(first_row, rows) of `rank` -- ro_shard_rows"""
    first, rows = C.c_int64(), C.c_int64()
    _check(library().ro_shard_rows(int(total_rows), world, rank, C.byref(first), C.byref(rows)))
    return first.value, rows.value


def shard_samples(first_row, rows, bins, overlap):
    """(first_sample, samples) a shard must hold, halo included -- ro_shard_samples"""
    s0, ns = C.c_int64(), C.c_int64()
    _check(library().ro_shard_samples(int(first_row), int(rows), bins, overlap, C.byref(s0), C.byref(ns)))
    return s0.value, ns.value


def shard_max_rows(total_rows, world):
    n = library().ro_shard_max_rows(int(total_rows), world)
    if n < 0:
        _check(int(n))
    return int(n)


def direct_schedule(world, rank, total_rows, k):
    """(to, from, recv_first_row, recv_rows) of step k of the direct exchange -- ro_direct_schedule"""
    to, frm = C.c_int(), C.c_int()
    f, n = C.c_int64(), C.c_int64()
    _check(library().ro_direct_schedule(world, rank, int(total_rows), k, C.byref(to), C.byref(frm), C.byref(f), C.byref(n)))
    return to.value, frm.value, f.value, n.value


def stitch_rows(gathered, total_rows, world):
    """numpy [world * max_rows, ...] (equal-block all-gather result) -> [total_rows, ...] in row order"""
    g = np.ascontiguousarray(gathered)
    row_bytes = g.strides[0]
    out = np.empty((int(total_rows),) + g.shape[1:], dtype=g.dtype)
    _check(library().ro_stitch_rows(C.c_void_p(g.ctypes.data), int(total_rows), world, row_bytes,
                                    C.c_void_p(out.ctypes.data)))
    return out


def ln_levels(ln, mn, mx):
    """the viewer's grey levels of log values given the image's range -- ro_ln_levels (host)"""
    a = np.ascontiguousarray(ln, dtype=np.float32)
    out = np.empty(a.shape, np.uint8)
    _check(library().ro_ln_levels(C.c_void_p(a.ctypes.data), a.size, float(mn), float(mx), C.c_void_p(out.ctypes.data)))
    return out


def bins_supported(bins):
    return bool(library().ro_bins_supported(bins))


def device_count():
    n = library().ro_device_count()
    if n < 0:
        _check(n)
    return n


def _ptr(x):
    """int address, torch tensor (data_ptr) or None -> c_void_p."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


# ---- the handle ---------------------------------------------------------------
class Stft:
    """One STFT stream on one GPU (ro_stft_t)."""

    def __init__(self, bins=32768, overlap=0, sample_rate=48000, window=RO_WINDOW_NUTTALL,
                 window_table=None, iq_gain=0.0, iq_phase_shift=0, device=0, max_batch_rows=0,
                 bands=None, tile=None, spare_cus_per_xcd=0, precision=RO_PRECISION_F32, tile_ln=False):
        cfg = Config()
        cfg.struct_size = C.sizeof(Config)
        cfg.bins, cfg.overlap, cfg.sample_rate = bins, overlap, sample_rate
        cfg.window_kind = window
        self._wt = None
        if window_table is not None:
            self._wt = np.ascontiguousarray(window_table, dtype=np.float32)
            if self._wt.size != bins:
                raise ValueError("window_table must have `bins` entries")
            cfg.window_kind = RO_WINDOW_CUSTOM
            cfg.window_table = self._wt.ctypes.data_as(C.POINTER(C.c_float))
        cfg.iq_gain = iq_gain
        cfg.iq_phase_shift = iq_phase_shift
        cfg.device = device
        cfg.max_batch_rows = max_batch_rows
        if bands is not None:
            cfg.enable_scan = 1
            cfg.bands = bands
        if tile is not None:
            cfg.tile_first_col, cfg.tile_cols = tile
        cfg.spare_cus_per_xcd = spare_cus_per_xcd
        cfg.precision = precision
        cfg.tile_ln = 1 if tile_ln else 0
        self._h = C.c_void_p()
        lib = library()
        require_one_hip_runtime()
        _check(lib.ro_stft_create(C.byref(cfg), C.byref(self._h)))
        self.bins = bins
        self.hop = library().ro_stft_hop(self._h)
        self.scan_enabled = bands is not None
        self.tile = tile

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            library().ro_stft_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def window(self):
        w = np.empty(self.bins, dtype=np.float32)
        _check(library().ro_stft_get_window(self._h, w.ctypes.data_as(C.POINTER(C.c_float))))
        return w

    @property
    def device_name(self):
        buf = C.create_string_buffer(256)
        _check(library().ro_stft_device_name(self._h, buf, 256))
        return buf.value.decode()

    def set_bands(self, bands):
        _check(library().ro_stft_set_bands(self._h, C.byref(bands)))
        self.scan_enabled = True

    # -- resident path
    def run_resident(self, d_iq, fmt, samples, first_row, rows, d_rows, row_stride=None, d_tile=None,
                     d_records=None, stream=None):
        _check(library().ro_stft_run_resident(self._h, _ptr(d_iq), fmt, samples, first_row, rows,
                                              _ptr(d_rows), row_stride or self.bins, _ptr(d_tile),
                                              _ptr(d_records), _ptr(stream)))

    def run_resident_ln(self, d_iq, fmt, samples, first_row, rows, d_rows, d_tile, d_ln=None, d_minmax=None,
                        row_stride=None, d_records=None, stream=None):
        """run_resident + the log of the tile and the rows' min / max of it (handle created with tile_ln=True)"""
        _check(library().ro_stft_run_resident_ln(self._h, _ptr(d_iq), fmt, samples, first_row, rows, _ptr(d_rows),
                                                 row_stride or self.bins, _ptr(d_tile), _ptr(d_ln), _ptr(d_minmax),
                                                 _ptr(d_records), _ptr(stream)))

    def fetch_ln(self, max_rows):
        """(first_row, tile, ln, minmax[rows, 2], records or None) of up to max_rows rows (tile_ln handles)"""
        cols = self.tile[1]
        tile = np.empty((max_rows, cols), np.float32)
        ln = np.empty((max_rows, cols), np.float32)
        mm = np.empty((max_rows, 2), np.float32)
        recs = np.empty(max_rows, dtype=SCAN_DTYPE) if self.scan_enabled else None
        first, got = C.c_int64(), C.c_int64()
        _check(library().ro_stft_fetch_ln(self._h, max_rows, C.c_void_p(tile.ctypes.data), C.c_void_p(ln.ctypes.data),
                                          C.c_void_p(mm.ctypes.data),
                                          C.c_void_p(recs.ctypes.data) if recs is not None else None,
                                          C.byref(first), C.byref(got)))
        g = got.value
        return first.value, tile[:g], ln[:g], mm[:g], (recs[:g] if recs is not None else None)

    def spectra_resident(self, d_iq, fmt, samples, first_row, rows, d_spectra, stride=None, stream=None):
        """complex spectra (rows x stride x {re, im} float32, bin k at element k) instead of magnitudes"""
        _check(library().ro_stft_spectra_resident(self._h, _ptr(d_iq), fmt, samples, first_row, rows,
                                                  _ptr(d_spectra), stride or self.bins, _ptr(stream)))

    def scan_resident(self, d_rows, rows, d_records, row_stride=None, stream=None):
        _check(library().ro_stft_scan_resident(self._h, _ptr(d_rows), row_stride or self.bins, rows,
                                               _ptr(d_records), _ptr(stream)))

    def ln_tile_resident(self, d_rows, rows, first_col, cols, d_ln=None, d_u8=None, d_minmax=None,
                         row_stride=None, stream=None):
        _check(library().ro_stft_ln_tile_resident(self._h, _ptr(d_rows), row_stride or self.bins, rows, first_col,
                                                  cols, _ptr(d_ln), _ptr(d_u8), _ptr(d_minmax), _ptr(stream)))

    def time_resident(self, d_iq, fmt, samples, first_row, rows, d_rows, iters, row_stride=None,
                      d_tile=None, d_records=None, stream=None):
        ms = (C.c_float * iters)()
        kern = (C.c_float * 2)()
        _check(library().ro_stft_time_resident(self._h, _ptr(d_iq), fmt, samples, first_row, rows,
                                               _ptr(d_rows), row_stride or self.bins, _ptr(d_tile),
                                               _ptr(d_records), _ptr(stream), iters, ms, kern))
        return np.array(ms[:], dtype=np.float64), float(kern[0]), float(kern[1])

    # -- streaming path
    def push(self, iq):
        """iq: numpy array, complex64 / complex128 / float32 [n,2] / float64 [n,2] / int16 [n,2]."""
        a = np.ascontiguousarray(iq)
        if a.dtype == np.complex64:
            a, fmt = a.view(np.float32), RO_IQ_F32
        elif a.dtype == np.complex128:
            a, fmt = a.view(np.float64), RO_IQ_F64
        elif a.dtype == np.float32:
            fmt = RO_IQ_F32
        elif a.dtype == np.float64:
            fmt = RO_IQ_F64
        elif a.dtype == np.int16:
            fmt = RO_IQ_I16
        else:
            raise TypeError("unsupported sample dtype %s" % a.dtype)
        n = a.size // 2
        ready = C.c_int64()
        _check(library().ro_stft_push(self._h, C.c_void_p(a.ctypes.data), fmt, n, C.byref(ready)))
        return ready.value

    def flush(self):
        ready = C.c_int64()
        _check(library().ro_stft_flush(self._h, C.byref(ready)))
        return ready.value

    def fetch(self, max_rows, first_col=None, cols=None, want_records=None):
        if first_col is None:                        # default: everything this handle brings to the host
            first_col = self.tile[0] if self.tile else 0
        if cols is None:
            cols = (self.tile[0] + self.tile[1] - first_col) if self.tile else self.bins - first_col
        want_records = self.scan_enabled if want_records is None else want_records
        rows = np.empty((max_rows, cols), dtype=np.float32)
        recs = np.empty(max_rows, dtype=SCAN_DTYPE) if want_records else None
        first = C.c_int64()
        got = C.c_int64()
        _check(library().ro_stft_fetch(
            self._h, max_rows, first_col, cols, rows.ctypes.data_as(C.POINTER(C.c_float)),
            recs.ctypes.data_as(C.POINTER(ScanRecord)) if recs is not None else None,
            C.byref(first), C.byref(got)))
        g = got.value
        return first.value, rows[:g], (recs[:g] if recs is not None else None)

    def reset(self):
        _check(library().ro_stft_reset(self._h))

    def set_row_sink(self, ring, first_slot=0):
        """ring: a C-contiguous float32 numpy array [capacity_rows, row_stride] that stays alive (and unmoved) while it
        is the sink (pinned_array gives page-locked ones); None removes the sink"""
        if ring is None:
            _check(library().ro_stft_set_row_sink(self._h, None, 0, 0, 0))
            self._sink = None
            return
        assert ring.dtype == np.float32 and ring.ndim == 2 and ring.flags["C_CONTIGUOUS"]
        _check(library().ro_stft_set_row_sink(self._h, C.c_void_p(ring.ctypes.data), ring.shape[1], ring.shape[0], first_slot))
        self._sink = ring

    def fetch_records(self, max_rows):
        """ro_stft_fetch with rows_out = NULL (the rows are in the sink): (first row index, rows got, records or None)"""
        recs = np.empty(max_rows, dtype=SCAN_DTYPE) if self.scan_enabled else None
        first, got = C.c_int64(), C.c_int64()
        _check(library().ro_stft_fetch(self._h, max_rows, 0, 0, None,
                                       recs.ctypes.data_as(C.POINTER(ScanRecord)) if recs is not None else None,
                                       C.byref(first), C.byref(got)))
        return first.value, got.value, (recs[:got.value] if recs is not None else None)

    def timing(self, reset=False):
        t = Timing()
        _check(library().ro_stft_timing(self._h, C.byref(t), 1 if reset else 0))
        return {k: getattr(t, k) for k, _ in Timing._fields_}

    def rows_complete(self):
        """rows ro_stft_fetch would hand over without waiting (their batches have finished on the device)"""
        n = C.c_int64()
        _check(library().ro_stft_rows_complete(self._h, C.byref(n)))
        return n.value

    def stats(self):
        s, r, l = C.c_int64(), C.c_int64(), C.c_int64()
        ms = C.c_double()
        _check(library().ro_stft_stats(self._h, C.byref(s), C.byref(r), C.byref(l), C.byref(ms)))
        return {"samples_in": s.value, "rows_out": r.value, "launches": l.value,
                "kernel_ms_total": ms.value}
