"""ctypes access to the product's host-side C++ mirror (radio-observer_amd/host/libro_host.so) through the test-only
harness library tests/harness/libro_host_harness.so, which links it.  Returns None when it has not been built yet."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("RO_HOST_LIB") or os.path.join(ROOT, "tests", "harness", "libro_host_harness.so")
_lib = False


def host_library():
    global _lib
    if _lib is False:
        _lib = C.CDLL(PATH) if os.path.exists(PATH) else None
        if _lib is not None:
            for n in ("capacity", "chunk_rows", "get_size", "is_full", "push", "mark"):
                getattr(_lib, "ro_host_ring_" + n).argtypes = [C.c_void_p]
                getattr(_lib, "ro_host_ring_" + n).restype = C.c_int
            for n in ("normalize", "size_from", "free_reservation", "is_dirty"):
                getattr(_lib, "ro_host_ring_" + n).argtypes = [C.c_void_p, C.c_int]
                getattr(_lib, "ro_host_ring_" + n).restype = C.c_int
            for n in ("size_between", "reserve"):
                getattr(_lib, "ro_host_ring_" + n).argtypes = [C.c_void_p, C.c_int, C.c_int]
                getattr(_lib, "ro_host_ring_" + n).restype = C.c_int
            _lib.ro_host_ring_push_run.argtypes = [C.c_void_p, C.c_int, C.c_float]
            _lib.ro_host_ring_push_run.restype = C.c_int
            _lib.ro_host_ring_push_written.argtypes = [C.c_void_p, C.c_int]
            _lib.ro_host_ring_push_written.restype = None
            _lib.ro_host_ring_mark_ahead.argtypes = [C.c_void_p, C.c_int]
            _lib.ro_host_ring_mark_ahead.restype = None
            _lib.ro_host_ring_at0.argtypes = [C.c_void_p, C.c_int]
            _lib.ro_host_ring_at0.restype = C.c_float
            _lib.ro_host_ring_set0.argtypes = [C.c_void_p, C.c_int, C.c_float]
            _lib.ro_host_ring_set0.restype = None
            _lib.ro_host_ring_create.argtypes = [C.c_int, C.c_int, C.c_int]
            _lib.ro_host_ring_create.restype = C.c_void_p
            _lib.ro_host_ring_destroy.argtypes = [C.c_void_p]
    return _lib


class HostRing:
    """RingBuffer2D<float> of the host mirror (radio-observer_amd/host/RingBuffer.h)."""

    def __init__(self, width, chunk, capacity=None):
        self.L = host_library()
        self.h = self.L.ro_host_ring_create(width, chunk, -1 if capacity is None else capacity)

    def __getattr__(self, name):
        fn = getattr(self.L, "ro_host_ring_" + name)
        return lambda *a: fn(self.h, *a)

    def __del__(self):
        try:
            self.L.ro_host_ring_destroy(self.h)
        except Exception:
            pass


class BolidEvent(C.Structure):
    _fields_ = [("row", C.c_int64), ("start", C.c_int), ("length", C.c_int), ("duration", C.c_float),
                ("noise", C.c_float), ("peakFreq", C.c_float), ("magnitude", C.c_float), ("fmin", C.c_float),
                ("fmax", C.c_float), ("rawLength", C.c_int)]


class HostPipeline:
    """Frontend -> HipWaterfallBackend -> BolidRecorder of the host mirror."""

    def __init__(self, bins, overlap, sample_rate=48000, start=(0, 0), max_batch_rows=0, snapshot_length=60,
                 detect=(10300.0, 10900.0), noise=(9000.0, 9600.0), advance_time=2.0, jitter_time=5.0,
                 avg_range=40.0, out_dir=None, origin="teststn", snap_band=(9000.0, 12000.0)):
        L = host_library()
        self.L = L
        L.ro_host_pipeline_create.restype = C.c_void_p
        L.ro_host_pipeline_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                              C.c_float, C.c_float, C.c_float, C.c_float, C.c_double, C.c_double,
                                              C.c_float]
        L.ro_host_pipeline_process.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        L.ro_host_pipeline_end.argtypes = [C.c_void_p]
        L.ro_host_pipeline_destroy.argtypes = [C.c_void_p]
        L.ro_host_pipeline_rows.argtypes = [C.c_void_p]
        L.ro_host_pipeline_rows.restype = C.c_int64
        L.ro_host_pipeline_error.argtypes = [C.c_void_p]
        L.ro_host_pipeline_error.restype = C.c_char_p
        for n in ("ring_capacity", "ring_mark", "raw_capacity", "state", "batch_rows"):
            getattr(L, "ro_host_pipeline_" + n).argtypes = [C.c_void_p]
            getattr(L, "ro_host_pipeline_" + n).restype = C.c_int
        L.ro_host_pipeline_ring_row.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        L.ro_host_pipeline_row_info.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_int64),
                                                C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        L.ro_host_pipeline_raw_handle.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                                  C.POINTER(C.c_int64)]
        L.ro_host_pipeline_bands.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ro_host_pipeline_events.argtypes = [C.c_void_p, C.POINTER(BolidEvent), C.c_int]
        L.ro_host_pipeline_create_snap.restype = C.c_void_p
        L.ro_host_pipeline_create_snap.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                                   C.c_float, C.c_float, C.c_char_p, C.c_char_p]
        L.ro_host_pipeline_bolid_files.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
        L.ro_host_pipeline_files.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        self.bins = bins
        if out_dir is None:                      # detector only, no files
            self.h = L.ro_host_pipeline_create(bins, overlap, sample_rate, start[0], start[1], max_batch_rows,
                                               snapshot_length, detect[0], detect[1], noise[0], noise[1],
                                               advance_time, jitter_time, avg_range)
        else:                                    # SnapshotRecorder + BolidRecorder (its default bands) writing FITS
            self.h = L.ro_host_pipeline_create_snap(bins, overlap, sample_rate, start[0], start[1], max_batch_rows,
                                                    snapshot_length, snap_band[0], snap_band[1],
                                                    str(out_dir).encode(), origin.encode())

    def process(self, iq):
        import numpy as np
        a = np.ascontiguousarray(iq, dtype=np.complex128).view(np.float64)
        self.L.ro_host_pipeline_process(self.h, a.ctypes.data_as(C.POINTER(C.c_double)), a.size // 2)

    def end(self):
        self.L.ro_host_pipeline_end(self.h)

    @property
    def rows(self):
        return self.L.ro_host_pipeline_rows(self.h)

    @property
    def error(self):
        return (self.L.ro_host_pipeline_error(self.h) or b"").decode()

    def ring_row(self, mark):
        import numpy as np
        out = np.empty(self.bins, np.float32)
        self.L.ro_host_pipeline_ring_row(self.h, mark, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def row_info(self, i):
        o, s, u, m = C.c_uint64(), C.c_int64(), C.c_int64(), C.c_int()
        assert self.L.ro_host_pipeline_row_info(self.h, i, C.byref(o), C.byref(s), C.byref(u), C.byref(m)) == 0
        return (o.value, s.value, u.value, m.value)

    def raw_handle(self, mark):
        m, s, u = C.c_int(), C.c_int64(), C.c_int64()
        self.L.ro_host_pipeline_raw_handle(self.h, mark, C.byref(m), C.byref(s), C.byref(u))
        return (m.value, s.value, u.value)

    def bands(self):
        out = (C.c_int * 7)()
        self.L.ro_host_pipeline_bands(self.h, out)
        return list(out)

    def events(self):
        buf = (BolidEvent * 64)()
        n = self.L.ro_host_pipeline_events(self.h, buf, 64)
        return [buf[i] for i in range(min(n, 64))]

    def snapshot_files(self):
        buf = C.create_string_buffer(1 << 16)
        self.L.ro_host_pipeline_files(self.h, buf, 1 << 16)
        return buf.value.decode().split()

    def bolid_files(self, raw=False):
        buf = C.create_string_buffer(1 << 16)
        self.L.ro_host_pipeline_bolid_files(self.h, 1 if raw else 0, buf, 1 << 16)
        return buf.value.decode().split()

    def ring_capacity(self):
        return self.L.ro_host_pipeline_ring_capacity(self.h)

    def ring_mark(self):
        return self.L.ro_host_pipeline_ring_mark(self.h)

    def raw_capacity(self):
        return self.L.ro_host_pipeline_raw_capacity(self.h)

    def state(self):
        return self.L.ro_host_pipeline_state(self.h)

    def batch_rows(self):
        return self.L.ro_host_pipeline_batch_rows(self.h)

    def close(self):
        if self.h:
            self.L.ro_host_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
