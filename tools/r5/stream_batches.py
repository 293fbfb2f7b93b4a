#!/usr/bin/env python3
"""GPU box: the streaming leg over rows per launch (one process, two rounds).  Diagnostic."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
H = ctypes.CDLL(os.path.join(ROOT, "tests", "harness", "libro_host_harness.so"))
H.ro_host_stream_bench.restype = ctypes.c_int
H.ro_host_stream_bench.argtypes = [ctypes.c_int] * 4 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
for rnd in range(2):
    for batch in (0, 12, 24, 48, 96, 256, 512, 1024, 2048):
        stats = (ctypes.c_double * 16)()
        rc = H.ro_host_stream_bench(32768, 24576, 48000, 4096, 1.5, batch, 80, stats)
        print("round %d batch %4d: rc %d, %.4g rows/s, push %.2f us, fetch %.2f us (%d fetches), batch on the device %.3f ms"
              % (rnd, batch, rc, stats[2] / stats[0], 1e3 * stats[8], 1e3 * stats[9], stats[13], stats[10]), flush=True)
