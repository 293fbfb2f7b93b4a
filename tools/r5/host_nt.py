#!/usr/bin/env python3
"""GPU box: the streaming leg at the default batch with the staging buffer written with ordinary / non-temporal stores
(RO_STAGE_NT_BYTES, read once per process: one child per setting).  Diagnostic."""
import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import ctypes, os, sys
sys.path.insert(0, %r)
import torch
H = ctypes.CDLL(os.path.join(%r, "tests", "harness", "libro_host_harness.so"))
H.ro_host_stream_bench.restype = ctypes.c_int
H.ro_host_stream_bench.argtypes = [ctypes.c_int] * 4 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
for batch in (0, 0, 24, 256):
    stats = (ctypes.c_double * 16)()
    rc = H.ro_host_stream_bench(32768, 24576, 48000, 4096, 2.0, batch, 80, stats)
    print("  batch %%d: rc %%d, %%.4g rows/s, push %%.2f us, fetch %%.2f us" %% (batch, rc, stats[2] / stats[0], 1e3 * stats[8], 1e3 * stats[9]), flush=True)
''' % (ROOT, ROOT)
for nt in ("8388608", "0", "8388608", "0"):
    env = dict(os.environ, RO_STAGE_NT_BYTES=nt)
    print("RO_STAGE_NT_BYTES=%s" % nt, flush=True)
    subprocess.run([sys.executable, "-c", code], env=env, check=False)
