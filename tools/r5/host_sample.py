#!/usr/bin/env python3
"""GPU box: where the host thread of the drop-in path spends its CPU time (tests/harness' PC sampler, RO_HOST_SAMPLE):
bench.py's streaming leg at the Backend's default batch (0) and at 256 rows per launch.  Diagnostic."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first: one HIP runtime in the process)
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r5host"
os.makedirs(out, exist_ok=True)
H = ctypes.CDLL(os.path.join(ROOT, "tests", "harness", "libro_host_harness.so"))
H.ro_host_stream_bench.restype = ctypes.c_int
H.ro_host_stream_bench.argtypes = [ctypes.c_int] * 4 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
for nblk, batch, sample in ((16, 0, False), (16, 256, False), (2, 0, False), (2, 256, False), (2, 0, True), (2, 256, True)):
    if True:
        os.environ["RO_STREAM_NBLK"] = str(nblk)
        if sample:
            os.environ["RO_HOST_SAMPLE"] = os.path.join(out, "pcs_batch%d.txt" % batch)
        else:
            os.environ.pop("RO_HOST_SAMPLE", None)
        stats = (ctypes.c_double * 16)()
        rc = H.ro_host_stream_bench(32768, 24576, 48000, 4096, 15.0 if sample else 2.0, batch, 2 * 8 + 64, stats)
        print("blocks %d batch %d sampler %s: rc %d, %.4g rows/s, push %.2f us, fetch %.2f us (%d calls), %d batches of %d rows"
              % (nblk, batch, "on" if sample else "off", rc, stats[2] / stats[0], 1e3 * stats[8], 1e3 * stats[9], stats[13],
                 stats[14], stats[4]), flush=True)
