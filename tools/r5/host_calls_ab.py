#!/usr/bin/env python3
"""GPU box: the streaming leg at the default batch, settings alternated INSIDE one process (process-to-process spread is
larger than what is measured): timing events around every graphed batch / one in eight; an `uploaded` event of its own
behind the graph / the batch's `done` event; the graph / its calls one by one.  Needs a -DRO_DIAG build of libro_stft.so preloaded:
  LD_PRELOAD=build/ab/libro_stft_diag.so python3 tools/r5/host_calls_ab.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = ctypes.CDLL(os.path.join(ROOT, "tests", "harness", "libro_host_harness.so"))
H.ro_host_stream_bench.restype = ctypes.c_int
H.ro_host_stream_bench.argtypes = [ctypes.c_int] * 4 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
settings = [("every batch timed, uploaded + done", "1", "1", "0"), ("one in 8 timed, uploaded + done", "8", "1", "0"),
            ("one in 8 timed, done only", "8", "0", "0"), ("one in 8 timed, calls instead of the graph", "8", "1", "1")]
if len(sys.argv) > 1:
    settings = [settings[int(i)] for i in sys.argv[1].split(",")]
res = {name: [] for name, _, _, _ in settings}
for rnd in range(5):
    for name, every, two, direct in settings:
        os.environ["RO_GRAPH_TIME_EVERY"] = every
        os.environ["RO_GRAPH_TWO_EVENTS"] = two
        os.environ["RO_GRAPH_DIRECT"] = direct
        stats = (ctypes.c_double * 16)()
        rc = H.ro_host_stream_bench(32768, 24576, 48000, 4096, 1.5, 0, 80, stats)
        assert rc == 0, rc
        res[name].append(stats[2] / stats[0])
for name, _, _, _ in settings:
    v = sorted(res[name])
    print("%-36s median %.4g rows/s   (%s)" % (name, v[len(v) // 2], " ".join("%.4g" % x for x in res[name])), flush=True)
