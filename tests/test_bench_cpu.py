"""CPU: the driver-facing contract of bench.py that can be checked without a GPU -- its flags exist, and with no
HIP device it refuses to run (there is no CPU fallback to time by accident)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_flags():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--gather", "--rows", "--bins", "--overlap"):
        assert flag in r.stdout, flag


def test_bench_refuses_to_run_without_a_device():
    import torch
    if torch.cuda.is_available():
        return                                     # (on the GPU box the -m gpu suite and the driver run it for real)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stderr + r.stdout)
    assert not r.stdout.strip().startswith("{")     # and no JSON line that could be mistaken for a measurement
