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
    for flag in ("--gpus", "--steps", "--warmup", "--prewarm", "--gather", "--exchange", "--workload", "--c5-seconds",
                 "--rows", "--bins", "--overlap", "--no-streaming", "--stream-seconds", "--pattern", "--no-large", "--no-strict"):
        assert flag in r.stdout, flag
    # the all-gather north_star names is what ends a step inside the timed region unless asked otherwise
    assert "all (default)" in " ".join(r.stdout.split())


def test_bench_capi_exchange_needs_one_gpu_per_rank():
    """--exchange capi with the two-ranks-on-one-GPU rehearsal transport is refused before anything is measured"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "--exchange capi needs one GPU per rank" in src


def test_bench_builds_nothing_after_the_first_hip_call():
    """the -O0 oracle twin is made before `import torch` (a fork + exec from a process that holds the GPU is a hazard)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("libro_oracle_O0.so") < main.index("import torch")
    assert main.count("subprocess.check_call") == 1


def test_bench_refuses_to_run_without_a_device():
    import torch
    if torch.cuda.is_available():
        return                                     # (on the GPU box the -m gpu suite and the driver run it for real)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stderr + r.stdout)
    assert not r.stdout.strip().startswith("{")     # and no JSON line that could be mistaken for a measurement


def test_bench_gpus_n_from_a_plain_shell_starts_n_ranks():
    """`python3 bench.py --gpus 2` with no WORLD_SIZE in the env starts the two ranks itself (as a child
    torch.distributed.run job, before this process has imported torch) and exits with the child's code.  Without a
    device every rank refuses to run, so what can be checked here is the launcher: two ranks came up with
    WORLD_SIZE = 2, a rank said why it stops, and the launcher handed the failure on."""
    import torch
    if torch.cuda.is_available():
        return                                     # (tests/test_gpu_bench.py runs the 2-rank rehearsal for real)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "gloo-host",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode != 0
    # (torch.distributed.run ends the other rank as soon as the first one has failed: one refusal is certain, two are usual)
    assert (r.stderr + r.stdout).count("no CPU fallback") >= 1, r.stderr[-2000:]
    assert "must be started by" not in r.stderr
    assert not r.stdout.strip().startswith("{")


def test_bench_launcher_runs_before_torch_is_imported():
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(a.gpus)") < main.index("import torch")
    launcher = src[src.index("def launch_ranks("):src.index("def main():")]
    assert "import torch" not in launcher and "torch.distributed.run" in launcher
