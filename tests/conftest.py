import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# torch bundles its own libamdhip64 (SONAME libamdhip64.so.7, the name libro_stft.so / libro_host.so ask the loader for):
# imported FIRST, its copy is the one every later library resolves against and the process holds ONE HIP runtime.  The
# other order maps a second copy (torch asks for its own by file name) -- capi.require_one_hip_runtime refuses handles
# then; tests/test_capi_cpu.py::test_two_hip_runtimes_are_refused covers both orders in child processes.
try:
    import torch  # noqa: F401,E402
except ImportError:
    pass


@pytest.fixture(scope="session")
def ro():
    """The product package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module("radio-observer_amd")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle -- the checker, never the thing under test."""
    import ro_oracle
    ro_oracle.lib()
    return ro_oracle


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return torch
