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


def pytest_collection_modifyitems(config, items):
    # torch bundles its own libamdhip64: when GPU tests are in the run it has to be the FIRST HIP runtime the process
    # loads -- a test that loads libro_host.so / libro_stft.so (linked against /opt/rocm's copy) before the first torch
    # test would leave torch with a second runtime that sees no device
    if any(it.get_closest_marker("gpu") for it in items):
        try:
            import torch  # noqa: F401
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
