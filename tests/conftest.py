import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Without a HIP device the gpu-marked tests are skipped (a plain `pytest tests` run must not error out of its fixtures)."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="needs a HIP device (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def cpu_backend():
    """Installs the torch-CPU emulation of the kernel layer (tests only) and a CPU variable store."""
    from gansynth_amd import kernels, variables
    from tests.cpu_kernels import CpuEmuKernels
    old_k, old_s = kernels._K, variables._default
    kernels.set_backend(CpuEmuKernels())
    variables.set_default_store(variables.VariableStore(device="cpu"))
    yield
    kernels._K, variables._default = old_k, old_s


@pytest.fixture
def gpu_store():
    from gansynth_amd import variables
    old = variables._default
    variables.set_default_store(variables.VariableStore(device="cuda"))
    yield variables.default_store()
    variables._default = old
