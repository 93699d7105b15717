"""CPU: the C-ABI library loads here (hipcc cross-compiled it; no GPU needed to dlopen), exports every symbol
include/gansynth_hip.h declares, and the host refuses to run without a device (no silent fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "gansynth_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from gansynth_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gansynth_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.gs_version() >= 100
    assert lib.gs_last_error() is not None


def test_argument_validation_without_gpu():
    """Pure host-side checks of the ABI (no kernel is launched)."""
    from gansynth_amd import _lib
    lib = _lib.load()
    assert lib.gs_conv2d_workspace_bytes(_lib.CONV_FWD, 8, 128, 1024, 32, 32, 3, 1, _lib.GS_F32) == 9 * 32 * 32 * 4
    assert lib.gs_conv2d_workspace_bytes(_lib.CONV_BWD_WEIGHT, 8, 128, 1024, 32, 32, 3, 1, _lib.GS_F32) == 512 * (9 * 32 * 32 + 32) * 4  # 512 pixel slices of fp32 partials (9 taps + a bias row)
    assert lib.gs_conv2d_fwd(None, None, None, 1, 8, 8, 32, 32, 5, 1, 1.0, 0, 0, None, 0, None) == -1
    assert b"ksize" in lib.gs_last_error()
    assert lib.gs_batch_stddev_fwd(None, None, 6, 32, 256, 1e-12, 0, None) == -1  # batch % 4 (ops.py:341)
    assert lib.gs_conv2d_fwd(None, None, None, 1, 7, 8, 32, 32, 3, 2, 1.0, 0, 0, None, 0, None) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a GPU-less host")
def test_no_silent_fallback_without_device():
    from gansynth_amd import kernels, _lib
    old = kernels._K
    kernels._K = None
    try:
        with pytest.raises(_lib.GansynthHipError):
            kernels.get()
    finally:
        kernels._K = old
