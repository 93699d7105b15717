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


def test_bias_fold_planning_without_gpu():
    """Host arithmetic of the deferred bias folds (GS_SUM_PARTIALS / gs_channel_fold_batch): rows a producer leaves, the second-level
    workspace, argument checks -- no launch."""
    import ctypes
    from gansynth_amd import _lib
    lib = _lib.load()
    rows = lib.gs_bias_partial_rows
    assert rows(_lib.BIAS_FROM_CHANNEL_SUM, 8, 8192, _lib.GS_BF16) == 0            # few rows, many channels: summed directly
    assert rows(_lib.BIAS_FROM_CHANNEL_SUM, 8 * 128 * 1024, 32, _lib.GS_BF16) == 2048   # capped
    assert 0 < rows(_lib.BIAS_FROM_ACT_BWD, 8 * 2 * 16, 256, _lib.GS_BF16) <= 64
    assert rows(_lib.BIAS_FROM_PIXEL_NORM_BWD, 8 * 2 * 16, 48, _lib.GS_BF16) == 0  # not a power of two: the entry point refuses it
    assert rows(_lib.BIAS_FROM_PIXEL_NORM_BWD, 8 * 128 * 1024, 32, _lib.GS_BF16) > 64
    jobs = (_lib.GsFoldJob * 3)()
    for jb, (n, c) in zip(jobs, [(2048, 32), (16, 256), (65, 64)]):
        jb.part, jb.out, jb.nparts, jb.c, jb.accumulate = 0x1000, 0x2000, n, c, 1
    ptr = ctypes.cast(jobs, ctypes.c_void_p)
    assert lib.gs_channel_fold_batch_workspace_bytes(ptr, 3) == (32 * 32 + 2 * 64) * 4   # slab sums of the jobs with more than 64 rows
    assert lib.gs_channel_fold_batch(ptr, 0, None, 0, None) == -1
    assert lib.gs_channel_fold_batch(ptr, 3, None, 0, None) != 0 and b"workspace" in lib.gs_last_error()


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


def _job(_lib, ci, co, h, w, n=(8,), stride=1, transposed=0, dtype=None, ksize=3, bias=False, gw=0x1000, stride_ci=0):
    jb = _lib.GsWgradJob()
    for i, k in enumerate(n):
        jb.x[i], jb.gy[i], jb.n[i] = 0x100000 * (i + 1), 0x200000 * (i + 1), k   # (never dereferenced: planning is host arithmetic)
    jb.nsrc, jb.bias_mask, jb.gw, jb.gb = len(n), (1 if bias else 0), gw, (0x3000 if bias else None)
    jb.h, jb.w, jb.ci, jb.co, jb.ksize, jb.stride, jb.transposed = h, w, ci, co, ksize, stride, transposed
    jb.alpha, jb.accumulate, jb.dtype, jb.gw_ci_stride = 0.5, 1, (_lib.GS_BF16 if dtype is None else dtype), stride_ci
    return jb


def test_weight_gradient_job_planning_without_gpu():
    """gs_conv_wgrad_jobs_workspace_bytes is pure host planning: the >= 64-channel bf16 layers of one conv mode share ONE set of
    stream-K partials (blocks + runs, 147.7 KB each) whatever their number, other layers keep their per-layer partials, bad jobs
    are refused (0 bytes + an error message)."""
    import ctypes
    from gansynth_amd import _lib
    lib = _lib.load()

    def nbytes(jobs):
        arr = (_lib.GsWgradJob * len(jobs))(*jobs)
        return lib.gs_conv_wgrad_jobs_workspace_bytes(ctypes.cast(arr, ctypes.c_void_p), len(jobs))

    part = (9 * 64 * 64 + 64) * 4
    layers = [(256, 256, 4, 32), (256, 256, 8, 64), (256, 256, 16, 128), (128, 128, 32, 256), (64, 64, 64, 512)]
    grouped = nbytes([_job(_lib, ci, co, h, w, n=(8, 8), gw=0x1000 * (i + 1)) for i, (ci, co, h, w) in enumerate(layers)])
    runs = sum((ci // 64) * (co // 64) for ci, co, _, _ in layers)
    assert grouped == (256 + runs) * part                     # one group: 256 blocks + one partial per (layer, channel tile) run
    single = sum(lib.gs_conv2d_workspace_bytes(_lib.CONV_BWD_WEIGHT, 16, h, w, ci, co, 3, 1, _lib.GS_BF16) for ci, co, h, w in layers)
    assert grouped * 3 < single                               # ... against blocks x partial per LAYER
    # a second conv mode is a second group, run after the first on the same workspace: the maximum, not the sum
    both = nbytes([_job(_lib, 64, 64, 64, 512), _job(_lib, 64, 128, 64, 512, stride=2, gw=0x2000), _job(_lib, 128, 64, 32, 256, stride=2, transposed=1, gw=0x4000)])
    assert both == max(nbytes([_job(_lib, 64, 64, 64, 512)]), nbytes([_job(_lib, 64, 128, 64, 512, stride=2), _job(_lib, 128, 64, 32, 256, stride=2, transposed=1, gw=0x4000)]))
    # thin / fp32 / 1x1 layers: per-layer partials, added behind the group's
    thin = nbytes([_job(_lib, 32, 32, 128, 1024)])
    assert thin == lib.gs_conv2d_workspace_bytes(_lib.CONV_BWD_WEIGHT, 8, 128, 1024, 32, 32, 3, 1, _lib.GS_BF16)
    assert nbytes([_job(_lib, 64, 64, 64, 512), _job(_lib, 32, 32, 128, 1024, gw=0x2000)]) == nbytes([_job(_lib, 64, 64, 64, 512)]) + thin
    f32 = nbytes([_job(_lib, 64, 64, 8, 64, dtype=_lib.GS_F32)])
    assert f32 == lib.gs_conv2d_workspace_bytes(_lib.CONV_BWD_WEIGHT, 8, 8, 64, 64, 64, 3, 1, _lib.GS_F32)
    # layers without a multi-source kernel are planned one pair at a time
    assert nbytes([_job(_lib, 1, 256, 2, 16, n=(8, 8, 8))]) == 3 * nbytes([_job(_lib, 1, 256, 2, 16)])
    # refused: no sources, a biased transposed conv, a slice stride below the channel count
    assert nbytes([_job(_lib, 64, 64, 8, 64, n=())]) == 0 and b"sources" in lib.gs_last_error()
    assert nbytes([_job(_lib, 64, 64, 8, 64, stride=2, transposed=1, bias=True)]) == 0 and b"transposed" in lib.gs_last_error()
    assert nbytes([_job(_lib, 64, 64, 8, 64, stride_ci=32)]) == 0 and b"gw_ci_stride" in lib.gs_last_error()
    assert lib.gs_conv_wgrad_jobs(None, 0, None, 0, None) == 0


def test_every_environment_switch_is_registered():
    """gansynth_amd/config.py: the Python layer reads its GS_* switches through config.flag / config.value only, and every name is in the
    documented table (no switch sprawl: VERDICT r5 weak 11)."""
    import glob
    from gansynth_amd import config
    used = set()
    for path in glob.glob(os.path.join(ROOT, "gansynth_amd", "*.py")):
        text = open(path).read()
        if not path.endswith("config.py"):
            assert not re.search(r"environ\.get\(\"GS_", text), f"{path} reads a GS_* switch past gansynth_amd.config"
        used |= set(re.findall(r"config\.(?:flag|value)\(\"(GS_[A-Z0-9_]+)\"", text))
    assert used, "no switch found: the pattern is stale"
    assert used <= set(config.KNOBS), sorted(used - set(config.KNOBS))
    unused = set(config.KNOBS) - used - {"GS_FORK_PROBED"}
    assert not unused, f"registered but never read: {sorted(unused)}"
    for name, (kind, text) in config.KNOBS.items():
        assert kind in ("operational", "schedule", "ablation") and len(text) > 10, name
    with pytest.raises(KeyError):
        config.flag("GS_NOT_A_SWITCH")
