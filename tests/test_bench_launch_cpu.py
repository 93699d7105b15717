"""CPU: `python bench.py --gpus N` starts its N ranks itself (the driver may run it with or without a launcher).

`--launch-check` keeps the device out of it: the ranks rendezvous over gloo exactly as they would over RCCL (torch.distributed.run,
127.0.0.1, one process per rank, LOCAL_RANK = the GPU a rank pins), all-reduce their ranks and rank 0 prints one JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n", [2, 4])
def test_bench_self_launches_n_ranks(n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--launch-check"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1, res.stdout   # ONE line, from rank 0
    out = lines[0]
    assert out["n_gpus"] == n and out["ranks_joined"] == n
    assert out["rank_sum"] == n * (n - 1) // 2 and out["local_rank_sum"] == n * (n - 1) // 2   # every rank took its own GPU index


@pytest.mark.timeout(600)
def test_bench_under_a_launcher_does_not_relaunch():
    """The driver's own command line for N > 1 (torch.distributed.run around bench.py): the ranks see RANK and run in place."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    port = 29700 + os.getpid() % 200
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks_joined"] == 2


def test_gpu_count_mismatch_is_an_error():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=1" in (res.stderr + res.stdout)


def test_bench_line_stays_parseable():
    """The driver parses the LAST stdout line out of a bounded tail (round 3: a 20.7 KB line carrying the per-stage table came back
    `parsed: null`).  Assemble the record from a canned measurement set -- the 59 stage rows of the committed round-3 run -- and
    check the line is compact and carries the contract's members; the per-stage table goes to the detail object."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    canned = json.loads(open(os.path.join(ROOT, "profiles", "r03_j_bf16_bench.json")).read().strip().splitlines()[-1])
    stages = canned["stages"]
    assert len(stages) >= 50
    for r in stages:   # (rows of round 3 carry no strict byte count: the helper falls back to `mbytes`)
        assert bench.strict_8d_bytes(r) == r["mbytes"]
    args = argparse.Namespace(batch=8, steps=20, warmup=5, dtype="bf16", no_graphs=False)
    legs = {"spectral": canned["spectral"], "spectral_inverse": canned["spectral_inverse"], "cpu_baseline": canned["cpu_baseline"]}
    fam = (3021, 59.85, 2.97e13, 1.565e11, 21.8, 16.4)
    out, detail = bench.assemble(args, 1, False, 0.1267, 5, fam, stages, {"total": 381, "hip_extension": 361, "torch_native_and_copies": 20},
                                 0.69, 0.7, legs)
    out["detail"] = "profiles/last_bench_detail.json"
    line = bench.compact_line(out)
    assert len(line) < 4096, len(line)
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert "stages" not in rec and len(detail["stages"]) == len(stages)
    rl = rec["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_strict_8d"):
        assert k in rl, k
    assert abs(rl["frac"] - 21.8 / 59.85) < 1e-9              # the per-launch binding-roof figure
    assert 0 < rl["frac_strict_8d"] <= 1.0
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    for leg in ("spectral", "spectral_inverse"):
        assert set(rec[leg]) <= {"value", "unit", "roofline", "cpu_baseline"}
        assert {"bound", "achieved", "peak", "frac"} <= set(rec[leg]["roofline"])
    assert abs(rec["value"] - 8 * 20 / 0.1267) < 1e-6
    # a record that would still be too long loses optional members, never the contract's
    fat = dict(out, junk="x" * 10000)
    rec2 = json.loads(bench.compact_line(fat))
    assert "junk" not in rec2 and "roofline" in rec2 and "cpu_baseline" in rec2


def test_call_log_prices_the_non_conv_families(cpu_backend):
    """bench.price_whole_step's byte side on CPU: the kernel layer's call log (kernels._Accounting, here around the torch-CPU emulation)
    of one reduced iteration, mapped to families -- every family of the elementwise rule gets bytes = operands read once + results
    written once, MFMA-path 3x3 convs are left to the library's own records, Adam is 28 B per parameter."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from gansynth_amd import kernels
    from gansynth_amd.models import GANSynth
    from gansynth_amd.networks import PGGAN
    from gansynth_amd.utils import Dict
    from oracle import torch_ref as R
    pg = PGGAN(min_resolution=[2, 16], max_resolution=[4, 32], min_channels=32, max_channels=64, growing_level=1.0)
    model = GANSynth(pg.generator, pg.discriminator, None, None, None, Dict(R.DEFAULT_HYPER))
    lat, lab, img = R.synthetic_batch(4, rank=0, image_shape=(2, 4, 32))
    lat, lab = lat[:, :64], lab
    K = kernels.get()
    model.discriminator_step(lat, lab, img)   # builds the variables
    with kernels._Accounting(K) as calls:
        model.discriminator_step(lat, lab, img)
        model.generator_step(lat, lab)
    assert not any(hasattr(K, "__dict__") and n in K.__dict__ for n in ("conv2d_fwd", "adam_tf_step"))   # the wrappers are gone again
    fams = {}
    for name, meta, rd, wr in calls:
        f = bench.family_of_call(name, meta)
        if f is not None:
            fams.setdefault(f, []).append((name, meta, rd, wr))
    assert {"thin_convs", "dense", "adam", "batch_stddev", "loss_heads"} <= set(fams), sorted(fams)
    # a 1x1 colour conv is a thin conv, a 3x3 conv with >= 32 channels on both sides is the MFMA path's
    assert bench.family_of_call("conv2d_fwd_bias_act", {"w": (1, 1, 2, 32), "ksize": 1}) == "thin_convs"
    assert bench.family_of_call("conv2d_fwd_bias_act", {"w": (3, 3, 64, 64), "ksize": 3}) is None
    assert bench.family_of_call("conv2d_bwd_weight", {"x": (8, 1, 2, 16), "gy": (8, 256, 2, 16), "ksize": 3}) == "thin_convs"
    name, meta, rd, wr = fams["adam"][0]
    n = 1
    for d in meta["p"]:
        n *= d
    assert rd == 16 * n and wr == 0   # (the log itself: four fp32 operands; bench prices the step at 28 B / parameter)
    for name, meta, rd, wr in fams["thin_convs"]:
        assert rd > 0 and (wr > 0 or "out" in meta)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("stall_in,stall_rank,expect", [(None, 0, "graph-one"), ("graph-one", 1, "graph-serial"),
                                                        ("graph-one,graph-serial,graph-overlapped", 0, "eager-same-stream")])
def test_first_contact_ladder_over_gloo(stall_in, stall_rank, expect):
    """N > 1 cannot hang.  Every rank process is a supervisor (gloo, no GPU) around a `--worker` child; a worker
    that stalls before the end of its warm-up (simulated: one rank sleeps in the named modes) is noticed by its supervisor's watchdog,
    ALL supervisors agree (MIN all-reduce), kill their workers and start the next mode of bench.DP_LADDER together on a fresh
    rendezvous; rank 0 still prints exactly one JSON line, and it names the mode that ran."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GS_LAUNCH_CHECK_LADDER="1", GS_WATCHDOG_WARMUP_S="6", GS_WATCHDOG_IMPORT_S="120", GS_TEST_STALL_RANK=str(stall_rank))
    if stall_in:
        env["GS_TEST_STALL_IN_MODE"] = stall_in
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1, res.stdout
    assert lines[0]["ranks_joined"] == 2 and lines[0]["dp_mode"] == expect, lines[0]
    if stall_in:
        assert "did not get every rank through" in res.stderr
