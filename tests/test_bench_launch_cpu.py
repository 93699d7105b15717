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
