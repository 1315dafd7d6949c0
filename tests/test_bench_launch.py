"""bench.py --gpus N starts its own N ranks (python/sglang/bench_one_batch.py:527-546 spawns one process per tp_rank):
the parent relays rank 0's JSON line, carries --gpus and the observed world size on it, and fails when any rank fails."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def _has_gpu():
    import torch

    return torch.cuda.is_available()


def test_world_size_must_agree_with_gpus_flag():
    """Under a launcher (WORLD_SIZE set) --gpus has to say the same: a flat 8-GPU scaling run must not be possible."""
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--model", "tiny"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "must agree" in r.stderr


@pytest.mark.skipif(_has_gpu(), reason="needs a box WITHOUT a GPU: the ranks then fail at start-up, which is the case under test")
def test_parent_fails_when_a_rank_fails():
    """Without a GPU every rank exits non-zero at once ('bench.py needs an MI355X'): the parent must report that, not hang."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--model", "tiny", "--no-cpu-baseline"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "a rank exited with code" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
@pytest.mark.parametrize("all_reduce", ["auto", "p2p"])
def test_bench_gpus_2_spawns_two_ranks(all_reduce):
    """Two ranks on ONE GPU over gloo (the rehearsal transport); p2p = the P2P communicator with both collectives inside the
    captured graph, auto = host-staged collectives, eager by design and marked so in the metric name."""
    cmd = [sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--model", "tiny", "--no-cpu-baseline", "--batch", "4",
           "--seq-len", "64", "--steps", "3", "--warmup", "1", "--all-reduce", all_reduce]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size_observed"] == 2
    assert out["config"]["parallelism"].startswith("tp2")
    if all_reduce == "p2p":
        assert out["config"]["hip_graph"] is True and out["config"]["all_reduce"].startswith("p2p")
        assert "EAGER" not in out["metric"]
    else:
        assert out["config"]["hip_graph"] is False and "EAGER" in out["metric"]
