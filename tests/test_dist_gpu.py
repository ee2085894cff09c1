"""RCCL really executes (missing 1): the default `nccl` backend of torch.distributed IS RCCL on ROCm; every N > 1
test of this repo forces gloo (a 1-GPU box cannot host two RCCL ranks: one rank per device), so the nccl branch of dist.init(),
barrier(device_ids=...) and the device-tensor all-reduce had never run.  Here they run in a one-rank group on the box's GPU, in a
child process (a process group is process-global state)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:    # as bench.self_launch picks its rendezvous port
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_one_rank_group_runs_every_collective_helper(cuda):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DYN_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["counts"] == [3, 1, 4, 159] and res["ids"] == ["a", "b"] and res["max"] == 2.5 and res["tensor_ok"]


def test_bench_gpus2_self_launched_runs_the_workload_on_two_ranks(cuda):
    """`python bench.py --gpus 2` (the driver's own command form, no torchrun environment): bench.py starts its two ranks itself
    and rank 0 prints the one JSON line.  Two gloo ranks share the one GPU of this box (RCCL wants one device per rank); small
    recordings, the timed region / barrier / max-over-ranks / counter all-reduce are the real ones."""
    env = dict(os.environ, DYN_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--seconds", "200",
                        "--prewarm_s", "0", "--no_cpu_baseline", "--chains", "1", "--side_steps", "0"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 200.0 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3       # whole-job aggregate over both ranks
    assert "cpu_baseline" not in d and d["config"]["sharding"].startswith("2 ranks")
