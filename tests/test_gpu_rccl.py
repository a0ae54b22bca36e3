"""Two-rank RCCL smoke of the N > 1 path on real GPUs: bench.py launched exactly as the driver launches it (torch.distributed.run,
one process per GPU, backend "nccl" = RCCL), rollouts sharded over the ranks, one fused all-reduce of the parameter gradient.
Skipped on boxes with fewer than two devices (the gloo version of the same control flow runs on CPU: test_distributed_gloo.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_over_rccl():
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2", "--total-batch", "16", "--grid", "40", "--fold-rows", "3",
           "--fold-gap", "0.05", "--cpu-steps", "0", "--tshirt", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["rollouts_total"] == 16 and d["config"]["rollouts_per_gpu"] == 8
    assert d["config"]["gradients_finite"] and d["config"]["converged_fraction"] == 1.0
    assert len(d["config"]["per_rank_sweep_ms"]) == 2 and all(t > 0 for t in d["config"]["per_rank_sweep_ms"])


def test_bench_two_ranks_on_one_gpu_exercises_the_n_gt_1_control_flow():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, RANK / LOCAL_RANK / WORLD_SIZE from the environment), with
    both ranks on the box's ONE GPU: gloo instead of RCCL (RCCL refuses two ranks on a device) through bench.py's development
    switches DC_BENCH_BACKEND / DC_BENCH_DEVICE. Everything else is the multi-GPU path: rollout sharding, per-rank engines and
    tapes, the barrier-bracketed timed region with the fused all-reduce inside, MAX over the ranks, rank 0's JSON line with the
    roofline block and the CPU baseline. (the first execution of this path must not happen under the driver.)"""
    pytest.importorskip("torch")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DC_BENCH_BACKEND="gloo", DC_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2", "--total-batch", "16", "--grid", "40", "--fold-rows", "3",
           "--fold-gap", "0.05", "--cpu-steps", "1", "--tshirt", "0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 2 and d["scaling"] == "strong"
    assert d["config"]["rollouts_total"] == 16 and d["config"]["rollouts_per_gpu"] == 8
    assert d["config"]["gradients_finite"] and d["config"]["converged_fraction"] == 1.0
    assert len(d["config"]["per_rank_sweep_ms"]) == 2 and all(t > 0 for t in d["config"]["per_rank_sweep_ms"])
    assert d["value"] > 0 and abs(d["value"] - 16 * 2 / (d["ms_per_step"] * 2 * 1e-3)) <= 1e-6 * d["value"]
    assert d["roofline"]["frac"] is not None and 0 < d["roofline"]["frac"] <= 1 and d["roofline"]["bound"] in ("hbm", "lds", "valu")
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
