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
