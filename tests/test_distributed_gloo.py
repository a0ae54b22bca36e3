"""World-size-2 gloo test (CPU) of the N > 1 path: rollout sharding and the fused loss/gradient all-reduce of
diffcloth_amd/distributed.py. The stepper itself needs a GPU; here each rank produces deterministic per-rollout
"gradients" so that the reduction can be checked exactly against the serial sum."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fake_rollout(gid):
    rng = np.random.default_rng(gid)
    return float(rng.uniform()), rng.standard_normal(18).astype(np.float32), rng.standard_normal((4, 3)).astype(np.float32)


def _worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffcloth_amd.distributed import allreduce_loss_and_grads, shard_rollouts
    first, count = shard_rollouts(total)
    loss, g1, g2 = 0.0, np.zeros(18, np.float32), np.zeros((4, 3), np.float32)
    for gid in range(first, first + count):
        l, a, b = _fake_rollout(gid)
        loss += l; g1 += a; g2 += b
    L, (G1, G2) = allreduce_loss_and_grads(loss, [g1, torch.from_numpy(g2)])
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), first=first, count=count, L=L, G1=G1, G2=G2)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_partition_is_exact():
    from diffcloth_amd.distributed import shard_rollouts
    for total in (256, 257, 7, 1):
        for world in (1, 2, 3, 8):
            blocks = [shard_rollouts(total, r, world) for r in range(world)]
            assert sum(c for _, c in blocks) == total
            assert blocks[0][0] == 0
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


def test_two_rank_allreduce_matches_serial_sum(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    total, world = 21, 2
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    ref_l, ref1, ref2 = 0.0, np.zeros(18, np.float32), np.zeros((4, 3), np.float32)
    for gid in range(total):
        l, a, b = _fake_rollout(gid)
        ref_l += l; ref1 += a; ref2 += b
    seen = 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        seen += int(z["count"])
        np.testing.assert_allclose(z["L"], ref_l, rtol=1e-6)
        np.testing.assert_allclose(z["G1"], ref1, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(z["G2"], ref2, rtol=1e-5, atol=1e-6)
        assert z["G2"].shape == (4, 3)
    assert seen == total
