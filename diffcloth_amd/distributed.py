"""Multi-GPU plumbing of the rollout-parallel stepper.

Rollouts are independent trajectories, so the data path needs no collective: every rank (one process per GPU,
`torch.distributed`, backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests) owns a contiguous block of rollouts
with its own tape and backward sweep. The only exchange is the optimiser-level reduction of the summed loss and
parameter gradients — one fused buffer, one all-reduce per optimiser step, as the reference's trainers would need
(hatController.py accumulates 20 rollouts sequentially on one Simulation object).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_rollouts(total, rank=None, world=None):
    """Contiguous block partition of `total` rollouts; returns (first, count) of this rank.
    Remainders go to the lowest ranks so counts differ by at most one."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def allreduce_loss_and_grads(loss, grads, device=None):
    """Sums `loss` (float) and every array of `grads` (list of numpy arrays or torch tensors) over all ranks with ONE
    all-reduce of a fused float32 buffer (the message is small — latency-bound on xGMI — so fusing is what matters).
    Returns (loss_sum, [grad sums as numpy arrays with the input shapes])."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(loss), [np.asarray(g) if not torch.is_tensor(g) else g.detach().cpu().numpy() for g in grads]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    flats = [torch.as_tensor(np.asarray(g) if not torch.is_tensor(g) else g).reshape(-1).to(torch.float32) for g in grads]
    fused = torch.cat([torch.tensor([float(loss)], dtype=torch.float32)] + flats).to(device)
    dist.all_reduce(fused, op=dist.ReduceOp.SUM)
    fused = fused.cpu()
    out, pos = [], 1
    for g, f in zip(grads, flats):
        n = f.numel()
        shape = tuple(g.shape) if hasattr(g, "shape") else (n,)
        out.append(fused[pos:pos + n].numpy().reshape(shape))
        pos += n
    return float(fused[0]), out
