import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.tensor([3.0], device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
import sys; sys.path.insert(0, "/root/repo")
from diffcloth_amd.distributed import allreduce_loss_and_grads
import numpy as np
print("nccl world=1 ok", t.item(), allreduce_loss_and_grads(1.5, [np.arange(3.0)]))
dist.destroy_process_group()
