"""Exercise the RCCL calls bench.py / sharding.py make, at world size 1 (the GPU box has one GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
dist.barrier(device_ids=[0])
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
out = torch.randn(4, 10, 3, dtype=torch.float64, device=dev)
full = torch.empty((1,) + tuple(out.shape), dtype=out.dtype, device=dev)
dist.all_gather_into_tensor(full, out)
torch.cuda.synchronize()
assert torch.equal(full[0], out) and float(t.item()) == 1.5
from nnmnkwii_amd import sharding
import numpy as np
W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
m = torch.randn(6, 50, 6, dtype=torch.float64, device=dev); v = torch.rand(6, 50, 6, dtype=torch.float64, device=dev) + 0.1
y = sharding.mlpg_batch_sharded(m, v, W, gather=True)
print("nccl world-1 ok", tuple(y.shape))
dist.destroy_process_group()
