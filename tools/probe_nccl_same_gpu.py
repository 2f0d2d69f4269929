"""Probe: can two ranks share one GPU under the nccl (RCCL) backend on this box?"""
import os
import sys
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
t = torch.full((4,), float(rank + 1), device="cuda", dtype=torch.float64)
dist.all_reduce(t)
print("rank", rank, "allreduce", t.tolist(), flush=True)
a = torch.full((8,), float(rank), device="cuda", dtype=torch.float64)
b = torch.zeros(8, device="cuda", dtype=torch.float64)
peer = 1 - rank
ops = [dist.P2POp(dist.isend, a, peer), dist.P2POp(dist.irecv, b, peer)]
for w in dist.batch_isend_irecv(ops):
    w.wait()
torch.cuda.synchronize()
print("rank", rank, "recv", b.tolist(), flush=True)
dist.destroy_process_group()
