#!/usr/bin/env python3
"""Can two ranks share ONE GPU under RCCL?  (rehearsal of the N>1 path on a 1-GPU box)"""
import os
import sys
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world)
t = torch.full((4,), float(rank), device="cuda:0")
out = [torch.zeros_like(t) for _ in range(world)]
dist.all_gather(out, t)
torch.cuda.synchronize()
print("rank", rank, [float(o[0]) for o in out], flush=True)
dist.destroy_process_group()
