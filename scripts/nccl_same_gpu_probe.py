"""Probe: can two RCCL ranks share ONE GPU on this stack?  (If yes the N=2 nccl leg can be
rehearsed on a 1-GPU box; if RCCL refuses, the answer is recorded in DESIGN.md.)"""
import os, sys
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world)
t = torch.full((1024,), rank + 1, dtype=torch.int64, device="cuda:0")
dist.all_reduce(t)
torch.cuda.synchronize()
print("rank", rank, "all_reduce ->", int(t[0].item()), flush=True)
dist.destroy_process_group()
