#!/usr/bin/env python3
"""tools/nccl_gather_probe.py -- one-rank RCCL group on cuda:0: checks that the calls bench.py makes for N > 1
(barrier, dist.gather via tyrant_amd.dist.agree_gather_works / gather_rows, dist.reduce) are accepted by the nccl
backend of this torch build.  (Two ranks cannot share one GPU under RCCL, so this is as far as a 1-GPU box goes.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1"), os.environ.setdefault("LOCAL_RANK", "0")
import torch
import torch.distributed as dist

from tyrant_amd import dist as tdist

torch.cuda.set_device(0)
tdist.init_process_group("nccl")
dist.barrier()
print("gather probe agreed:", tdist.agree_gather_works("cuda:0"))
H, W = 8, 4
a = torch.arange(H * W * 4, dtype=torch.float32, device="cuda:0")
t = torch.full((4,), 7.0, device="cuda:0")
parts = [torch.empty_like(t)]
dist.gather(t, gather_list=parts, dst=0)
assert float(parts[0][0]) == 7.0
dist.reduce(a, dst=0, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
print("nccl gather / reduce / barrier ok")
dist.destroy_process_group()
