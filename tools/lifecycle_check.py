#!/usr/bin/env python3
"""tools/lifecycle_check.py -- create / render / destroy many contexts of varying size in one process and watch the free
device memory: tyr_destroy must give everything back (rocm-smi style check through torch.cuda.mem_get_info)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tyrant_amd import binding, scenes

sc = scenes.cornell_soup(3000)
nodes, prims = binding.bvh_build(sc.triangles)
torch.cuda.init()
free_cold = torch.cuda.mem_get_info()[0]
free0 = None
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 41
for i in range(n_ctx):
    W, H, N = 64 + 8 * (i % 7), 48 + 4 * (i % 5), 4096 << (i % 9)
    r = binding.Renderer(W, H, N)
    r.load_scene(sc, nodes, prims)
    r.render(1 + i % 3)
    assert r.counters()["device_error"] == 0
    r.close()
    if i == 9:
        # after one context of every queue size: the runtime's own one-time allocations are in (code objects, and
        # per hardware queue -- the ctx stream and the side stream of the deferred connect -- scratch and ring
        # buffers, which the runtime pools and keeps: ~470 + ~100 MiB, constant from here to 240 contexts)
        free0 = torch.cuda.mem_get_info()[0]
    elif i > 9 and i % 40 == 0:
        print(f"  after {i} contexts: {(free0 - torch.cuda.mem_get_info()[0]) >> 20} MiB below that", flush=True)
free1 = torch.cuda.mem_get_info()[0]
print(f"free device memory: cold {free_cold >> 20} MiB, after the first ten contexts {free0 >> 20} MiB, after {n_ctx - 10} more {free1 >> 20} MiB, difference {(free0 - free1) >> 20} MiB", flush=True)
assert free0 - free1 < (16 << 20), "device memory was not returned"
print("lifecycle ok")
