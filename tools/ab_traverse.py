#!/usr/bin/env python3
"""tools/ab_traverse.py -- interleaved A/B timing of the traversal kernels' launch shapes in ONE process
(cdna_hip_programming.md rule 24): render a few iterations of a workload so the queues hold a realistic
mix of primary and bounce rays, then time `extend` and `connect` on that frozen input for each setting."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c2")
ap.add_argument("--iters", type=int, default=6, help="wavefront iterations to run before freezing the queues")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--configs", default="0:16:0:0:32:128,2:16:0:12:32:128,3:16:0:12:32:128,3:16:0:16:32:128,3:16:0:8:32:128,3:8:0:12:32:128,3:32:0:12:32:128,3:16:0:12:16:128,3:16:0:12:48:128,3:16:0:12:32:64", help="variant:refill_min_idle:waves_per_simd:stack_lds_depth:min_traversing:ticket_chunk:rays_per_block:min_leaves,...")
args = ap.parse_args()

sc = {"c1": scenes.cornell_box, "c2": lambda: scenes.cornell_soup(10000), "c3": lambda: scenes.mesh_scene(706)}[args.workload]()
nodes, prims = binding.bvh_build(sc.triangles)
W, H, N = 1920, 1080, 2097152
flags = binding.TYR_FLAG_PROFILE | (binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0)
r = binding.Renderer(W, H, N, flags=flags, diag=True)  # variants 0-3 live in libtyrant_hip_diag.so
r.load_scene(sc, nodes, prims)
for _ in range(args.iters):
    r.launch_kernels()
# freeze: begin + primary of the next iteration, then extend / shade once so a shadow queue exists
r.stage("begin"), r.stage("primary")
k = r.counters()
print(f"frozen at iteration {args.iters}: n_live {k['n_live']}", flush=True)
configs = [tuple(int(x) for x in c.split(":")) for c in args.configs.split(",")]
ref_hit = None
res = {c: {"extend": [], "connect": []} for c in configs}
r.set_tuning(*configs[0])
r.stage("extend")
r.stage("shade")
print(f"shadow rays {r.counters()['shadow_ray_cnt']}", flush=True)
for rep in range(args.reps):
    for c in configs:
        r.set_tuning(*c)
        r.timings(reset=True)
        r.stage("extend")
        t = r.timings()
        res[c]["extend"].append(t["extend"]["ms"])
        q = r.ray_queue(0, 200000)
        sig = (q["distance"].view(np.uint32).astype(np.uint64).sum(), q["identifier"][q["distance"] < 1e20].astype(np.int64).sum())
        if ref_hit is None:
            ref_hit = sig
        assert sig == ref_hit, f"config {c} changed the hits"
        r.timings(reset=True)
        r.stage("connect")  # adds to the pixels again each time: harmless here
        t = r.timings()
        res[c]["connect"].append(t["connect"]["ms"])
print(f"{'variant:refill:waves:lds:mintrav:chunk:rpb:minleaf':>40s} {'extend ms (min / med)':>24s} {'connect ms (min / med)':>24s}")
for c in configs:
    e, cn = np.array(res[c]["extend"]), np.array(res[c]["connect"])
    print(f"{':'.join(map(str, c)):>40s} {e.min():10.3f} / {np.median(e):8.3f}   {cn.min():10.3f} / {np.median(cn):8.3f}", flush=True)
