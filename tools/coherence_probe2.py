#!/usr/bin/env python3
"""tools/coherence_probe2.py -- round 5: what would placing coherent rays side by side be worth on TODAY's traversal kernel,
at the bench's shape?  C3 at 1080p with the 16.6 M-slot queue; the rays of iteration 2 (the fat bounce launch) and of
iteration 3 (a thin one) that can enter the tree are exported, re-imported in several orders and the extend stage is
timed on each (same rays, same answers, different lane / wave neighbours).  `queue order` is the serial order (what the
export presents); `stable by octant` is what eight segments chosen by direction octant would hold."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

args = dict(a.split("=") for a in sys.argv[1:])
if args.pop("scene", "c3") == "c5":  # the config that IS byte-bound (10 M triangles, 4K): 2 spp in flight = the same 16.6 M slots
    sc = scenes.glass_dof_scene()
    W, H = 3840, 2160
    N = 2 * W * H
else:
    sc = scenes.mesh_scene(706)
    W, H = 1920, 1080
    N = 8 * W * H
nodes, prims = binding.bvh_build(sc.triangles)
flags = binding.TYR_FLAG_PROFILE | (binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0)
r = binding.Renderer(W, H, N, flags=flags)
r.load_scene(sc, nodes, prims)
r.set_budget(N)
for k, v in args.items():
    r.set_tuning(**{k: int(v)})

tv = sc.triangles["vert"].astype(np.float64)
p1, p2 = tv + sc.triangles["e1"], tv + sc.triangles["e2"]
lo = np.minimum(np.minimum(tv, p1), p2).min(0)
hi = np.maximum(np.maximum(tv, p1), p2).max(0)


def enters_root(q):
    o = q["origin"].astype(np.float64)
    d = q["direction"].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0, t1 = (lo - o) * inv, (hi - o) * inv
    tn, tf = np.minimum(t0, t1).max(1), np.maximum(t0, t1).min(1)
    return (tn <= tf) & (tf > 0)


def morton(p, bits):
    l, h = p.min(0), p.max(0)
    g = np.clip(((p - l) / np.maximum(h - l, 1e-9) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(p), dtype=np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((g[:, a] >> b) & 1) << (3 * b + a)
    return code


def time_order(q, name, order):
    qq = q[order] if order is not None else q
    n = len(qq)
    r.import_work_queue(qq, n)
    r.stage("primary")  # budget 0: nothing new, n_live = n
    best = []
    for _ in range(5):
        r.timings(reset=True)
        r.stage("extend")
        best.append(r.timings()["extend"]["ms"])
    best.sort()
    print(f"    {name:44s} extend  min {best[0]:7.3f}  median {best[2]:7.3f} ms", flush=True)


want_first = args_first = bool(int(os.environ.get("TYR_PROBE_FIRST", "0")))  # TYR_PROBE_FIRST=1: also the camera rays (iteration 1), with the orders below
for it in (1, 2, 3):
    for s in ("begin", "primary"):
        r.stage(s)
    k = r.counters()
    if it >= 2 or want_first:
        n = k["n_live"]
        qfull = r.ray_queue(0, n)
        q = qfull[enters_root(qfull)]
        d, o = q["direction"], q["origin"]
        octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)
        hist = np.bincount(octant, minlength=8)
        print(f"iteration {it}: {n} rays in the queue, {len(q)} can enter the tree; by octant {hist.tolist()}", flush=True)
        rng = np.random.default_rng(0)
        time_order(q, "queue order (the serial order)", None)
        time_order(q, "stable by direction octant (8 bins)", np.argsort(octant, kind="stable"))
        time_order(q, "octant, then origin morton 4 bits", np.lexsort((morton(o, 4), octant)))
        time_order(q, "octant, then origin morton 7 bits", np.lexsort((morton(o, 7), octant)))
        time_order(q, "origin morton 7 bits alone", np.argsort(morton(o, 7), kind="stable"))
        time_order(q, "random shuffle", rng.permutation(len(q)))
        # longest-first guesses (a launch ends on its longest rays: would they finish sooner if they started first?): the most
        # grazing rays (|d.z| small: along the terrain), the lowest origins, and the same the other way round
        time_order(q, "most grazing first (|d.z| ascending)", np.argsort(np.abs(d[:, 2]), kind="stable"))
        time_order(q, "most grazing last (|d.z| descending)", np.argsort(-np.abs(d[:, 2]), kind="stable"))
        time_order(q, "lowest origin first (o.z ascending)", np.argsort(o[:, 2], kind="stable"))
        # the imports replaced the work queue: put the real one back (every ray, the serial order) before going on
        r.import_work_queue(qfull, n)
        r.stage("primary")
    for s in ("extend", "shade", "connect", "end"):
        r.stage(s)
