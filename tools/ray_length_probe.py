#!/usr/bin/env python3
"""tools/ray_length_probe.py [c2|c3] [out.npz] -- how long single rays are, and whether that can be seen coming
(a -DTYR_QUAD_STATS -DTYR_RAY_STEPS build through TYRANT_HIP_LIBRARY: k_trace_flat leaves every ray's quad steps in
the next queue's hit column).  For the trace launches of wavefront iterations 1..4 of an 8-spp 1080p frame: the
distribution of quad steps per extend ray and per shadow ray, how much of the launch's longest rays a few cheap
predictors catch, and -- the question behind it -- how long the launch's tail would be if rays were handed out
longest-predicted first instead of in queue order (a lane-level list-scheduling estimate).  Saves a subsample of
(origin, direction, steps) for offline study."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

args = [a for a in sys.argv[1:] if "=" not in a]
wl = args[0] if args else "c3"
out_path = args[1] if len(args) > 1 else None
sc = {"c2": lambda: scenes.cornell_soup(10000), "c3": lambda: scenes.mesh_scene(706)}[wl]()
nodes, prims = binding.bvh_build(sc.triangles)
W, H, SPP = 1920, 1080, 8
N = W * H * SPP
r = binding.Renderer(W, H, N, flags=binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0)
r.load_scene(sc, nodes, prims)
r.render(SPP)  # warm
LANES = 5120 * 64  # the persistent grid: 5 waves per SIMD


def tail_estimate(steps_in_order: np.ndarray) -> tuple[float, float]:
    """lane-level list scheduling: LANES independent lanes take rays in the given order, a ray costs steps + 1.
    -> (makespan, mean load) in steps; makespan - mean load ~ the launch's drain"""
    cost = steps_in_order.astype(np.float64) + 1.0
    total = cost.sum()
    if cost.size <= LANES:
        return float(cost.max()), total / LANES
    # greedy with a heap is O(n log L); a good closed-form stand-in: lanes finish their share at ~total/LANES, the last
    # LANES rays started end at (start + cost) where start ~ mean load minus what is left
    import heapq
    h = [0.0] * LANES
    heapq.heapify(h)
    for c in cost:
        t = heapq.heappop(h)
        heapq.heappush(h, t + c)
    return float(max(h)), total / LANES


saved = {}
for iters in (2, 3, 4, 5):
    r.reset_accum()
    r.render(SPP, iters)
    c = r.counters()
    n_ext = int(c["n_live"])
    rec = traced = None
    for which in (0, 1):
        q = r.ray_queue(which, min(N, n_ext + (1 << 22)))
        flag = q["identifier"].view(np.float32)[:n_ext]
        ok = np.isin(flag, (0.0, 1.0)).mean() > 0.999 and (q["distance"][:n_ext] >= 0).all() and (q["distance"][:n_ext] == np.floor(q["distance"][:n_ext])).all()
        if ok:
            rec = q
        else:
            traced = q
    if rec is None or traced is None:
        print(f"iteration {iters - 1}: could not tell the queues apart")
        continue
    steps = rec["distance"][:n_ext].astype(np.int32)
    kind = rec["identifier"].view(np.float32)
    sh = (kind[n_ext:] == 1.0) & (rec["distance"][n_ext:] == np.floor(rec["distance"][n_ext:]))
    sh_steps = rec["distance"][n_ext:][sh].astype(np.int32)
    o, d = traced["origin"][:n_ext], traced["direction"][:n_ext]
    print(f"== {wl}: trace launch of iteration {iters - 1}: {n_ext} extend rays (+ {sh.sum()} shadow-ray records behind them)")
    for name, s in (("extend", steps), ("shadow", sh_steps)):
        if s.size == 0:
            continue
        pc = np.percentile(s, (50, 90, 99, 99.9, 99.99))
        print(f"   {name}: mean {s.mean():.1f} steps, 50/90/99/99.9/99.99 % = {pc[0]:.0f}/{pc[1]:.0f}/{pc[2]:.0f}/{pc[3]:.0f}/{pc[4]:.0f}, max {s.max()}, in tree {np.mean(s > 0) * 100:.0f} %; rays > 64: {(s > 64).sum()}, > 96: {(s > 96).sum()}, > 128: {(s > 128).sum()}")
    # ---- predictors of a long extend ray, from the ray alone ----
    zlo, zhi = float(sc.triangles["vert"][:, 2].min()), float(sc.triangles["vert"][:, 2].max())
    dz = np.abs(d[:, 2])
    horiz = 1.0 - dz                       # grazing rays cross many cells
    feats = {
        "1 - |dz| (grazing)": horiz,
        "-dz (going down)": -d[:, 2],
        "origin z (low first)": -o[:, 2],
    }
    long_thr = 64
    is_long = steps > long_thr
    print(f"   long = more than {long_thr} steps: {is_long.sum()} rays ({is_long.mean() * 100:.3f} %)")
    for name, f in feats.items():
        order = np.argsort(-f, kind="stable")
        for frac in (0.05, 0.1, 0.25):
            k = int(frac * n_ext)
            caught = is_long[order[:k]].sum() / max(1, is_long.sum())
            print(f"      {name:24s}: the top {frac * 100:4.0f} % holds {caught * 100:5.1f} % of the long rays")
    # ---- what order would buy: list scheduling on LANES lanes ----
    sub = slice(None)
    if n_ext > 6_000_000:
        print("   (list scheduling skipped: too many rays for the heap estimate)")
    else:
        base = tail_estimate(steps[sub])
        best = tail_estimate(np.sort(steps)[::-1])
        print(f"   list scheduling on {LANES} lanes (steps): queue order makespan {base[0]:.0f} vs mean load {base[1]:.1f}; longest-first (oracle) {best[0]:.0f}")
        for name, f in feats.items():
            order = np.argsort(-f, kind="stable")
            m = tail_estimate(steps[order])
            print(f"      ordered by {name:24s}: makespan {m[0]:.0f}")
        # two buckets: the predicted-long tenth first, the rest in queue order
        f = horiz
        thr = np.quantile(f, 0.9)
        order = np.concatenate([np.nonzero(f >= thr)[0], np.nonzero(f < thr)[0]])
        m = tail_estimate(steps[order])
        print(f"      grazing tenth first, the rest in queue order: makespan {m[0]:.0f}")
    if out_path:
        rng = np.random.default_rng(iters)
        pick = rng.choice(n_ext, size=min(n_ext, 150_000), replace=False)
        pick = np.union1d(pick, np.nonzero(steps > 96)[0])
        saved[f"o{iters - 1}"] = o[pick]
        saved[f"d{iters - 1}"] = d[pick]
        saved[f"steps{iters - 1}"] = steps[pick]
        saved[f"slot{iters - 1}"] = pick.astype(np.uint32)
        saved[f"hitdist{iters - 1}"] = traced["distance"][:n_ext][pick]
if out_path:
    np.savez_compressed(out_path, **saved)
