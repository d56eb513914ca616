#!/usr/bin/env python3
"""tools/coherence_whatif.py -- what would C5's traversal launch gain from rays that lie in the queue by WHERE THEY ARE?

C5 (10 M triangles, 0.9 GB of records and triangles) is the one configuration bound by bytes: 6.3 TB/s across the fabric, 0.86 of
the guide's random-gather rate.  The queues are "physically unordered, virtually in the serial order" (DESIGN.md 4.1): where a
record lies is the library's choice, the results do not depend on it.  Chunk c of a queue is traversed by a block of XCD c % 8
(segment w of a queue = the chunks w, w + 8, ...: blocks are placed round robin), so a shade that appended a survivor to the
segment of the REGION its origin lies in would give every XCD's 4 MB L2 an eighth of the tree to hold instead of all of it.

This prices that before anything is built, with the staged API: the ray queue of iteration K of a 4K C5 render is exported, the
rays that pass the root box are imported again in several physical orders, and `tyr_stage_extend` is timed on each:

  as_is        the virtual (serial) order
  shuffled     a random permutation (the floor)
  morton       sorted by the Morton code of the origin, laid out one after the other (all XCDs walk the same region at a time)
  morton_xcd   the same order cut into eight equal runs, run w dealt to the chunks w, w + 8, ... (XCD w holds region w)
  hit          sorted by the triangle the ray will hit (a ceiling: nobody knows that beforehand), one after the other
  hit_xcd      ... cut into eight runs for the XCDs

    python tools/coherence_whatif.py [K=2] [reps=3]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tyrant_amd import binding, scenes  # noqa: E402

W, H, N = 3840, 2160, 1 << 25


def morton(p, lo, hi):
    q = np.clip(((p - lo) / (hi - lo) * 1023.0), 0, 1023).astype(np.uint64)
    out = np.zeros(p.shape[0], dtype=np.uint64)
    for b in range(10):
        for a in range(3):
            out |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return out


def in_root(rays, lo, hi):
    o, d = rays["origin"].astype(np.float64), rays["direction"].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0, t1 = (lo - o) * inv, (hi - o) * inv
    tn, tf = np.minimum(t0, t1), np.maximum(t0, t1)
    return (np.nanmax(tn, 1) <= np.nanmin(tf, 1)) & (np.nanmin(tf, 1) > 0)


def xcd_layout(order):
    """order: ray numbers, a multiple of 512 long; run w (an eighth of it) -> the chunks w, w + 8, ..."""
    m = order.shape[0] // 8
    out = np.empty_like(order)
    j = np.arange(m)
    for w in range(8):
        out[((j // 64) * 8 + w) * 64 + j % 64] = order[w * m:(w + 1) * m]
    return out


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    global W, H, N
    if os.environ.get("WL", "c5") == "c3":
        W, H, N = 1920, 1080, 8 * 1920 * 1080
        sc = scenes.mesh_scene(706)
    else:
        sc = scenes.glass_dof_scene(2236)
    nodes, prims = binding.bvh_build(sc.triangles)
    g = binding.Renderer(W, H, N, flags=(1 if sc.triangle_materials else 0) | binding.TYR_FLAG_PROFILE)
    g.load_scene(sc, nodes, prims)
    g.set_budget(16 * W * H)
    for _ in range(K):
        for s in ("begin", "primary", "extend", "shade", "connect", "end"):
            g.stage(s)
    g.stage("begin"), g.stage("primary")
    n_live = g.counters()["n_live"]
    q = g.ray_queue(0, n_live)
    lo, hi = nodes[0]["bounds"][0].astype(np.float64), nodes[0]["bounds"][1].astype(np.float64)
    keep = in_root(q, lo, hi)
    rays = q[keep]
    n = rays.shape[0] // 512 * 512
    rays = rays[:n].copy()
    rays["distance"] = 1e20
    rays["geometry_type"] = 1
    print(f"iteration {K}: {n_live} rays in the queue, {int(keep.sum())} pass the root box, {n} used; bounces 0: {int((rays['bounces'] == 0).sum())}", flush=True)

    def run(order, name):
        r = rays if order is None else rays[order]
        ms = []
        for _ in range(reps):
            g.stage("begin")
            g.import_work_queue(r, n)
            g.set_budget(0)
            g.stage("primary")
            g.timings(reset=True)
            g.stage("extend")
            ms.append(g.timings()["extend"]["ms"])
        assert g.counters()["device_error"] == 0
        print(f"{name:11s} extend {min(ms):7.3f} ms (" + " ".join(f"{m:.3f}" for m in ms) + f")  {n / min(ms) / 1e3:7.1f} Mrays/s", flush=True)
        return min(ms)

    base = run(None, "as_is")
    # the triangle each ray hits: from the run just made
    hitq = g.ray_queue(0, n)
    tri = np.where(hitq["distance"] < 1e20, hitq["identifier"].astype(np.int64), np.int64(1) << 40)
    rng = np.random.default_rng(1)
    run(rng.permutation(n), "shuffled")
    t0 = time.perf_counter()
    om = np.argsort(morton(rays["origin"].astype(np.float64), lo, hi), kind="stable")
    print(f"(morton sort on the host: {time.perf_counter() - t0:.1f} s)", flush=True)
    run(om, "morton")
    run(xcd_layout(om), "morton_xcd")
    oh = np.argsort(tri, kind="stable")
    run(oh, "hit")
    run(xcd_layout(oh), "hit_xcd")
    # origin AND direction: the octant of the direction in front of the origin's code
    d = rays["direction"]
    octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << np.uint64(1)) | ((d[:, 2] < 0).astype(np.uint64) << np.uint64(2)))
    od = np.argsort((morton(rays["origin"].astype(np.float64), lo, hi) >> np.uint64(9)) << np.uint64(12) | (octant << np.uint64(9)) | (morton(rays["origin"].astype(np.float64), lo, hi) & np.uint64(511)), kind="stable")
    run(od, "morton_dir")
    run(None, "as_is")
    if os.environ.get("SHARES"):
        # is what the XCD layouts lose their balance?  The fixed part of the dealing (12 sixteenths: block b owns chunks b, b + G, ...) ties a region
        # to its XCD; the ticketed rest lets a block that is out of work take another XCD's chunks
        for share in (int(x) for x in os.environ["SHARES"].split(",")):
            g.set_tuning(static_share=share)
            print(f"static_share {share}", flush=True)
            run(None, "as_is")
            run(om, "morton")
            run(xcd_layout(om), "morton_xcd")
            run(xcd_layout(oh), "hit_xcd")


if __name__ == "__main__":
    main()
