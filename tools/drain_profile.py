#!/usr/bin/env python3
"""tools/drain_profile.py [c2|c3] -- how a traversal launch ends, wave by wave (a -DTYR_LAUNCH_ANATOMY build through
TYRANT_HIP_LIBRARY): for one thin launch of an 8-spp frame (the fourth wavefront iteration: ~2 M bounce rays + the
shadow rays of the one before) every wave's time from its start to "queue used up" and to its exit.  Prints how many of
the grid's waves are still running t microseconds after the median wave found the queue empty -- the machine's load
through the drain -- and the distribution of the waves' drain lengths."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
tune = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:])}
sc = {"c2": lambda: scenes.cornell_soup(10000), "c3": lambda: scenes.mesh_scene(706)}[wl]()
nodes, prims = binding.bvh_build(sc.triangles)
W, H, SPP = 1920, 1080, 8
r = binding.Renderer(W, H, W * H * SPP, flags=binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0)
r.load_scene(sc, nodes, prims)
if tune:
    r.set_tuning(**tune)
r.render(SPP)  # warm
for iters in (2, 4):
    r.reset_accum()
    r.render(SPP, iters)  # the last trace launch of this call left its per-wave records in the queue that is now the WORK queue's partner
    # after `iters` iterations stage_end has swapped the queues `iters` times: the records sit in what is now queue 0 or 1
    best = None
    for which in (0, 1):
        q = r.ray_queue(which, 8192)
        t_exh, t_end = q["distance"], q["identifier"].view(np.float32)
        ok = (t_exh > 0) & (t_end >= t_exh) & (t_end < 1e5)
        if best is None or ok.sum() > best[0].sum():
            best = (ok, t_exh, t_end)
    ok, t_exh, t_end = best
    n = int(ok.sum())
    t_exh, t_end = t_exh[ok], t_end[ok]
    ref = np.median(t_exh)
    print(f"{wl} {tune}: trace launch of iteration {iters - 1}: {n} waves; queue used up at {ref:.0f} us (median wave), last exit at {t_end.max():.0f} us")
    print("   t after the queue ran out [us]:  waves still running")
    for t in (0, 25, 50, 75, 100, 150, 200, 250, 300, 350):
        print(f"   {t:6d}  {(t_end > ref + t).sum():6d}  ({(t_end > ref + t).mean() * 100:5.1f}%)")
    d = t_end - t_exh
    print(f"   a wave's own drain (exit - its 'used up'): median {np.median(d):.0f} us, mean {d.mean():.0f}, 90 % {np.percentile(d, 90):.0f}, 99 % {np.percentile(d, 99):.0f}, max {d.max():.0f}")
