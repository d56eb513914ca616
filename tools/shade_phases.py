#!/usr/bin/env python3
"""tools/shade_phases.py [c2|c3] -- diagnostic build only (make ... EXTRA_HIPFLAGS=-DTYR_SHADE_TIMING, loaded through
TYRANT_HIP_LIBRARY): where a tile of k_shade spends its time, per wavefront iteration, in s_memtime ticks (100 MHz)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
sc = {"c2": lambda: scenes.cornell_soup(10000), "c3": lambda: scenes.mesh_scene(706)}[wl]()
nodes, prims = binding.bvh_build(sc.triangles)
W, H, SPP = 1920, 1080, 8
r = binding.Renderer(W, H, W * H * SPP, flags=(binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0) | binding.TYR_FLAG_PROFILE)
r.load_scene(sc, nodes, prims)
names = ["load+shade", "ranks+barrier", "places (the append atomics of the tile before) arrive", "(unused)", "copy out+barrier", "stage+pixel atomics"]
for rep in range(2):
    r.reset_accum()
    r.set_budget(W * H * SPP)
    prev = r.counters()["debug"]
    for it in range(6):
        for st in ("begin", "primary", "extend"):
            r.stage(st)
        t0 = r.timings(reset=True)
        r.stage("shade")
        t = r.timings(reset=True)
        r.stage("connect"), r.stage("end")
        d = r.counters()["debug"]
        delta = [d[i] - prev[i] for i in range(8)]
        prev = d
        if rep == 1:
            tiles = max(delta[7], 1)
            print(f"it {it}: shade {t['shade']['ms']:.3f} ms, {tiles} tiles by wave 0; ticks per tile: " + ", ".join(f"{n} {delta[i]/tiles:.0f}" for i, n in enumerate(names)) + f"; sum {sum(delta[:6])/tiles:.0f}")
