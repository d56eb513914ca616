#!/usr/bin/env python3
"""tools/launch_tail.py [c2|c3] [knob=value ...] -- anatomy of every extend launch of one 8-spp frame on a
-DTYR_QUAD_STATS build (TYRANT_HIP_LIBRARY): wall time from the first wave's start to the moment the FIRST wave finds the
queue used up ("feed"), and from there to the last wave's exit ("drain": every wave finishing the rays it holds, the
launch as long as its longest ray).  s_memrealtime ticks, 100 MHz."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
tune = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:])}
sc = {"c2": lambda: scenes.cornell_soup(10000), "c3": lambda: scenes.mesh_scene(706)}[wl]()
nodes, prims = binding.bvh_build(sc.triangles)
W, H, SPP = 1920, 1080, 8
flags = binding.TYR_FLAG_PROFILE | (binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0)
r = binding.Renderer(W, H, W * H * SPP, flags=flags)
r.load_scene(sc, nodes, prims)
if tune:
    r.set_tuning(**tune)
M = (1 << 64) - 1
for rep in range(2):  # second pass: warm
    r.reset_accum()
    r.set_budget(W * H * SPP)
    rows = []
    while True:
        r.stage("begin"), r.stage("primary")
        k0 = r.counters()
        r.timings(reset=True)
        r.stage("extend")
        k1 = r.counters()
        t = r.timings()["extend"]["ms"]
        d0, d1 = k0["debug"], k1["debug"]
        # the three words are running maxima over the whole life of the ctx: a launch's values are the new maxima
        start, exh, end = (~d1[13]) & M, (~d1[14]) & M, d1[15]
        rows.append((k1["n_live"], t, (exh - start) / 100.0, (end - exh) / 100.0, (end - start) / 100.0))
        r.stage("shade"), r.stage("connect"), r.stage("end")
        k = r.counters()
        if k["primary_ray_cnt"] == 0 and k["budget_remaining"] == 0:
            break
print(f"{wl} {tune}: extend launches of one frame (warm pass)")
print(" it      rays  hipEvent ms |  feed us  drain us  total us  drain share")
for i, (n, t, feed, drain, tot) in enumerate(rows):
    print(f"{i:3d} {n:9d} {t:12.3f} | {feed:8.1f} {drain:9.1f} {tot:9.1f} {drain / max(tot, 1e-9) * 100:8.1f}%")
