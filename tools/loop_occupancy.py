#!/usr/bin/env python3
"""tools/loop_occupancy.py [c2|c3] [knob=value ...] -- SIMD lane occupancy of the extend kernel's loops from
tyr_counters.debug: lanes doing work / 64 in the node-test, pop and triangle loops and in the refills.
With the standard library this is the counting build (pair nodes, variant 2); with a diagnostic build
(make ... EXTRA_HIPFLAGS=-DTYR_QUAD_STATS, loaded through TYRANT_HIP_LIBRARY) and `production=1` it is the production
quad kernel, iteration by iteration."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
kv = dict(a.split("=") for a in sys.argv[2:])
production = int(kv.pop("production", 0))
tune = {k: int(v) for k, v in kv.items()}
sc = {"c1": scenes.cornell_box, "c2": lambda: scenes.cornell_soup(10000), "c3": lambda: scenes.mesh_scene(706)}[wl]()
nodes, prims = binding.bvh_build(sc.triangles)
W, H, SPP = 1920, 1080, 8
flags = (0 if production else binding.TYR_FLAG_COUNT_VISITS) | (binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0)
r = binding.Renderer(W, H, W * H * SPP, flags=flags)
r.load_scene(sc, nodes, prims)
if tune:
    r.set_tuning(**tune)
r.set_budget(W * H * SPP)
prev = r.counters()
print(f"{wl} {'production quad kernel' if production else 'counting build (pair nodes)'} {tune}")
print(" it      rays | node trips/64 rays  lanes | pop trips  lanes | triangle trips  lanes | refills  lanes" + (" || descent trips: lanes at a leaf / without a ray / finished, stale pops per trip" if production else ""))
for it in range(6):
    r.launch_kernels()
    k = r.counters()
    d = [k["debug"][i] - prev["debug"][i] for i in range(16)]
    rays = k["total_extend_rays"] - prev["total_extend_rays"]
    prev = k
    if rays == 0:
        break
    f = lambda i: f"{d[i] / rays * 64:9.2f} {d[i + 1] / max(d[i], 1) / 64 * 100:5.1f}%"  # noqa: E731
    census = ""
    if production and d[8]:
        # TYR_QUAD_STATS: lane states at the top of every descent trip (64 lanes = 100 %)
        census = f" || {d[9] / d[8] / 64 * 100:5.1f}% {d[10] / d[8] / 64 * 100:5.1f}% {d[11] / d[8] / 64 * 100:5.1f}%  {d[12] / d[8]:5.2f}"
    print(f"{it:3d} {rays:9d} | {f(0)}        | {f(2)} | {f(4)}      | {f(6)}{census}")
