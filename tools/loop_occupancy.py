#!/usr/bin/env python3
"""tools/loop_occupancy.py -- SIMD occupancy of the extend kernel's loops from the counting build
(tyr_counters.debug): lanes doing work / 64 in the node-test, pop and triangle loops."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
sc = {"c1": scenes.cornell_box, "c2": lambda: scenes.cornell_soup(10000), "c3": lambda: scenes.mesh_scene(706)}[wl]()
nodes, prims = binding.bvh_build(sc.triangles)
flags = binding.TYR_FLAG_COUNT_VISITS | (binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0)
for variant, refill, mintrav in ((1, 16, 32), (2, 16, 16), (2, 16, 32), (2, 16, 48), (2, 4, 32)):
    r = binding.Renderer(1920, 1080, 2097152, flags=flags)
    r.load_scene(sc, nodes, prims)
    r.set_tuning(traversal_variant=variant, refill_min_idle=refill, min_traversing=mintrav)
    r.render(2)
    k = r.counters()
    d = k["debug"]
    rays = k["total_extend_rays"]
    print(f"{wl} variant {variant} refill>={refill} minTrav {mintrav}: rays {rays}  nodes/ray {k['nodes_extend']/rays:.1f} tris/ray {k['tris_extend']/rays:.2f}")
    for name, i in (("node tests", 0), ("stack pops", 2), ("triangle tests", 4), ("refills", 6)):
        w, l = d[i], d[i + 1]
        print(f"   {name:15s} wave-iterations/ray {w/rays*64:8.2f} (per 64 rays)  lane-iterations/ray {l/rays:7.2f}  occupancy {l/max(w,1)/64*100:5.1f} %")
    r.close()
