#!/usr/bin/env python3
"""tools/iteration_profile.py [c2|c3] -- one 8-spp frame, wavefront iteration by iteration: rays in the queue,
nodes and triangles visited per ray (counting build) and the production kernels' time per ray.
Shows where a frame's traversal time goes once the coherent primary rays are gone."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
tune = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:])}  # e.g. min_traversing=24
sc = {"c1": scenes.cornell_box, "c2": lambda: scenes.cornell_soup(10000), "c3": lambda: scenes.mesh_scene(706)}[wl]()
nodes, prims = binding.bvh_build(sc.triangles)
base = binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0
W, H, SPP = 1920, 1080, 8
rows = {}
for counting in (True, False):
    r = binding.Renderer(W, H, W * H * SPP, flags=base | (binding.TYR_FLAG_COUNT_VISITS if counting else binding.TYR_FLAG_PROFILE))
    r.load_scene(sc, nodes, prims)
    if tune:
        r.set_tuning(**tune)
    for rep in range(2 if not counting else 1):  # production: second pass is the warm one
        r.reset_accum()
        r.set_budget(W * H * SPP)
        prev = r.counters()
        it = 0
        while True:
            r.stage("begin")
            r.stage("primary")
            k0 = r.counters()
            t0 = r.timings(reset=True)
            r.stage("extend")
            r.stage("shade")
            r.stage("connect")
            r.stage("end")
            k = r.counters()
            t = r.timings(reset=True)
            row = rows.setdefault(it, {})
            n_ext = k["total_extend_rays"] - prev["total_extend_rays"]
            n_sh = k["total_shadow_rays"] - prev["total_shadow_rays"]
            if counting:
                row.update(rays=n_ext, shadow=n_sh, nodes=(k["nodes_extend"] - prev["nodes_extend"]) / max(n_ext, 1), tris=(k["tris_extend"] - prev["tris_extend"]) / max(n_ext, 1),
                           cnodes=(k["nodes_connect"] - prev["nodes_connect"]) / max(n_sh, 1))
            else:
                row.update(ext_ms=t["extend"]["ms"], con_ms=t["connect"]["ms"], shade_ms=t["shade"]["ms"])
            prev = k
            it += 1
            if n_ext == 0 or it > 64:
                break
    r.close()
print(f"{wl}: {W}x{H} {SPP} spp, queue {W*H*SPP}, tuning {tune}")
print(" it      rays   nodes/ray tris/ray  extend ms  ps/ray  ps/node |   shadow  nodes/ray connect ms  ps/ray  ps/node | shade ms")
for it in sorted(rows):
    x = rows[it]
    if not x.get("rays"):
        continue
    e, c = x.get("ext_ms", 0.0), x.get("con_ms", 0.0)
    print(f"{it:3d} {x['rays']:9d} {x['nodes']:9.1f} {x['tris']:8.2f} {e:10.3f} {e*1e9/x['rays']:7.0f} {e*1e9/x['rays']/max(x['nodes'],1e-9):8.1f} | {x['shadow']:8d} {x['cnodes']:9.1f} {c:10.3f} {c*1e9/max(x['shadow'],1):7.0f} {c*1e9/max(x['shadow'],1)/max(x['cnodes'],1e-9):8.1f} | {x.get('shade_ms',0):.3f}")
