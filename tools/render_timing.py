#!/usr/bin/env python3
"""tools/render_timing.py -- per-kernel milliseconds per render (1080p, 8 spp, queue sized for the GPU) on C2 and C3,
through whichever library TYRANT_HIP_LIBRARY names: the quick A/B for tagged builds (make -C tyrant_amd/csrc tagged TAG=...).
No oracle, no assertions on the picture."""
import sys, os
sys.path.insert(0, os.getcwd())
from tyrant_amd import binding, scenes
import numpy as np
W,H,spp=1920,1080,8
args = [a for a in sys.argv[1:] if "=" not in a]
tune = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a)}  # e.g. min_traversing=24 staged_nodes=0
N = int(args[0]) if args else W*H*spp  # queue size; 2097152 = the reference's
for name, sc in (("c2", scenes.cornell_soup(10000)), ("c3", scenes.mesh_scene(706))):
    bb=scenes.triangle_bboxes(sc.triangles)
    nodes, prims = binding.bvh_build(sc.triangles, bb)
    flags = (0 if os.environ.get('NOPROFILE') else binding.TYR_FLAG_PROFILE) | (1 if sc.triangle_materials else 0)
    r = binding.Renderer(W,H,N, flags=flags)
    r.load_scene(sc,nodes,prims)
    if tune:
        r.set_tuning(**tune)
    r.render(spp)
    r.reset_accum(); r.timings(reset=True)
    import time
    t0=time.perf_counter()
    for _ in range(3):
        r.reset_accum(); r.render(spp)
    dt=(time.perf_counter()-t0)/3
    tm=r.timings()
    assert r.counters()["device_error"] == 0
    print(os.path.basename(os.environ.get("TYRANT_HIP_LIBRARY","default")), tune, name, "N", N, "ms/render %.3f"%(dt*1e3), {k:round(v["ms"]/3,3) for k,v in tm.items() if v["launches"]}, "launches/render", tm["extend"]["launches"]//3, "(no per-kernel events)" if os.environ.get("NOPROFILE") else "")
