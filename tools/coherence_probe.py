#!/usr/bin/env python3
"""tools/coherence_probe.py -- how much does ray ORDER in the queue matter to the extend kernel?
Takes a real mixed queue (survivors + fresh primaries) a few iterations into a render, re-imports it in
several orders and times extend on each (same rays, same answers, different lane neighbours)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
sc = {"c2": lambda: scenes.cornell_soup(10000), "c3": lambda: scenes.mesh_scene(706)}[wl]()
nodes, prims = binding.bvh_build(sc.triangles)
W, H, N = 1920, 1080, 2097152
flags = binding.TYR_FLAG_PROFILE | (binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0)
r = binding.Renderer(W, H, N, flags=flags)
r.load_scene(sc, nodes, prims)
for _ in range(5):
    r.launch_kernels()
r.stage("begin"), r.stage("primary")
q = r.ray_queue(0, N)
r.set_budget(0)


def time_order(name, order):
    qq = q[order] if order is not None else q
    r.import_work_queue(qq, N)
    r.stage("primary")  # budget 0: nothing new, n_live = N
    best = 1e9
    for _ in range(4):
        r.timings(reset=True)
        r.stage("extend")
        best = min(best, r.timings()["extend"]["ms"])
    print(f"{name:38s} extend {best:7.3f} ms", flush=True)


d = q["direction"]
o = q["origin"]
octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)


def morton(p, bits):
    lo, hi = p.min(0), p.max(0)
    g = np.clip(((p - lo) / np.maximum(hi - lo, 1e-9) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(p), dtype=np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((g[:, a] >> b) & 1) << (3 * b + a)
    return code


# direction cell inside the octant: 4x4 grid on the dominant-axis face
ad = np.abs(d)
major = ad.argmax(1)
u = np.take_along_axis(d, ((major + 1) % 3)[:, None], 1)[:, 0] / np.take_along_axis(ad, major[:, None], 1)[:, 0]
v = np.take_along_axis(d, ((major + 2) % 3)[:, None], 1)[:, 0] / np.take_along_axis(ad, major[:, None], 1)[:, 0]
sign = np.take_along_axis(d, major[:, None], 1)[:, 0] < 0
face = major * 2 + sign
ucell = np.clip(((u + 1) * 2).astype(np.int64), 0, 3)
vcell = np.clip(((v + 1) * 2).astype(np.int64), 0, 3)
dircell = face * 16 + ucell * 4 + vcell  # 96 direction bins

rng = np.random.default_rng(0)
time_order("queue order (as rendered)", None)
time_order("stable by direction octant (8 bins)", np.argsort(octant, kind="stable"))
time_order("stable by cube-map cell (96 bins)", np.argsort(dircell, kind="stable"))
time_order("octant, then origin morton 4 bits", np.lexsort((morton(o, 4), octant)))
time_order("cube cell, then origin morton 5 bits", np.lexsort((morton(o, 5), dircell)))
time_order("origin morton 6 bits, then cube cell", np.lexsort((dircell, morton(o, 6))))
time_order("random shuffle", rng.permutation(N))
