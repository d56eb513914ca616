#!/usr/bin/env python3
"""tools/concurrency_probe.py [c2|c3] -- do two renders running at once (two contexts, two streams, two host threads)
finish sooner than the same two renders back to back?  If they do, the GPU has idle capacity inside one render (thin
launches, tails, gaps) that overlapping connect(i) with extend(i+1) could use; if not, there is nothing to win."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
sc = {"c2": lambda: scenes.cornell_soup(10000), "c3": lambda: scenes.mesh_scene(706)}[wl]()
nodes, prims = binding.bvh_build(sc.triangles)
W, H, SPP = 1920, 1080, 8
flags = binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0
rs = []
for _ in range(2):
    r = binding.Renderer(W, H, W * H * SPP, flags=flags)
    r.load_scene(sc, nodes, prims)
    r.render(SPP)  # warm
    rs.append(r)


def run(r, n):
    for _ in range(n):
        r.reset_accum()
        r.render(SPP)


REPS = 4
t0 = time.perf_counter()
for r in rs:
    run(r, REPS)
seq = time.perf_counter() - t0
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(r, REPS)) for r in rs]
for t in th:
    t.start()
for t in th:
    t.join()
par = time.perf_counter() - t0
print(f"{wl}: {2 * REPS} renders back to back {seq * 1e3 / (2 * REPS):.2f} ms each; two at a time {par * 1e3 / (2 * REPS):.2f} ms each ({seq / par:.2f}x)")
