#!/usr/bin/env python3
"""tools/shard_probe.py [c2|c3] [S ...] -- one 1080p 8-spp frame rendered as S concurrent pixel shards on ONE GPU (S
contexts with rank s of S, S streams, S host threads, queue N / S each) against the same frame as one loop.  The drain
of every traversal launch -- each wave finishing the rays it holds once the queue is used up, 40-60 % of a launch
(tools/launch_tail.py) -- is latency bound and leaves the machine mostly idle; another shard's kernels fill it."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
shard_counts = [int(a) for a in sys.argv[2:] if "=" not in a] or [1, 2, 3, 4]
tune = {k: int(v) for k, v in (a.split("=") for a in sys.argv[2:] if "=" in a)}
sc = {"c2": lambda: scenes.cornell_soup(10000), "c3": lambda: scenes.mesh_scene(706)}[wl]()
nodes, prims = binding.bvh_build(sc.triangles)
W, H, SPP = 1920, 1080, 8
flags = binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0
REPS = 5
for S in shard_counts:
    if H % S:
        continue
    rs = []
    for s in range(S):
        r = binding.Renderer(W, H, W * H * SPP // S, rank=s, nranks=S, flags=flags)
        r.load_scene(sc, nodes, prims)
        if tune:
            r.set_tuning(**tune)
        r.render(SPP)  # warm
        rs.append(r)

    def run(r, n):
        for _ in range(n):
            r.reset_accum()
            r.render(SPP)

    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(r, REPS)) for r in rs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = (time.perf_counter() - t0) / REPS
    rays = sum(r.counters()["total_extend_rays"] + r.counters()["total_shadow_rays"] for r in rs) / (REPS + 1)
    assert all(r.counters()["device_error"] == 0 for r in rs)
    print(f"{wl} {tune} shards {S}: {dt * 1e3:.3f} ms per frame, {rays / dt / 1e6:.0f} Mrays/s")
    for r in rs:
        r.close()
