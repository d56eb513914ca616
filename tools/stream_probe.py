"""tools/stream_probe.py [knob=value ...] -- three C3 renders (1080p, 8 spp, every primary ray in flight) with the given tuning
knobs, for `rocprofv3 --kernel-trace` (tools/render_timeline.py prints the last render's kernels)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

from tyrant_amd import binding, scenes  # noqa: E402

knobs = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:])}
renders = knobs.pop("renders", 3)
queue = knobs.pop("queue", 1920 * 1080 * 8)
profile = knobs.pop("profile", 0)  # 1: hipEvent pairs around every stage (TYR_FLAG_PROFILE), the stages' times of the last render printed
sc = scenes.mesh_scene(706)
nodes, prims = binding.bvh_build(sc.triangles)
g = binding.Renderer(1920, 1080, queue, flags=1 | (2 if profile else 0))
g.load_scene(sc, nodes, prims)
g.set_tuning(**knobs)
for r in range(renders):
    g.reset_accum()
    if profile:
        g.timings(reset=True)
    t0 = time.perf_counter()
    it = g.render(8)
    print(f"render {r}: {(time.perf_counter() - t0) * 1e3:.3f} ms, {it} iterations, err {g.counters()['device_error']}", flush=True)
if profile:
    print("stages of the last render (ms):", {k: (round(v["ms"], 4), v["launches"]) for k, v in g.timings().items() if v["launches"]})
if os.environ.get("TYR_PROBE_DEBUG"):
    d = g.counters()["debug"]
    tiles = max(d[7], 1)
    names = ["load+shade", "ranks+barrier", "places arrive", "draw (+ wait for the traversal)", "copy out+barrier(+publish)", "stage+pixel atomics"]
    print(f"shade tiles (all renders, all launches) {tiles}; ticks (10 ns) per tile: " + ", ".join(f"{n} {d[i] / tiles:.0f}" for i, n in enumerate(names)) + f"; sum {sum(d[:6]) / tiles:.0f}")
