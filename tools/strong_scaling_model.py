#!/usr/bin/env python3
"""tools/strong_scaling_model.py [c3] -- what the ONE GPU of a gpurun box can say about BASELINE config C4 (the
1 M-triangle scene, 1080p, 64 spp in total, rows y % R == rank): the time one rank needs for its shard at R = 1, 2, 4, 8
(first and last rank), measured back to back on the same device.  T(1) / max_rank T(R) is the speed-up the path would
reach at R GPUs if the combine were free; the combine's own cost (a gather of 33.2 MB / R per rank over xGMI, overlapped
with nothing in this estimate) is added from the link rate.  Round 6: instead of a free combine the model also charges what
CAN be measured on one device -- `tyr_dist_combine(GATHER)` + `tyr_dist_wait` on a one-rank communicator over the whole
1080p frame (pack every row, scatter every row: an upper bound of what the root does locally at any R, and of what a rank packs)
and the host's work between two renders (`tyr_set_frame` + `tyr_reset_accum`, as bench.py's step does).  Not a scaling
measurement: the driver's SCALE run is, and no N > 1 exchange has run on hardware."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (first: one HIP runtime per process, tests/conftest.py)

from tyrant_amd import binding, scenes  # noqa: E402

W, H, SPP = 1920, 1080, 64
sc = scenes.mesh_scene(706)
nodes, prims = binding.bvh_build(sc.triangles)
flags = binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0
XGMI_GBS = 153.0 * 0.8  # one link, 80 % of its rate (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point)
# what one device can measure of the exchange and of the host's per-render work
g = binding.Renderer(W, H, W * H, flags=flags)
g.load_scene(sc, nodes, prims)
g.render(1)
frame = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
d = binding.Dist(g, binding.dist_unique_id(), 0, 1)
torch.cuda.synchronize()
tc, th = [], []
for _ in range(30):
    t0 = time.perf_counter()
    d.combine(frame.data_ptr(), mode=binding.TYR_DIST_GATHER, root=0)
    d.wait()
    tc.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    g.set_frame(1)
    g.reset_accum()
    th.append(time.perf_counter() - t0)
combine_s, host_s = statistics.median(tc[5:]), statistics.median(th[5:])
d.close()
g.close()
print(f"measured on this device: tyr_dist_combine(GATHER) + tyr_dist_wait on a one-rank communicator over the whole {W}x{H} frame {combine_s * 1e6:.0f} us (median of 25);"
      f" tyr_set_frame + tyr_reset_accum between two renders {host_s * 1e6:.0f} us")
t1 = None
print(f"C4 shard times on one MI355X: {W}x{H}, {SPP} spp in total, rows y % R == rank, queue = min(spp x local pixels, 32 Mi)")
for R in (1, 2, 4, 8):
    worst = 0.0
    per = []
    for rank in sorted({0, R - 1}):
        N = min(SPP * W * (H // R), 1 << 25)
        r = binding.Renderer(W, H, N, rank=rank, nranks=R, flags=flags)
        r.load_scene(sc, nodes, prims)
        r.render(SPP)
        ts = []
        for _ in range(3):
            r.reset_accum()
            t0 = time.perf_counter()
            r.render(SPP)
            ts.append(time.perf_counter() - t0)
        c = r.counters()
        rays = (c["total_extend_rays"] + c["total_shadow_rays"]) / 4  # counters run since the ctx was made: one warm-up + three timed renders
        assert c["device_error"] == 0
        r.close()
        t = min(ts)
        per.append((rank, t, rays))
        worst = max(worst, t)
    if R == 1:
        t1 = worst
    gather_s = (W * H * 16 / R) * (R - 1) / R / (XGMI_GBS * 1e9) if R > 1 else 0.0  # every rank receives the other ranks' rows; per-link bound
    line = ", ".join(f"rank {k}: {t * 1e3:.2f} ms ({n / t / 1e6:.0f} Mrays/s)" for k, t, n in per)
    charged = worst + host_s + (combine_s + gather_s if R > 1 else 0.0)  # nothing overlapped: host work, the local pack / scatter and the link transfer one behind the other
    print(f"  R = {R}: {line}; speed-up if the combine were free {t1 / worst:.2f} (efficiency {t1 / worst / R:.2f}); with a {gather_s * 1e3:.2f} ms gather at {XGMI_GBS:.0f} GB/s per link: {t1 / (worst + gather_s):.2f};"
          f" with the measured host work and one-rank combine charged too, nothing overlapped: {(t1 + host_s) / charged:.2f}")
