"""oracle/cpu_baseline.py -- the CPU baseline leg of bench.py (TEST INFRASTRUCTURE, like everything under oracle/: only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg import it; the product path never does).

What is timed, on ONE host core: the oracle (a serial CPU port of the reference's loop) on the first iterations of the
workload; then ONE ray set -- the queue of the next iteration, bounce rays included -- through the port's traversal and,
when oracle/_ref is built, through the reference's own CachedBVH::intersect (bvh.h:118-161), side by side; and the SAH
build of the scene's tree by the port and by the reference's own bvh.cpp."""
from __future__ import annotations

import os
import time


def cpu_baseline(sc, W, H, N, iterations, tri_materials, spp):
    """the oracle (a serial CPU port of the reference's loop) on the first `iterations` wavefront iterations of the same
    workload: same scene, resolution, queue size, seeds; 1 core.  Then ONE ray set -- the queue of the next iteration,
    bounce rays included -- through the port's traversal and through the reference's own, side by side."""
    import ctypes

    import numpy as np

    from oracle import pyorc
    from tyrant_amd import scenes  # (scene records and dtypes only: no device code)

    t0 = time.perf_counter()
    nodes, prims = pyorc.bvh_build(sc.triangles, scenes.triangle_bboxes(sc.triangles))
    t_build = time.perf_counter() - t0
    o = pyorc.Oracle(W, H, N, flags=1 if tri_materials else 0)
    o.load_scene(sc, nodes, prims)
    o.set_budget(spp * W * H)  # the same primary-ray budget as the timed render: no top-up once it is spent
    t0 = time.perf_counter()
    for _ in range(iterations):
        o.launch_kernels()
    dt = time.perf_counter() - t0
    k = o.counters()
    rays = k["total_extend_rays"] + k["total_shadow_rays"]
    whole = {
        "value": round(rays / dt / 1e6, 4),
        "unit": "Mrays/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {iterations} wavefront iterations of the same workload ({k['total_extend_rays']} extend + {k['total_shadow_rays']} shadow rays, all stages) in {dt:.1f} s; mostly rays that end at the root box",
    }
    # one common ray set: the first 2 Mi rays of the next iteration's queue, traversal only (no spheres, no shading)
    o.stage("begin"), o.stage("primary")
    n = min(o.counters()["n_live"], 1 << 21)
    q = np.ascontiguousarray(o.ray_queue(0, n))
    q["distance"] = 1e20  # VERY_FAR, variables.h:13: extend starts every ray there
    L = pyorc.lib()
    nd, pr = np.ascontiguousarray(nodes), np.ascontiguousarray(prims)
    qa = q.copy()
    hit_p = np.zeros(n, dtype=np.int32)
    t0 = time.perf_counter()
    L.orc_bvh_intersect_batch(nd.ctypes.data, pr.ctypes.data, qa.ctypes.data, n, hit_p.ctypes.data)
    dtp = time.perf_counter() - t0
    hits_port = int(hit_p.sum())
    trace = {
        "ray_set": f"the first {n} rays of iteration {iterations + 1}'s queue (survivors of {iterations} bounces in front, fresh primary rays behind), BVH only",
        "port": {"value": round(n / dtp / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port", "seconds": round(dtp, 6), "hits": int(hits_port), "note": "orc_bvh_intersect_batch (the oracle's restatement of bvh.h:118-161), one call for the batch"},
    }
    R = pyorc.ref()
    if R is not None:
        qb = q.copy()
        hit = np.zeros(n, dtype=np.int32)
        t0 = time.perf_counter()
        R.ref_bvh_intersect(nd.ctypes.data_as(ctypes.c_void_p), pr.ctypes.data_as(ctypes.c_void_p), qb.ctypes.data_as(ctypes.c_void_p), n, hit.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), None)
        dtr = time.perf_counter() - t0
        same = bool(np.array_equal(qa["distance"].view(np.uint32), qb["distance"].view(np.uint32)))
        trace["reference"] = {"value": round(n / dtr / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "reference", "seconds": round(dtr, 6), "hits": int(hit.sum()),
                              "note": "CachedBVH::intersect of the reference's bvh.h (oracle/_ref/libref_traverse.so), one call for the batch", "distances_bit_identical_to_port": same}
    # the reference's own builder on the same triangles (bvh.cpp:3-225 compiled into oracle/_ref), beside the port's
    builds = {"port": round(t_build, 6), "port_us": round(t_build * 1e6, 1)}  # full precision: a 36-triangle tree builds in 0.2 ms (round 3 rounded this to 0.0)
    if R is not None and hasattr(R, "ref_bvh_build"):
        tp = np.ascontiguousarray(sc.triangles.copy())
        bb = np.ascontiguousarray(scenes.triangle_bboxes(sc.triangles))
        nd2 = np.zeros(max(2 * tp.shape[0] - 1, 1), dtype=scenes.NODE_DTYPE)
        t0 = time.perf_counter()
        nn = R.ref_bvh_build(tp.ctypes.data, tp.shape[0], bb.ctypes.data, nd2.ctypes.data, 2)
        builds["reference"] = round(time.perf_counter() - t0, 6)
        builds["reference_nodes_identical_to_port"] = bool(nn == nodes.shape[0] and nd2[:nn].tobytes() == nodes.tobytes())
    # lead with the like-for-like figure: ONE ray set through the reference's own traversal (else the port's)
    lead = trace.get("reference", trace["port"])
    out = {
        "value": lead["value"],
        "unit": "Mrays/s",
        "cores": 1,
        "kind": lead["kind"],
        "sample": trace["ray_set"] + f": {n} rays in {lead['seconds']} s, traversal only, one core",
        "trace_same_ray_set": trace,
        "whole_path_first_iterations": whole,
        "bvh_build_s": builds,
        "host_cpus": os.cpu_count(),
    }
    return out
