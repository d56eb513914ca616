#!/usr/bin/env python3
"""tests/fuzz_parity.py [n] [seed] -- randomised end-to-end parity: random scenes (triangle soups and height-field meshes of
random size, with and without per-triangle materials, emissive triangles and colour palettes), random resolutions, pixel shards (rank / nranks), queue sizes, cameras and
launch-shape knobs (merged / separate traversal launches, run-ahead, work distribution);
each render is compared with the oracle's: identical iteration and ray counts, queues of the last iteration bit-exact,
radiance within 1e-5 relative.  A checker like the tests (it is the only other place that drives the oracle), not collected by pytest (run time grows with n); prints one line per case."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyorc  # noqa: E402
from tyrant_amd import binding, scenes  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    kind = rng.integers(0, 3)
    if kind == 0:
        sc = scenes.cornell_soup(int(rng.integers(50, 6000)), seed=int(rng.integers(1, 1 << 30)))
    elif kind == 1:
        sc = scenes.mesh_scene(int(rng.integers(4, 90)), seed=int(rng.integers(1, 1 << 30)), spec_fraction=float(rng.uniform(0, 0.5)), refr_fraction=float(rng.uniform(0, 0.2)))
    else:
        sc = scenes.tyrant_default(int(rng.integers(4, 40)), seed=int(rng.integers(1, 1 << 30)))
    W, H = int(rng.integers(17, 140)), int(rng.integers(11, 90))
    nranks = int(rng.choice([1, 1, 1, 2, 3, 4, 8]))  # pixel sharding (rows y % nranks == rank), checked against the oracle's shard
    H = max(nranks, H - H % nranks)
    rank = int(rng.integers(0, nranks))
    N = int(rng.integers(65, 9000))
    spp = int(rng.integers(1, 4))
    cam = scenes.Camera(position=tuple(np.array(sc.camera.position) + rng.normal(0, 3, 3)), direction=sc.camera.direction, up=sc.camera.up,
                        focalDistance=float(rng.uniform(1, 80)), lensRadius=float(rng.choice([0.0, 0.0, rng.uniform(0.1, 3.0)])))
    knobs = dict(refill_min_idle=int(rng.integers(1, 65)), min_traversing=int(rng.integers(1, 65)), ticket_chunk=int(rng.choice([64, 128, 1024])), static_share=int(rng.integers(0, 16)),
                 staged_nodes=int(rng.integers(0, 65)), merge_trace=int(rng.integers(0, 2)), static_interleave=int(rng.integers(0, 2)), run_ahead=int(rng.integers(0, 3)), wide_drain=int(rng.integers(0, 2)),
                 waves_per_simd=int(rng.choice([0, 0, 1, 3])), fold_spheres=int(rng.integers(0, 2)), retire_sky=int(rng.integers(0, 2)), resolve_shadows=int(rng.integers(0, 2)),
                 wide_block_min_items=int(rng.choice([-1, 0, 0, 3 << 20])), fold_prologue=int(rng.integers(0, 2)), scan_in_trace=int(rng.integers(0, 2)), kernel_snapshot=int(rng.integers(0, 2)))  # (0: every traversal launch as 768-thread blocks, six waves per SIMD)
    if sc.triangle_materials and rng.random() < 0.5:  # emissive triangles + light list (TYR_FLAG_LIGHT_LIST)
        lit = rng.choice(len(sc.triangles), size=int(rng.integers(1, min(40, len(sc.triangles)))), replace=False)
        sc.triangles["materialType"][lit] = scenes.LIGHT
        sc.light_list, sc.triangle_emission = True, tuple(float(v) for v in rng.uniform(0.5, 6.0, 3))
        sc.name += "+lights"
    if sc.triangle_materials and rng.random() < 0.4:  # per-triangle colour / emission palette (TYR_FLAG_TRIANGLE_COLORS)
        sc.triangles["pad_"][:, 0] = rng.integers(0, 256, len(sc.triangles)).astype(sc.triangles["pad_"].dtype)
        sc.triangle_colors = True
        sc.palette_color = rng.uniform(0.05, 1.0, (256, 3)).astype(np.float32)
        sc.palette_emission = rng.uniform(0.0, 6.0, (256, 3)).astype(np.float32)
        sc.name += "+colors"
    flags = (1 if sc.triangle_materials else 0) | (8 if sc.light_list else 0) | (16 if sc.triangle_colors else 0)
    bb = scenes.triangle_bboxes(sc.triangles)
    nodes, prims = pyorc.bvh_build(sc.triangles, bb)
    o = pyorc.Oracle(W, H, N, rank=rank, nranks=nranks, flags=flags)
    g = binding.Renderer(W, H, N, rank=rank, nranks=nranks, flags=flags)
    g.set_tuning(layout_on_device=int(rng.integers(0, 2)))  # where tyr_scene_upload's layout pass runs: the same bytes either way
    for r in (o, g):
        r.load_scene(sc, nodes, prims)
        r.set_camera(cam)
    if rng.random() < 0.25:  # ... or the tree built and laid out on the device in one call (tyr_scene_build_upload): the same scene again
        g.build_upload(sc.triangles, bb, want_nodes=False)
    sun = (float(rng.uniform(0, 1)), float(rng.uniform(0.05, 0.49)))
    o.set_sun_position(*sun), g.set_sun_position(*sun)
    g.set_tuning(**knobs)
    ok, why = True, ""
    try:
        io, ig = o.render(spp), g.render(spp)
        ko, kg = o.counters(), g.counters()
        if io != ig or kg["device_error"]:
            ok, why = False, f"iterations {io} vs {ig}, device_error {kg['device_error']}"
        for f in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible", "start_position", "frame"):
            if ko[f] != kg[f]:
                ok, why = False, why + f" {f}: {ko[f]} vs {kg[f]}"
        bo, bg = o.blit_buffer(), g.blit_buffer()
        if not (np.array_equal(bo[:, 3], bg[:, 3]) and np.allclose(bg[:, :3], bo[:, :3], rtol=1e-5, atol=1e-6)):
            ok, why = False, why + " radiance differs"
        # one more iteration, stage by stage, queues bit for bit
        for st in ("begin", "primary", "extend", "shade"):
            o.stage(st), g.stage(st)
        ko, kg = o.counters(), g.counters()
        ns, nh = ko["primary_ray_cnt"], ko["shadow_ray_cnt"]
        qo, qg = o.ray_queue(1, ns), g.ray_queue(1, kg["primary_ray_cnt"])
        same = (ns, nh) == (kg["primary_ray_cnt"], kg["shadow_ray_cnt"]) and all(np.ascontiguousarray(qo[f]).tobytes() == np.ascontiguousarray(qg[f]).tobytes() for f in ("origin", "direction", "direct", "index", "bounces", "lastSpecular"))
        if not (same and o.shadow_queue(nh).tobytes() == g.shadow_queue(nh).tobytes()):
            ok, why = False, why + " queues differ"
    except Exception as e:  # noqa: BLE001
        ok, why = False, repr(e)
    bad += not ok
    print(f"case {case:3d} {sc.name:26s} {len(sc.triangles):7d} tris {W:3d}x{H:<3d} rank {rank}/{nranks} N={N:5d} spp={spp} lens={cam.lensRadius:.2f} {knobs} -> {'ok' if ok else 'FAIL ' + why}", flush=True)
    g.close()
print("failures:", bad)
sys.exit(1 if bad else 0)
