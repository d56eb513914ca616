"""The streamed tail (TYR_TUNE_STREAM_TAIL) against the oracle and against the launch-per-iteration path, on a GPU box:
   python tools/stream_tail_check.py parity     small scenes + C3 at 1080p / 1 spp, every counter and the radiance
   python tools/stream_tail_check.py time       C3 at 1080p / 8 spp, ms per render with the tail streamed and not"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (one HIP runtime per process: tests/conftest.py)

from tyrant_amd import binding, scenes  # noqa: E402

FIELDS = ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible", "start_position", "frame", "primary_ray_cnt", "shadow_ray_cnt", "n_live")


def parity():
    from conftest import built_scene
    from oracle import pyorc

    cases = [("cornell36", 128, 128, 16384, 2, 0), ("tyrant_default", 160, 96, 10000, 2, 0), ("cornell_soup10k", 128, 72, 128 * 72 * 3, 3, 0), ("mesh128", 96, 96, 8192, 2, 0), ("glass_dof48", 128, 72, 8192, 3, 0),
             ("cornell_area_light", 128, 96, 8192, 4, 0), ("cornell_colored", 128, 96, 8192, 4, 0), ("mesh706", 1920, 1080, 1920 * 1080, 1, 1), ("mesh706", 1920, 1080, 1 << 21, 2, 1)]
    bad = 0
    for name, W, H, N, spp, fl in cases:
        sc, nodes, prims = built_scene(name)
        flags = fl | (1 if sc.triangle_materials else 0) | (8 if sc.light_list else 0) | (16 if sc.triangle_colors else 0)
        o = pyorc.Oracle(W, H, N, flags=flags & 25)
        o.load_scene(sc, nodes, prims)
        it_o = o.render(spp)
        ko, bo = o.counters(), o.blit_buffer()
        for shade_per_cu in (1,):
            g = binding.Renderer(W, H, N, flags=flags)
            g.load_scene(sc, nodes, prims)
            g.set_tuning(run_ahead=0, stream_tail=1, stream_shade_per_cu=shade_per_cu)
            t0 = time.perf_counter()
            try:
                it_g = g.render(spp)
            except Exception as e:  # noqa: BLE001
                print(f"FAIL {name} {W}x{H} N={N} spp={spp}: {e}; counters {g.counters()}", flush=True)
                bad += 1
                continue
            dt = time.perf_counter() - t0
            kg, bg = g.counters(), g.blit_buffer()
            diffs = [(f, ko[f], kg[f]) for f in FIELDS if ko[f] != kg[f]]
            cnt_ok = np.array_equal(bo[:, 3], bg[:, 3])
            rad_ok = np.allclose(bg[:, :3], bo[:, :3], rtol=1e-5, atol=1e-6)
            ok = it_o == it_g and not diffs and cnt_ok and rad_ok and kg["device_error"] == 0
            bad += 0 if ok else 1
            print(f"{'ok  ' if ok else 'FAIL'} {name} {W}x{H} N={N} spp={spp}: iterations {it_o}/{it_g}, err {kg['device_error']}, diffs {diffs}, counts {cnt_ok}, radiance {rad_ok}, {dt * 1e3:.1f} ms", flush=True)
    print("parity:", "all ok" if bad == 0 else f"{bad} FAILED", flush=True)
    return bad


def timing():
    sc = scenes.mesh_scene(706)
    nodes, prims = binding.bvh_build(sc.triangles)
    W, H, spp = 1920, 1080, 8
    for N in (W * H * spp, 1 << 21):
        for label, knobs in (("per-iteration launches", dict(stream_tail=0)), ("streamed tail", dict(stream_tail=1, run_ahead=0)), ("streamed tail, 2 shade blocks / 3 trace blocks per CU", dict(stream_tail=1, run_ahead=0, stream_shade_per_cu=2, stream_trace_per_cu=3)),
                             ("per-iteration launches, spheres folded into shade", dict(stream_tail=0, fold_spheres=1))):
            g = binding.Renderer(W, H, N, flags=1)
            g.load_scene(sc, nodes, prims)
            g.set_tuning(**knobs)
            ts = []
            for r in range(6):
                g.reset_accum()
                t0 = time.perf_counter()
                it = g.render(spp)
                ts.append(time.perf_counter() - t0)
            k = g.counters()
            rays = (k["total_extend_rays"] + k["total_shadow_rays"]) / 6
            best = min(ts[1:])
            print(f"N={N:9d} {label:60s}: {best * 1e3:7.3f} ms per render (median {np.median(ts[1:]) * 1e3:7.3f}), {it} iterations, {rays / best / 1e6:8.1f} Mrays/s, err {k['device_error']}", flush=True)
            del g


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "parity"
    rc = parity() if what == "parity" else timing()
    sys.exit(1 if rc else 0)
