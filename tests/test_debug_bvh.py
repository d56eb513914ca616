"""extend_debug_BVH (kernel.cu:300-328): the reference's compile-time BVH_DEBUG picture -- every pixel coloured by the
number of traversal steps CachedBVH::intersect_debug (bvh.h:164-209) takes for its ray.  TYR_FLAG_DEBUG_BVH.

Pinned to the reference itself: tests/golden/ref_traverse_*.npz carries intersect_debug's own step counts for its ray sets
(compiled from the reference's bvh.h, oracle/ref_harness.cpp); the picture the oracle (CPU) and the HIP kernel (GPU) paint
for those rays must be the one that follows from them."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, built_scene

FLAG = 32


def expected_pixels(traversals):
    green = np.minimum(((np.float32(0.0002) * traversals.astype(np.float32)) * np.float32(255.99)).astype(np.int32), 255).astype(np.float32)
    costly = traversals >= 70
    px = np.zeros((len(traversals), 4), dtype=np.float32)
    px[:, 1] = np.where(costly, 0.0, green)
    px[:, 0] = np.where(costly, green, 0.0)
    px[:, 3] = 1.0
    return px


def fixture(name):
    from tyrant_amd import scenes

    z = np.load(os.path.join(GOLDEN, f"ref_traverse_{name}.npz"))
    nodes = np.ascontiguousarray(z["nodes"]).view(scenes.NODE_DTYPE).reshape(-1)
    prims = np.ascontiguousarray(z["prims"]).view(scenes.TRIANGLE_DTYPE).reshape(-1)
    keep = z["distance_in"] >= np.float32(1e20)  # the debug path starts every ray at VERY_FAR (kernel.cu:145)
    n = int(keep.sum())
    rays = np.zeros(n, dtype=scenes.RAY_DTYPE)
    rays["origin"], rays["direction"], rays["direct"] = z["origin"][keep], z["direction"][keep], 1.0
    rays["index"] = np.arange(n, dtype=np.int32)
    return nodes, prims, rays, z["traversals"][keep], n


def paint(r, nodes, prims, rays, n, stage):
    r.upload(nodes, prims)
    r.stage("begin")
    r.import_work_queue(rays, n)
    r.set_budget(0)
    r.stage("primary")  # budget 0: no new rays, n_live = n
    r.stage(stage)
    return r.blit_buffer()[:n]


@pytest.mark.parametrize("name", ["cornell36", "soup2k", "mesh32"])
def test_oracle_heat_map_follows_the_reference_step_counts(orc, name):
    nodes, prims, rays, trav, n = fixture(name)
    o = orc.Oracle(64, (n + 63) // 64, n, flags=FLAG)
    assert np.array_equal(paint(o, nodes, prims, rays, n, "extend_debug"), expected_pixels(trav))
    assert trav.max() >= 70 or name == "cornell36"  # the red branch is exercised on the two larger trees


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell36", "soup2k", "mesh32"])
def test_device_heat_map_follows_the_reference_step_counts(hip, name):
    nodes, prims, rays, trav, n = fixture(name)
    g = hip.Renderer(64, (n + 63) // 64, n, flags=FLAG)
    got = paint(g, nodes, prims, rays, n, "extend")
    assert g.counters()["device_error"] == 0
    assert np.array_equal(got, expected_pixels(trav))


@pytest.mark.gpu
def test_debug_render_matches_oracle(orc, hip):
    """launch_kernels under BVH_DEBUG = primary + extend_debug_BVH only (kernel.cu:720-722): nothing survives, the cursor
    walks on, the picture equals the oracle's exactly (plain stores, one ray per pixel)"""
    sc, nodes, prims = built_scene("mesh128")
    W, H = 128, 72
    o = orc.Oracle(W, H, W * H, flags=FLAG)
    g = hip.Renderer(W, H, W * H, flags=FLAG)
    o.load_scene(sc, nodes, prims), g.load_scene(sc, nodes, prims)
    for _ in range(2):
        o.launch_kernels(), g.launch_kernels()
    ko, kg = o.counters(), g.counters()
    assert kg["device_error"] == 0 and kg["primary_ray_cnt"] == ko["primary_ray_cnt"] == 0 and kg["start_position"] == ko["start_position"] and kg["frame"] == ko["frame"] == 3
    bo, bg = o.blit_buffer(), g.blit_buffer()
    assert np.array_equal(bo, bg) and (bo[:, 1] > 0).any() and np.all(bo[:, 3] == 1)
    assert o.render(1) == g.render(1) == 1  # a budgeted render under the flag: one pass, nothing to drain
    assert np.array_equal(o.blit_buffer(), g.blit_buffer())
