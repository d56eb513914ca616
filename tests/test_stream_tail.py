"""The streamed tail (TYR_TUNE_STREAM_TAIL: one traversal kernel across the iterations behind a render's last top-up, shade
resident beside it) on placements chosen to stress its hand-off invariants -- random fuzzing alone did not
protect round 3's queue design (the segment overflow needed an adversarial placement to show):

  * fill[chunk] / `closed`: queues that are NOT multiples of 64 in any segment (every segment ends in a partial chunk that only
    becomes ready when its iteration closes), down to queues smaller than one chunk per segment;
  * done[tile]: tiles with a single valid record; class 0 empty while class 1 is not (the tree behind the camera: no ray ever
    enters it, the traversal kernel only ever sees shadow rays) and the reverse (a room that every ray stays inside);
  * the end of the tail: renders whose last iterations hold shadow rays only, and tails of different lengths (rays of
    different ages when the budget runs out: queue < budget, several top-ups before the tail starts);
  * traversal grids of 1, 2, 3 blocks per CU beside the shade block (starved and crowded chunk tickets).
Every case: iteration count, every counter, the last iteration's shadow queue and the radiance against the oracle."""
import dataclasses

import numpy as np
import pytest

from conftest import bits, built_scene
from test_gpu_parity import assert_accum_close

pytestmark = pytest.mark.gpu

FIELDS = ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible", "start_position", "frame", "n_live", "shadow_ray_cnt", "primary_ray_cnt")


def check(orc, hip, sc, nodes, prims, W, H, N, spp, knobs, what):
    flags = (1 if sc.triangle_materials else 0) | (8 if sc.light_list else 0) | (16 if sc.triangle_colors else 0)
    o = orc.Oracle(W, H, N, flags=flags)
    g = hip.Renderer(W, H, N, flags=flags)
    o.load_scene(sc, nodes, prims), g.load_scene(sc, nodes, prims)
    g.set_tuning(merge_trace=1, run_ahead=0, stream_tail=1, **knobs)
    for rnd in range(2):  # twice: the second render starts from the state the first tail left behind
        io, ig = o.render(spp), g.render(spp)
        ko, kg = o.counters(), g.counters()
        assert kg["device_error"] == 0, (what, kg)
        assert io == ig, (what, rnd, io, ig)
        for f in FIELDS:
            assert ko[f] == kg[f], (what, rnd, f, ko[f], kg[f])
        if knobs.get("resolve_shadows") == 0:  # every shadow ray queued: the last iteration's shadow queue, record for record
            nh = ko["shadow_ray_cnt"]
            so, sg = o.shadow_queue(nh), g.shadow_queue(nh)
            for f in ("origin", "direction", "color", "closestDistance"):
                assert np.array_equal(bits(so[f]), bits(sg[f])), (what, rnd, "shadow queue", f)
        assert_accum_close(o.blit_buffer(), g.blit_buffer(), f"{what}, render {rnd}")
    g.close()


@pytest.mark.parametrize("name,W,H,N,spp", [
    ("cornell_soup2k", 61, 37, 61 * 37 * 2, 2),   # the whole budget in flight: the tail starts at iteration 2; 4514 slots: eight ragged segments
    ("cornell_soup2k", 61, 37, 999, 3),           # queue < budget: seven top-ups, the tail's rays are of every age
    ("mesh128", 48, 30, 65, 2),                   # barely more than ONE chunk in the whole queue
    ("mesh128", 40, 24, 40 * 24, 1),              # 960 slots = 15 chunks over 8 segments x 2 classes
    ("glass_dof48", 96, 54, 96 * 54 * 2, 2),      # refraction + thin lens: long specular chains, every path to max bounces
    ("cornell_area_light", 80, 48, 80 * 48 * 3, 3),
])
def test_ragged_queues_and_tails_of_every_length(orc, hip, name, W, H, N, spp):
    sc, nodes, prims = built_scene(name)
    check(orc, hip, sc, nodes, prims, W, H, N, spp, {}, f"{name} {W}x{H} N={N} spp={spp}")


@pytest.mark.parametrize("trace_per_cu", [1, 2, 3])
def test_starved_and_crowded_chunk_tickets(orc, hip, trace_per_cu):
    sc, nodes, prims = built_scene("cornell_soup10k")
    check(orc, hip, sc, nodes, prims, 160, 90, 160 * 90 * 4, 4, dict(stream_trace_per_cu=trace_per_cu, resolve_shadows=trace_per_cu & 1), f"soup10k, {trace_per_cu} traversal blocks per CU")


def test_no_ray_ever_enters_the_tree(orc, hip):
    """the triangles lie behind the camera: class 0 stays empty in every iteration, the traversal kernel is handed shadow rays
    only (they miss the root box and retire at the refill) and must still find the end of the render"""
    from tyrant_amd import scenes

    sc, _, _ = built_scene("tyrant_default")
    tris = sc.triangles.copy()
    tris["vert"] = tris["vert"] + np.array([0.0, -5000.0, 0.0], dtype=np.float32)  # far behind the camera (it looks along +y from y = -250)
    moved = dataclasses.replace(sc, triangles=tris, name="tyrant_default_tree_behind_camera")
    nodes, prims = orc.bvh_build(moved.triangles, scenes.triangle_bboxes(moved.triangles))
    check(orc, hip, moved, nodes, prims, 96, 64, 96 * 64 * 2, 2, {}, moved.name)


def test_every_ray_stays_in_the_tree(orc, hip):
    """a closed Cornell room seen from inside, no sky: class 1 only ever holds the rays that hit the light sphere"""
    sc, nodes, prims = built_scene("cornell36")
    check(orc, hip, sc, nodes, prims, 72, 72, 72 * 72 * 3, 3, {}, "cornell36 from inside")


def test_a_render_too_short_for_the_tail_uses_launches(orc, hip):
    """max_iterations below what a tail may need: the streamed form is not entered (it cannot stop half-way), results as ever"""
    sc, nodes, prims = built_scene("cornell_soup2k")
    o = orc.Oracle(80, 48, 80 * 48 * 2)
    g = hip.Renderer(80, 48, 80 * 48 * 2)
    o.load_scene(sc, nodes, prims), g.load_scene(sc, nodes, prims)
    g.set_tuning(merge_trace=1, run_ahead=0, stream_tail=1)
    assert o.render(2, 3) == g.render(2, 3) == 3
    ko, kg = o.counters(), g.counters()
    for f in FIELDS:
        assert ko[f] == kg[f], f
    assert_accum_close(o.blit_buffer(), g.blit_buffer(), "render cut at three iterations")
    # ... and the rest of it, now long enough: the tail is streamed from a queue an ordinary iteration left
    assert o.render(0) == g.render(0)
    ko, kg = o.counters(), g.counters()
    assert kg["device_error"] == 0
    for f in FIELDS:
        assert ko[f] == kg[f], f
    assert_accum_close(o.blit_buffer(), g.blit_buffer(), "the rest of the render, streamed")


def test_more_traversal_blocks_than_leave_room_for_shade_are_refused(hip):
    """five traversal blocks of 31.7 KB of LDS per CU leave none for the 27.8 KB shade block that must be resident beside them:
    the traversal would poll for chunks nobody can publish until its bounded waits expire.  The knob stops at four."""
    sc, nodes, prims = built_scene("cornell36")
    g = hip.Renderer(32, 32, 1024)
    g.load_scene(sc, nodes, prims)
    g.set_tuning(stream_trace_per_cu=4)
    with pytest.raises(hip.TyrError):
        g.set_tuning(stream_trace_per_cu=5)
    g.close()
