"""Parity of the HIP path (libtyrant_hip.so, called through its C ABI) with the oracle on the
same seeded inputs.  Needs an MI355X: `pytest -m gpu`.

The bar (DESIGN.md "Numeric contract"): every queue record -- origins, directions,
throughputs, hit distances and identifiers, survivor order, shadow rays -- is BIT-EXACT,
because both sides evaluate the same IEEE operation sequence and compaction is stable.
The accumulation buffer is compared to 1e-5 relative: its float atomics add in a
different order (north_star asks for 1e-4)."""
import numpy as np
import pytest

from conftest import bits, built_scene

pytestmark = pytest.mark.gpu

LIVE_FIELDS = ("origin", "direction", "direct")


def pair(orc, hip, name, W, H, N, flags=0, rank=0, nranks=1):
    sc, nodes, prims = built_scene(name)
    if sc.triangle_materials:
        flags |= 1
    if sc.light_list:
        flags |= 8
    if sc.triangle_colors:
        flags |= 16
    o = orc.Oracle(W, H, N, rank=rank, nranks=nranks, flags=flags & 25)
    o.load_scene(sc, nodes, prims)
    g = hip.Renderer(W, H, N, rank=rank, nranks=nranks, flags=flags)
    g.load_scene(sc, nodes, prims)
    return o, g


def hash_spread(i: np.ndarray, seed: int) -> np.ndarray:
    from tyrant_amd import scenes

    return scenes.hash_unit(np.asarray(i), seed)


def assert_state_equal(qo, qg, what):
    for f in LIVE_FIELDS:
        assert np.array_equal(bits(qo[f]), bits(qg[f])), f"{what}: {f} differs in {np.count_nonzero(np.any(bits(qo[f]) != bits(qg[f]), axis=-1))} records"
    for f in ("index", "bounces", "lastSpecular"):
        assert np.array_equal(qo[f], qg[f]), f"{what}: {f}"


def assert_accum_close(bo, bg, what):
    assert np.array_equal(bo[:, 3], bg[:, 3]), f"{what}: completed-path counts differ"
    assert np.allclose(bg[:, :3], bo[:, :3], rtol=1e-5, atol=1e-6), f"{what}: max rel err {np.max(np.abs(bg[:, :3] - bo[:, :3]) / np.maximum(np.abs(bo[:, :3]), 1e-3))}"


@pytest.mark.parametrize("name,W,H,N,flags", [("cornell36", 64, 64, 6000, 0), ("tyrant_default", 96, 64, 5000, 0), ("cornell_soup2k", 80, 48, 4096, 0), ("mesh32", 64, 64, 4096, 0), ("glass_dof48", 96, 54, 4096, 0),
                                              ("cornell_area_light", 96, 64, 5000, 0), ("cornell_colored", 96, 64, 5000, 0), ("cornell_soup2k", 80, 48, 4096, 4), ("glass_dof48", 96, 54, 4096, 4)])
def test_stage_by_stage_bit_parity(orc, hip, name, W, H, N, flags):
    """every kernel of every iteration, fed by its predecessors on each side, matches the oracle bit for bit
    (flags = 4: the counting build of the traversal on pair nodes, TYR_FLAG_COUNT_VISITS)"""
    o, g = pair(orc, hip, name, W, H, N, flags=flags)
    for it in range(5):
        tag = f"{name} iteration {it}"
        o.stage("begin"), g.stage("begin")
        o.stage("primary"), g.stage("primary")
        ko, kg = o.counters(), g.counters()
        assert kg["device_error"] == 0
        for f in ("n_live", "start_position", "total_primary_rays", "total_extend_rays"):
            assert ko[f] == kg[f], (tag, f)
        n = ko["n_live"]
        assert_state_equal(o.ray_queue(0, n), g.ray_queue(0, n), tag + " after primary")
        if it > 0 and not (flags & 4):
            assert g.queue_rank_check(0) == (n, 0), tag + " rank tables: survivors in front, fresh primary rays behind"

        o.stage("extend"), g.stage("extend")
        qo, qg = o.ray_queue(0, n), g.ray_queue(0, n)
        assert np.array_equal(bits(qo["distance"]), bits(qg["distance"])), tag + " extend distance"
        hit = qo["distance"] < 1e20
        assert np.array_equal(qo["identifier"][hit], qg["identifier"][hit]), tag + " extend identifier"
        assert np.array_equal(qo["geometry_type"][hit], qg["geometry_type"][hit]), tag + " extend geometry_type"

        o.stage("shade"), g.stage("shade")
        ko, kg = o.counters(), g.counters()
        assert kg["device_error"] == 0
        assert ko["primary_ray_cnt"] == kg["primary_ray_cnt"] and ko["shadow_ray_cnt"] == kg["shadow_ray_cnt"], tag
        ns, nh = ko["primary_ray_cnt"], ko["shadow_ray_cnt"]
        assert_state_equal(o.ray_queue(1, ns), g.ray_queue(1, ns), tag + " survivors")
        # the order above comes from the export's host-side sort by virtual slot; the kernels resolve the same keys through the
        # scan's rank tables: every survivor's table rank must be its place in that order
        assert g.queue_rank_check(1) == (ns, 0), tag + " rank tables of the scan"
        so, sg = o.shadow_queue(nh), g.shadow_queue(nh)
        for f in ("origin", "direction", "color", "closestDistance"):
            assert np.array_equal(bits(so[f]), bits(sg[f])), f"{tag} shadow {f}"
        assert np.array_equal(so["buffer_index"], sg["buffer_index"]), tag

        o.stage("connect"), g.stage("connect")
        assert o.counters()["n_shadow_visible"] == g.counters()["n_shadow_visible"], tag
        assert_accum_close(o.blit_buffer(), g.blit_buffer(), tag)
        o.stage("end"), g.stage("end")


@pytest.mark.parametrize("name,W,H,N,spp", [("cornell36", 128, 128, 16384, 4), ("tyrant_default", 160, 96, 10000, 4), ("cornell_soup10k", 128, 72, 8192, 3), ("mesh128", 96, 96, 8192, 2), ("glass_dof48", 128, 72, 8192, 3), ("cornell_area_light", 128, 96, 8192, 4), ("cornell_colored", 128, 96, 8192, 4)])
def test_render_matches_oracle(orc, hip, name, W, H, N, spp):
    """launch_kernels loop with a primary budget: same iteration count, same ray totals, radiance within 1e-5 rel"""
    o, g = pair(orc, hip, name, W, H, N)
    it_o = o.render(spp)
    it_g = g.render(spp)
    assert it_o == it_g
    ko, kg = o.counters(), g.counters()
    assert kg["device_error"] == 0
    for f in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible", "start_position", "frame"):
        assert ko[f] == kg[f], f
    bo, bg = o.blit_buffer(), g.blit_buffer()
    assert np.all(bg[:, 3] == spp)
    assert_accum_close(bo, bg, name)


@pytest.mark.parametrize("second_wall", [False, True])
def test_every_survivor_lands_in_one_queue_segment(orc, hip, second_wall):
    """The worst case the queue segments are sized for (hip/kernels.hpp "Queues").  A shade tile appends to
    segment (tile / 2) % 8, so records whose index j has (j / 512) % 8 == 0 all send their survivors and shadow rays to
    segment 0: exactly those rays hit a wall here, the other seven eighths leave for the sky.  The top-up primaries that
    follow spread evenly over the segments on top of that: segment 0 then holds N/8 + 7N/64 records, more than
    1.5 x its share.  No overflow, and the oracle's counts, queues and radiance, iteration after iteration.
    One wall: the survivors leave the tree's box for good (class 1, the traversal kernel never sees them).  A second wall
    behind the rays, facing the first: they bounce between the two (class 0: the lopsided segments, holes included, go
    through the traversal kernel), and so do the top-up primaries, whose camera looks at the second wall."""
    from tyrant_amd import scenes

    W, H = 512, 256
    N = W * H
    j = np.arange(N)
    stay = (j // 512) % 8 == 0
    rays = np.zeros(N, dtype=scenes.RAY_DTYPE)
    rays["origin"] = np.stack([hash_spread(j, 1) * 400.0 - 200.0, np.full(N, -100.0), hash_spread(j, 2) * 400.0 - 200.0], axis=1).astype(np.float32)
    rays["direction"] = np.where(stay[:, None], np.float32([0, 1, 0]), np.float32([0, 0, 1]) if second_wall else np.float32([0, -1, 0]))
    rays["direct"] = 1.0
    rays["distance"] = 1e20
    rays["index"] = j
    wall = scenes._quad((-5000.0, 0.0, -5000.0), (5000.0, 0.0, -5000.0), (5000.0, 0.0, 5000.0), (-5000.0, 0.0, 5000.0), (0, -1, 0))
    if second_wall:
        wall = np.concatenate([wall, scenes._quad((-5000.0, -200.0, -5000.0), (-5000.0, -200.0, 5000.0), (5000.0, -200.0, 5000.0), (5000.0, -200.0, -5000.0), (0, 1, 0))])
    spheres = scenes.cornell_spheres()
    spheres[4] = spheres[0]  # no ground, no light
    spheres[6] = spheres[1]
    cam = scenes.Camera(position=(0.0, -190.0, 50.0), direction=(0.0, -1.0, 0.0), up=(0.0, 0.0, 1.0))  # looks away from the wall
    sc = scenes.SceneData("one_segment", wall, spheres, cam)
    nodes, prims = orc.bvh_build(sc.triangles, scenes.triangle_bboxes(sc.triangles))
    o, g = orc.Oracle(W, H, N), hip.Renderer(W, H, N)
    for r in (o, g):
        r.load_scene(sc, nodes, prims)
        r.set_camera(cam)
        r.stage("begin")
        r.import_work_queue(rays, N)
        r.set_budget(0)
        r.stage("primary")
    for it in range(4):
        if it:
            for r in (o, g):
                r.set_budget(N * 8)
                r.stage("begin"), r.stage("primary")
        for r in (o, g):
            r.stage("extend"), r.stage("shade")
        ko, kg = o.counters(), g.counters()
        assert kg["device_error"] == 0, f"iteration {it}"
        for f in ("n_live", "primary_ray_cnt", "shadow_ray_cnt", "total_primary_rays", "total_extend_rays"):
            assert ko[f] == kg[f], (it, f)
        if it == 0:
            assert ko["primary_ray_cnt"] > N // 8 - N // 256, "every ray aimed at the wall is meant to survive its first bounce"
        ns, nh = ko["primary_ray_cnt"], ko["shadow_ray_cnt"]
        assert_state_equal(o.ray_queue(1, ns), g.ray_queue(1, ns), f"iteration {it} survivors")
        assert o.shadow_queue(nh).tobytes() == g.shadow_queue(nh).tobytes(), f"iteration {it} shadow rays"
        for r in (o, g):
            r.stage("connect"), r.stage("end")
    assert o.counters()["n_shadow_visible"] == g.counters()["n_shadow_visible"]
    assert_accum_close(o.blit_buffer(), g.blit_buffer(), "one segment")
    g.close()


def test_reference_traversal_fixture_through_the_abi(orc, hip):
    """the committed answers of the reference's own bvh.h code (tests/golden/ref_traverse_*.npz), reproduced
    by the HIP extend kernel: rays are imported as queue records, extended, exported"""
    import os

    from conftest import GOLDEN
    from tyrant_amd import scenes

    for name in ("cornell36", "soup2k", "mesh32"):
        z = np.load(os.path.join(GOLDEN, f"ref_traverse_{name}.npz"))
        nodes = np.ascontiguousarray(z["nodes"]).view(scenes.NODE_DTYPE).reshape(-1)
        prims = np.ascontiguousarray(z["prims"]).view(scenes.TRIANGLE_DTYPE).reshape(-1)
        n = z["origin"].shape[0]
        g = hip.Renderer(64, 64, n)
        g.upload(nodes, prims)
        # park every sphere where no ray can reach it: only the BVH answers
        s = scenes.cornell_spheres()
        s["position"] = np.array([0.0, 1e6, -1e6], dtype=np.float32)
        s["radius"] = 1.0
        g.set_spheres(s)
        rays = np.zeros(n, dtype=scenes.RAY_DTYPE)
        rays["origin"], rays["direction"], rays["direct"] = z["origin"], z["direction"], 1.0
        keep = z["distance_in"] >= np.float32(1e20)  # the fixture also has pre-shortened rays; extend always starts from VERY_FAR
        g.stage("begin")
        g.import_work_queue(rays, n)
        g.set_budget(0)
        g.stage("primary")  # budget 0: no new rays, n_live = n
        assert g.counters()["n_live"] == n
        g.stage("extend")
        q = g.ray_queue(0, n)
        hit = z["hit"].astype(bool) & keep
        assert np.array_equal(q["distance"][keep] < 1e20, z["hit"][keep].astype(bool)), name
        assert np.array_equal(bits(q["distance"][hit]), bits(z["distance"][hit])), name
        assert np.array_equal(q["identifier"][hit], z["identifier"][hit]), name


@pytest.mark.parametrize("flags", [0, 4])
def test_axis_aligned_rays_and_long_leaves(orc, hip, flags):
    """rays with zero direction components (1/d infinite: the generic box test, not the packed one) through a scene
    whose leaves are longer than the device layout's inline limit (the quad layout's chains of -inf..+inf boxes) and
    through an ordinary mesh: extend answers bit for bit, for mixed waves (regular and axis-aligned rays side by side);
    flags = 4 is the counting build (pair nodes, TYR_FLAG_COUNT_VISITS)"""
    from tyrant_amd import scenes

    rng = np.random.default_rng(5)
    # 40 identical-centroid triangles in one leaf + a second over-long leaf elsewhere, + the Cornell box around them
    def stack(x0, n):
        t = scenes.make_triangles(np.tile([x0 - 30, 0, 10], (n, 1)), np.tile([x0 + 30, 0, 10], (n, 1)), np.tile([x0, 0, 70], (n, 1)))
        return t
    tris = np.concatenate([scenes.cornell_box().triangles, stack(-10.0, 40), stack(15.0, 70)])
    nodes, prims = orc.bvh_build(tris, scenes.triangle_bboxes(tris))
    assert nodes["primitiveCount"].max() >= 40
    n = 8192
    rays = np.zeros(n, dtype=scenes.RAY_DTYPE)
    rays["origin"] = np.stack([rng.uniform(-45, 45, n), np.full(n, -100.0), rng.uniform(2, 98, n)], axis=1).astype(np.float32)
    axes = np.array([[0, 1, 0], [0, 1, 0], [0.6, 0.8, 0], [0, 0.8, 0.6], [0, 0.8, -0.6], [-0.6, 0.8, 0]], dtype=np.float32)
    d = axes[rng.integers(0, len(axes), n)]
    generic = rng.random(n) < 0.5  # the other half: ordinary directions, so that waves are mixed
    dd = rng.normal(size=(n, 3)).astype(np.float32)
    dd[:, 1] = np.abs(dd[:, 1]) + 0.5
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    rays["direction"] = np.where(generic[:, None], d, dd.astype(np.float32))
    rays["direct"] = 1.0
    rays["distance"] = 1e20
    s = scenes.cornell_spheres()
    s["position"] = np.array([0.0, 1e6, -1e6], dtype=np.float32)  # only the BVH answers
    s["radius"] = 1.0
    o = orc.Oracle(64, 64, n)
    g = hip.Renderer(64, 64, n, flags=flags)
    for r in (o, g):
        r.upload(nodes, prims)
        r.set_spheres(s)
        r.stage("begin")
        r.import_work_queue(rays, n)
        r.set_budget(0)
        r.stage("primary")
        r.stage("extend")
    assert g.counters()["device_error"] == 0
    qo, qg = o.ray_queue(0, n), g.ray_queue(0, n)
    assert np.array_equal(bits(qo["distance"]), bits(qg["distance"]))
    hit = qo["distance"] < 1e20
    assert hit.sum() > n // 2
    assert np.array_equal(qo["identifier"][hit], qg["identifier"][hit]) and np.array_equal(qo["geometry_type"][hit], qg["geometry_type"][hit])


def test_visit_counters_match_reference_counting_rule(orc, hip):
    """TYR_FLAG_COUNT_VISITS reproduces the oracle's nodes-visited / triangles-tested counts (bvh.h:164-209 rule)"""
    o, g = pair(orc, hip, "cornell_soup2k", 96, 64, 6144, flags=4)
    for _ in range(3):
        o.launch_kernels(), g.launch_kernels()
    ko, kg = o.counters(), g.counters()
    for f in ("nodes_extend", "tris_extend", "nodes_connect", "tris_connect"):
        assert ko[f] == kg[f] and ko[f] > 0, f


def test_edge_cases(orc, hip):
    from tyrant_amd import scenes

    # empty scene (Scene.cpp:49-52): spheres only
    sc = scenes.tyrant_default()
    o = orc.Oracle(48, 32, 1536)
    o.set_spheres(sc.spheres), o.set_camera(sc.camera)
    g = hip.Renderer(48, 32, 1536)
    g.upload(np.zeros(0, dtype=scenes.NODE_DTYPE), np.zeros(0, dtype=scenes.TRIANGLE_DTYPE))
    g.set_spheres(sc.spheres), g.set_camera(sc.camera)
    assert o.render(2) == g.render(2)
    assert_accum_close(o.blit_buffer(), g.blit_buffer(), "empty scene")
    # a single triangle: the root is a leaf
    t = scenes.make_triangles([[-30, 0, 10]], [[30, 0, 10]], [[0, 0, 70]])
    nodes, prims = orc.bvh_build(t, scenes.triangle_bboxes(t))
    o = orc.Oracle(32, 32, 1024)
    g = hip.Renderer(32, 32, 1024)
    for r in (o, g):
        r.upload(nodes, prims), r.set_spheres(scenes.cornell_spheres()), r.set_camera(scenes.CORNELL_CAMERA)
    assert o.render(2) == g.render(2)
    assert_accum_close(o.blit_buffer(), g.blit_buffer(), "single triangle")
    # a leaf longer than the device layout's inline limit (identical centroids, bvh.cpp:103-111): 40 stacked triangles
    t = scenes.make_triangles(np.tile([-30, 0, 10], (40, 1)) + np.arange(40)[:, None] * np.array([0, 0.01, 0]), np.tile([30, 0, 10], (40, 1)), np.tile([0, 0, 70], (40, 1)))
    t["e1"], t["e2"] = np.float32([60, 0, 0]), np.float32([30, 0, 60])
    t["vert"][:, 1] = 0.0  # identical boxes -> identical centroids -> one 40-primitive leaf
    nodes, prims = orc.bvh_build(t, scenes.triangle_bboxes(t))
    assert len(nodes) == 1 and nodes["primitiveCount"][0] == 40
    o = orc.Oracle(32, 32, 1024)
    g = hip.Renderer(32, 32, 1024)
    for r in (o, g):
        r.upload(nodes, prims), r.set_spheres(scenes.cornell_spheres()), r.set_camera(scenes.CORNELL_CAMERA)
    assert o.render(1) == g.render(1)
    assert_accum_close(o.blit_buffer(), g.blit_buffer(), "long leaf")
    # depth of field on (kernel.cu:286-293)
    sc, nodes, prims = built_scene("cornell36")
    cam = scenes.Camera(position=(0.0, -190.0, 50.0), direction=(0.0, 1.0, 0.0), focalDistance=60.0, lensRadius=4.0)
    o = orc.Oracle(64, 48, 3072)
    g = hip.Renderer(64, 48, 3072)
    for r in (o, g):
        r.load_scene(sc, nodes, prims), r.set_camera(cam)
    o.stage("begin"), g.stage("begin"), o.stage("primary"), g.stage("primary")
    assert_state_equal(o.ray_queue(0), g.ray_queue(0), "thin lens")
    # TYR_FLAG_LIGHT_LIST: needs TRIANGLE_MATERIALS; with no emissive triangle in the scene it changes nothing
    with pytest.raises(Exception):
        hip.Renderer(32, 32, 1024, flags=8)
    sc, nodes, prims = built_scene("mesh32")
    acc = []
    for flags in (1, 9):
        g = hip.Renderer(64, 48, 3072, flags=flags)
        g.load_scene(sc, nodes, prims)
        g.render(2)
        acc.append((g.counters(), g.blit_buffer()))
    assert acc[0][0]["total_shadow_rays"] == acc[1][0]["total_shadow_rays"] and acc[0][0]["total_extend_rays"] == acc[1][0]["total_extend_rays"]
    assert np.array_equal(acc[0][1][:, 3], acc[1][1][:, 3]) and np.allclose(acc[0][1], acc[1][1], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("N", [1, 63, 65, 257, 1000])
def test_tiny_and_ragged_queue_sizes(orc, hip, N):
    """queues far smaller than a wave / a tile / the frame: the top-up cursor walks the frame N rays at a time
    (kernel.cu:227-244), every persistent grid is one partial block, and the render still equals the oracle's"""
    W, H, spp = 16, 12, 2
    o, g = pair(orc, hip, "cornell36", W, H, N)
    for it in range(3):  # three iterations stage by stage (survivors + top-up in a queue of N slots)
        o.stage("begin"), g.stage("begin")
        o.stage("primary"), g.stage("primary")
        n = o.counters()["n_live"]
        assert n == g.counters()["n_live"] and n <= N
        assert_state_equal(o.ray_queue(0, n), g.ray_queue(0, n), f"N={N} iteration {it} after primary")
        for st in ("extend", "shade", "connect", "end"):
            o.stage(st), g.stage(st)
        ko, kg = o.counters(), g.counters()
        assert kg["device_error"] == 0
        assert ko["primary_ray_cnt"] == kg["primary_ray_cnt"] and ko["shadow_ray_cnt"] == kg["shadow_ray_cnt"]
    o, g = pair(orc, hip, "cornell36", W, H, N)
    assert o.render(spp) == g.render(spp)
    ko, kg = o.counters(), g.counters()
    assert kg["device_error"] == 0
    for f in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible", "start_position", "frame"):
        assert ko[f] == kg[f], (N, f)
    assert np.all(g.blit_buffer()[:, 3] == spp)
    assert_accum_close(o.blit_buffer(), g.blit_buffer(), f"N={N}")


def test_sharded_ranks_match_oracle(orc, hip):
    """pixel sharding: each rank's render equals the oracle run with the same (rank, nranks); the sum covers the frame"""
    W, H, R, spp = 64, 48, 4, 2
    total = np.zeros((W * H, 4), dtype=np.float32)
    for r in range(R):
        o, g = pair(orc, hip, "cornell36", W, H, 2048, rank=r, nranks=R)
        assert o.render(spp) == g.render(spp)
        assert_accum_close(o.blit_buffer(), g.blit_buffer(), f"rank {r}")
        total += g.blit_buffer()
    assert np.all(total[:, 3] == spp)


def test_full_size_properties(hip, orc):
    """BASELINE size (1080p, N = 2 Mi, config C2): size-independent properties instead of an oracle run --
    exact sample counts, ray conservation, finite non-negative radiance, idempotent reset, determinism of the queues"""
    W, H, N, spp = 1920, 1080, 2097152, 2
    sc, nodes, prims = built_scene("cornell_soup10k")
    g = hip.Renderer(W, H, N)
    g.load_scene(sc, nodes, prims)
    it = g.render(spp)
    k = g.counters()
    assert k["device_error"] == 0 and k["total_primary_rays"] == spp * W * H
    assert k["total_extend_rays"] == k["total_primary_rays"] + k["n_survive"] <= 6 * k["total_primary_rays"]
    b = g.blit_buffer()
    assert np.all(b[:, 3] == spp) and np.all(np.isfinite(b)) and np.all(b[:, :3] >= 0)
    # a second renderer reproduces the ray totals exactly (stable compaction => deterministic slots and seeds)
    g2 = hip.Renderer(W, H, N)
    g2.load_scene(sc, nodes, prims)
    assert g2.render(spp) == it
    k2 = g2.counters()
    for f in ("total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible"):
        assert k[f] == k2[f], f
    assert np.allclose(g2.blit_buffer(), b, rtol=1e-5, atol=1e-6)
    # the first wavefront of the oracle at full size is affordable: primary rays are bit-exact for all 2 Mi slots
    o = orc.Oracle(W, H, N)
    o.load_scene(sc, nodes, prims)
    g3 = hip.Renderer(W, H, N)
    g3.load_scene(sc, nodes, prims)
    o.stage("begin"), g3.stage("begin"), o.stage("primary"), g3.stage("primary")
    assert_state_equal(o.ray_queue(0), g3.ray_queue(0), "1080p primary rays")


def test_full_size_first_wavefront_on_the_million_triangle_scene(hip, orc):
    """BASELINE config C3 (996,882 triangles) at 1080p and the reference's queue size: the first wavefront -- 2 Mi
    primary rays through the 1.1 M-node tree, shaded -- is bit-exact against the oracle; the rest of the render is
    checked through ray conservation and exact sample counts"""
    W, H, N, spp = 1920, 1080, 2097152, 1
    sc, nodes, prims = built_scene("mesh706")
    o = orc.Oracle(W, H, N, flags=1)
    g = hip.Renderer(W, H, N, flags=1)
    o.load_scene(sc, nodes, prims), g.load_scene(sc, nodes, prims)
    for r in (o, g):
        r.stage("begin"), r.stage("primary"), r.stage("extend")
    qo, qg = o.ray_queue(0), g.ray_queue(0)
    assert np.array_equal(bits(qo["distance"]), bits(qg["distance"]))
    hit = qo["distance"] < 1e20
    assert hit.mean() > 0.25 and np.array_equal(qo["identifier"][hit], qg["identifier"][hit]) and np.array_equal(qo["geometry_type"][hit], qg["geometry_type"][hit])
    o.stage("shade"), g.stage("shade")
    ko, kg = o.counters(), g.counters()
    assert kg["device_error"] == 0 and ko["primary_ray_cnt"] == kg["primary_ray_cnt"] and ko["shadow_ray_cnt"] == kg["shadow_ray_cnt"]
    assert_state_equal(o.ray_queue(1, ko["primary_ray_cnt"]), g.ray_queue(1, kg["primary_ray_cnt"]), "C3 survivors of the first wavefront")
    assert o.shadow_queue(ko["shadow_ray_cnt"]).tobytes() == g.shadow_queue(kg["shadow_ray_cnt"]).tobytes()
    # a whole render at the GPU-sized queue: conservation and sample counts
    g2 = hip.Renderer(W, H, W * H * 2, flags=1)
    g2.load_scene(sc, nodes, prims)
    g2.render(2)
    k = g2.counters()
    assert k["device_error"] == 0 and k["total_primary_rays"] == 2 * W * H and k["total_extend_rays"] == k["total_primary_rays"] + k["n_survive"]
    b = g2.blit_buffer()
    assert np.all(b[:, 3] == 2) and np.all(np.isfinite(b)) and np.all(b[:, :3] >= 0)


def test_cpp_host_api_example(hip):
    """examples/render_main.cpp -- the reference's main.cpp loop written against include/tyrant/*.h -- runs
    (BVH build, upload, 24 frames of launch_kernels + the caller's swap, resolve, PPM)"""
    import os
    import subprocess

    from conftest import ROOT

    exe = os.path.join(ROOT, "tyrant_amd", "bin", "render_main")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tyrant_amd", "csrc"), "example"], check=True)
    out = os.path.join(ROOT, "gpurun_out", "render_main.ppm")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    p = subprocess.run([exe, "0", "24", out], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "frame counter 25" in p.stdout, p.stdout  # kernel.cu:667, 739: starts at 1, +1 per call
    assert os.path.getsize(out) > 640 * 360 * 3
    # ... and with a scene FILE, through the reference's own one-line call `scene.Load(path)` (Scene.h:9, main.cpp:113) on the
    # process-default context (tyrant::set_default_ctx)
    ply = os.path.join(ROOT, "gpurun_out", "render_main_scene.ply")
    n = 24
    xs = np.linspace(-80.0, 80.0, n + 1)
    verts = [(x, y, -18.0 + 6.0 * np.sin(0.09 * x) * np.cos(0.08 * y)) for y in xs for x in xs]
    faces = [(j * (n + 1) + i, j * (n + 1) + i + 1, (j + 1) * (n + 1) + i + 1, (j + 1) * (n + 1) + i) for j in range(n) for i in range(n)]
    with open(ply, "w") as f:
        f.write(f"ply\nformat ascii 1.0\nelement vertex {len(verts)}\nproperty float x\nproperty float y\nproperty float z\nelement face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n")
        f.write("".join(f"{x:.6f} {y:.6f} {z:.6f}\n" for x, y, z in verts))
        f.write("".join(f"4 {a} {b} {c} {d}\n" for a, b, c, d in faces))
    p = subprocess.run([exe, "0", "8", out, ply], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "Loading scene:" in p.stdout and "frame counter 9" in p.stdout, p.stdout
    # ... and with `BVH bvh(primitives, bboxes, SAH)` built ON THE DEVICE (tyrant::set_build_device): the same tree, so the same picture's ray totals
    q = subprocess.run([exe, "0", "8", out, ply], capture_output=True, text=True, timeout=300, env=dict(os.environ, TYR_BUILD_ON_DEVICE="1"))
    assert q.returncode == 0, q.stdout + q.stderr
    nodes_line = [l for l in p.stdout.splitlines() if "total nodes" in l]
    assert nodes_line and nodes_line == [l for l in q.stdout.splitlines() if "total nodes" in l], (p.stdout, q.stdout)


@pytest.mark.parametrize("knobs", [
    dict(static_share=0, ticket_chunk=64), dict(static_share=15, ticket_chunk=4096), dict(static_share=8, ticket_chunk=256, staged_nodes=0),
    dict(staged_nodes=1), dict(staged_nodes=21), dict(refill_min_idle=1, min_traversing=1), dict(refill_min_idle=64, min_traversing=64),
    dict(static_interleave=0, static_share=4), dict(wide_drain=0), dict(waves_per_simd=2),
])
def test_work_distribution_knobs_never_change_results(orc, hip, knobs):
    """every launch-shape knob of tyr_set_tuning -- how queue slots reach the waves, how much of the tree sits in LDS,
    when the descent loop is left -- at its extremes: queues stay bit-identical to the oracle's, on a scene whose
    tree is deeper than the LDS stack and on a queue that is not a multiple of anything"""
    for name, W, H, N in (("mesh128", 72, 40, 2999), ("cornell_soup2k", 50, 30, 777)):
        o, g = pair(orc, hip, name, W, H, N)
        g.set_tuning(**knobs)
        for it in range(3):
            o.launch_kernels(), g.launch_kernels()
            ko, kg = o.counters(), g.counters()
            assert kg["device_error"] == 0
            for f in ("primary_ray_cnt", "shadow_ray_cnt", "n_shadow_visible", "total_shadow_rays", "n_survive"):
                assert ko[f] == kg[f], (name, knobs, it, f)
            ns, nh = ko["primary_ray_cnt"], ko["shadow_ray_cnt"]
            assert_state_equal(o.ray_queue(0, ns), g.ray_queue(0, ns), f"{name} {knobs} iteration {it}")
            assert o.shadow_queue(nh).tobytes() == g.shadow_queue(nh).tobytes()
        assert_accum_close(o.blit_buffer(), g.blit_buffer(), f"{name} {knobs}")


@pytest.mark.parametrize("name,W,H,N,spp", [("cornell_soup2k", 160, 90, 9000, 5), ("mesh128", 128, 72, 20000, 3), ("cornell_area_light", 96, 64, 3000, 4)])
def test_render_with_and_without_merged_launches(orc, hip, name, W, H, N, spp):
    """tyr_render in launch_kernels' order (merge_trace = 0), with connect(i) inside the launch of extend(i + 1) and the
    rays that miss the tree shaded beside that launch (the default), one iteration ahead of the counts, and merged without
    the overlapped shade: same iteration count, same counters and the same radiance as the oracle; a render, a camera move (reset of the accumulation buffer while
    nothing may be in flight) and a second render back to back"""
    o, g1 = pair(orc, hip, name, W, H, N)
    _, g0 = pair(orc, hip, name, W, H, N)
    _, g2 = pair(orc, hip, name, W, H, N)
    _, g3 = pair(orc, hip, name, W, H, N)
    g0.set_tuning(merge_trace=0)
    g1.set_tuning(merge_trace=1, run_ahead=1, fold_prologue=0)  # every iteration opened by its own k_primary / k_pad_holes launches (round 5's default lets the previous iteration's last kernel do that once the budget is spent)
    g2.set_tuning(merge_trace=1, wide_block_min_items=0)  # (the default) connect(i) inside the launch of extend(i + 1): k_trace_flat, iteration i + 1 queued ahead of iteration i's counts -- and every launch as 768-thread blocks (six waves per SIMD: by default only launches of 3 Mi rays and more; the others here run the 256-thread form)
    g3.set_tuning(merge_trace=1, run_ahead=0, wide_drain=0)  # ... with the host waiting for every iteration's counts, and a wave's last rays left one to a lane
    _, g4 = pair(orc, hip, name, W, H, N)
    _, g5 = pair(orc, hip, name, W, H, N)
    g4.set_tuning(merge_trace=1, run_ahead=1, scan_in_trace=0, kernel_snapshot=0)  # one iteration ahead with round 5's two later steps off: the slot scan a launch of its own (its last block opens the next iteration), the counts copied behind shade and signalled by an event
    g3.set_tuning(resolve_shadows=0)  # (and one merged path that queues every shadow ray although shade has done their sphere halves)
    g5.set_tuning(merge_trace=1, fold_spheres=0, retire_sky=0)  # the sphere pre-pass kernels instead of shade doing their work for the rays it emits; camera rays that hit nothing queued for shade instead of finished by k_primary
    from tyrant_amd import scenes

    sc, _, _ = built_scene(name)
    moved = scenes.Camera(position=tuple(np.array(sc.camera.position) + np.array([3.0, 2.0, -1.0])), direction=sc.camera.direction, up=sc.camera.up)
    for cam in (sc.camera, moved):
        for r in (o, g0, g1, g2, g3, g4, g5):
            r.set_camera(cam)
        io, i0, i1, i2, i3, i4, i5 = o.render(spp), g0.render(spp), g1.render(spp), g2.render(spp), g3.render(spp), g4.render(spp), g5.render(spp)
        assert io == i0 == i1 == i2 == i3 == i4 == i5
        ko, k0, k1, k2, k3, k4, k5 = o.counters(), g0.counters(), g1.counters(), g2.counters(), g3.counters(), g4.counters(), g5.counters()
        assert k0["device_error"] == 0 and k1["device_error"] == 0 and k2["device_error"] == 0 and k3["device_error"] == 0 and k4["device_error"] == 0 and k5["device_error"] == 0
        # (n_live, shadow_ray_cnt: the iteration a run-ahead render queues for nothing must not show in the counters)
        for f in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible", "start_position", "frame", "n_live", "shadow_ray_cnt", "primary_ray_cnt"):
            assert ko[f] == k0[f] == k1[f] == k2[f] == k3[f] == k4[f] == k5[f], (name, f)
        assert_accum_close(o.blit_buffer(), g4.blit_buffer(), name + " run-ahead, scan launch + copied snapshot")
        assert_accum_close(o.blit_buffer(), g5.blit_buffer(), name + " sphere pre-pass kernels")
        # the last iteration's shadow queue, whatever path made it (the run-ahead path once exported the EMPTY look-ahead iteration's)
        nh = ko["shadow_ray_cnt"]
        so = o.shadow_queue(nh)
        for g in (g0, g3, g5):  # (the paths that queue every shadow ray: the others answer in place those that cannot reach a triangle, TYR_TUNE_RESOLVE_SHADOWS)
            sg = g.shadow_queue(nh)
            for f in ("origin", "direction", "color", "closestDistance"):
                assert np.array_equal(bits(so[f]), bits(sg[f])), f"{name}: shadow queue after the render, {f}"
        assert_accum_close(o.blit_buffer(), g3.blit_buffer(), name + " merged trace launches, no run-ahead")
        assert_accum_close(o.blit_buffer(), g0.blit_buffer(), name + " one stream")
        assert_accum_close(o.blit_buffer(), g1.blit_buffer(), name + " merged, always one iteration ahead")
        assert_accum_close(o.blit_buffer(), g2.blit_buffer(), name + " merged trace launches")
    # stage by stage right after a render with deferred / merged connects: nothing is left in flight or owed
    for st in ("begin", "primary", "extend", "shade", "connect", "end"):
        o.stage(st), g1.stage(st), g2.stage(st), g4.stage(st)
    assert_accum_close(o.blit_buffer(), g1.blit_buffer(), name + " staged iteration after the render")
    assert_accum_close(o.blit_buffer(), g2.blit_buffer(), name + " staged iteration after the merged render")
    assert_accum_close(o.blit_buffer(), g4.blit_buffer(), name + " staged iteration after a render with the scan launch + copied snapshot")
    # a render cut short by max_iterations still settles its last shadow rays before it returns
    o.reset_accum(), g2.reset_accum()
    assert o.render(spp, 2) == g2.render(spp, 2) == 2
    assert o.counters()["n_shadow_visible"] == g2.counters()["n_shadow_visible"]
    assert_accum_close(o.blit_buffer(), g2.blit_buffer(), name + " two iterations of a merged render")
    # ... and the queues are where the reference's loop would have left them: the rest of the render, stage by stage
    for it in range(2):
        for st in ("begin", "primary"):
            o.stage(st), g2.stage(st)
        n = o.counters()["n_live"]
        assert n == g2.counters()["n_live"]
        assert_state_equal(o.ray_queue(0, n), g2.ray_queue(0, n), f"{name}: iteration {2 + it} after a render cut at two")
        for st in ("extend", "shade", "connect", "end"):
            o.stage(st), g2.stage(st)


def test_bench_two_ranks_on_one_gpu(hip):
    """bench.py's N > 1 path under an external launcher (the driver's form: torch.distributed.run sets RANK / WORLD_SIZE),
    weak scaling: two ranks (gloo, both on device 0), rows dealt y % 2 == rank, 2 spp per GPU = 4 in total, sum-reduce onto
    rank 0 -- bench.py itself asserts that every pixel of the combined frame holds exactly spp_total completed paths.
    (The self-launching form and strong scaling: tests/test_bench_contract.py.)"""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    import socket

    with socket.socket() as sock:  # a port nobody holds right now
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo", "--workload", "c1", "--width", "320", "--height", "180", "--queue", "32768", "--spp", "2", "--scaling", "weak", "--combine", "reduce", "--no-reference-queue"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["spp_total"] == 4 and d["scaling"] == "weak" and d["value"] > 0


def test_shadow_queue_after_a_render_that_ends_early(orc, hip):
    """ADVICE (round 5): a run-ahead render whose rays all die before kMaxBounces ends on an EMPTY iteration that was queued ahead -- and,
    the budget being spent, folded (its shade launch would open a successor).  That fold must not happen: it zeroes the counters of the
    last REAL iteration's shadow queue, which tyr_shadow_export hands out after the render.  A nearly black ground ends most paths by
    Russian roulette at once: six 16 x 16 renders in a row end after 3 to 6 iterations, some with shadow rays left in the last one.
    Default tuning except resolve_shadows = 0 (so that every shadow ray is queued and the export is the reference's queue)."""
    import dataclasses

    from tyrant_amd import scenes

    sc0 = scenes.tyrant_default()
    sp = sc0.spheres.copy()
    sp["color"][4] = (0.06, 0.05, 0.04)
    sp["color"][0] = (0.1, 0.1, 0.1)
    sc = dataclasses.replace(sc0, spheres=sp)
    nodes, prims = hip.bvh_build(sc.triangles)
    W = H = 16
    N = 256
    o, g = orc.Oracle(W, H, N), hip.Renderer(W, H, N)
    o.load_scene(sc, nodes, prims), g.load_scene(sc, nodes, prims)
    g.set_tuning(resolve_shadows=0)
    early = with_shadows = 0
    for rep in range(6):
        io, ig = o.render(1), g.render(1)
        ko, kg = o.counters(), g.counters()
        assert kg["device_error"] == 0 and io == ig, (rep, io, ig)
        for f in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible", "start_position", "frame", "n_live", "shadow_ray_cnt", "primary_ray_cnt"):
            assert ko[f] == kg[f], (rep, f, ko[f], kg[f])
        nh = ko["shadow_ray_cnt"]
        so, sg = o.shadow_queue(nh), g.shadow_queue(nh)
        for f in ("origin", "direction", "color", "closestDistance"):
            assert np.array_equal(bits(so[f]), bits(sg[f])), f"render {rep} ({io} iterations): shadow queue after the render, {f}"
        early += io < 6
        with_shadows += (io < 6 and nh > 0)
    assert early >= 2 and with_shadows >= 1  # (the oracle's run: renders of 5, 5, 3 (2 shadow rays), 6, 6, 4 iterations)
    # ... and the ctx is where the reference's loop would have left it: one more iteration, stage by stage
    for st in ("begin", "primary", "extend", "shade", "connect", "end"):
        o.stage(st), g.stage(st)
    assert_accum_close(o.blit_buffer(), g.blit_buffer(), "staged iteration after renders that ended early")


def test_triangle_colors_defaults_and_errors(orc, hip):
    """TYR_FLAG_TRIANGLE_COLORS (per-triangle colour / emission, the reference's commented-out Scene.cpp:44): needs
    TRIANGLE_MATERIALS; with the default palette (white, (3,3,3)) it is the run without the flag, bit for bit in the
    queues; a palette set on a ctx without the flag is refused"""
    with pytest.raises(Exception):
        hip.Renderer(32, 32, 1024, flags=16)
    sc, nodes, prims = built_scene("cornell_area_light")  # emission (4, 3.5, 3): set the palette's entry 0 to the same
    em = np.full((256, 3), 3.0, dtype=np.float32)
    em[0] = sc.triangle_emission
    runs = []
    for flags in (9, 25):
        g = hip.Renderer(96, 64, 4000, flags=flags)
        g.load_scene(sc, nodes, prims)
        if flags & 16:
            g.set_triangle_palette(np.ones((256, 3), dtype=np.float32), em)
        for _ in range(3):
            g.launch_kernels()
        k = g.counters()
        runs.append((k, g.ray_queue(0, k["primary_ray_cnt"]), g.shadow_queue(k["shadow_ray_cnt"]).tobytes(), g.blit_buffer()))
        if not flags & 16:
            with pytest.raises(hip.TyrError):
                g.set_triangle_palette(np.ones((256, 3), dtype=np.float32))
    assert_state_equal(runs[0][1], runs[1][1], "survivors with and without the palette")  # (distance / identifier of a survivor are stale until extend: quirk 18)
    assert runs[0][2] == runs[1][2]
    assert np.allclose(runs[0][3], runs[1][3], rtol=1e-5, atol=1e-6)
