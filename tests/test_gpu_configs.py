"""BASELINE.json's configurations C4 and C5 and the remaining kernel-level parity holes, on an MI355X (`pytest -m gpu`),
through the C ABI:

  C4  the 1 M-triangle scene sharded over R = 2 / 4 / 8 ranks at 1080p: every rank's first wavefront (primary, extend,
      shade) bit-exact against the oracle run with the same (rank, nranks) -- EVERY rank at R = 2, 4 and 8 --; a full render per rank
      checked through ray conservation and exact sample counts on the rows the rank owns (and zeros elsewhere).
  C5  the 10 M-triangle glass + depth-of-field + sun scene at 3840x2160, BVH built by the PRODUCT's builder
      (tyr_bvh_build, 16 threads): first 2 Mi-slot wavefront against the oracle, the device layout inside its encoding
      limits (25-bit quad index, 26-bit primitive offset, 64-entry stack), a whole 1-spp render at 4K against orc_render.
  a15 blit_onto_framebuffer (kernel.cu:648-662): tyr_resolve bit-exact against orc_resolve, the 0/0 pixel included.
  a11 the reference's any-hit answers (tests/golden/ref_traverse_*.npz, CachedBVH::intersectSimple) through the HIP
      connect kernel.
  f1  tyr_bvh_build on the GPU box's host: C3's tree, bytes against the oracle's builder.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits, built_scene
from test_gpu_parity import assert_state_equal

pytestmark = pytest.mark.gpu

W1080, H1080, N2M = 1920, 1080, 2097152


def first_wavefront_bit_exact(o, g, tag, min_hit=0.0):
    for r in (o, g):
        r.stage("begin"), r.stage("primary")
    ko, kg = o.counters(), g.counters()
    assert kg["device_error"] == 0 and ko["n_live"] == kg["n_live"] and ko["start_position"] == kg["start_position"], tag
    n = ko["n_live"]
    assert_state_equal(o.ray_queue(0, n), g.ray_queue(0, n), tag + " primary rays")
    o.stage("extend"), g.stage("extend")
    qo, qg = o.ray_queue(0, n), g.ray_queue(0, n)
    assert np.array_equal(bits(qo["distance"]), bits(qg["distance"])), tag + " extend distance"
    hit = qo["distance"] < 1e20
    assert hit.mean() >= min_hit, (tag, hit.mean())
    assert np.array_equal(qo["identifier"][hit], qg["identifier"][hit]) and np.array_equal(qo["geometry_type"][hit], qg["geometry_type"][hit]), tag + " extend identifier"
    o.stage("shade"), g.stage("shade")
    ko, kg = o.counters(), g.counters()
    assert kg["device_error"] == 0 and ko["primary_ray_cnt"] == kg["primary_ray_cnt"] and ko["shadow_ray_cnt"] == kg["shadow_ray_cnt"], tag
    assert_state_equal(o.ray_queue(1, ko["primary_ray_cnt"]), g.ray_queue(1, kg["primary_ray_cnt"]), tag + " survivors")
    assert o.shadow_queue(ko["shadow_ray_cnt"]).tobytes() == g.shadow_queue(kg["shadow_ray_cnt"]).tobytes(), tag + " shadow rays"
    o.stage("connect"), g.stage("connect")
    assert o.counters()["n_shadow_visible"] == g.counters()["n_shadow_visible"], tag
    return ko


@pytest.mark.parametrize("R,ranks", [(8, (0, 1, 2, 3, 4, 5, 6, 7)), (2, (0, 1)), (4, (0, 1, 2, 3))])
def test_c4_sharded_million_triangle_scene(orc, hip, R, ranks):
    """BASELINE config C4's workload per rank: mesh706 at 1920x1080, the reference's queue size, rows y % R == rank"""
    sc, nodes, prims = built_scene("mesh706")
    spp = 8
    for rank in ranks:
        tag = f"C4 rank {rank} of {R}"
        o = orc.Oracle(W1080, H1080, N2M, rank=rank, nranks=R, flags=1)
        g = hip.Renderer(W1080, H1080, N2M, rank=rank, nranks=R, flags=1)
        o.load_scene(sc, nodes, prims), g.load_scene(sc, nodes, prims)
        first_wavefront_bit_exact(o, g, tag, min_hit=0.2)
        o.close(), g.close()
        # a whole render of this rank's shard: conservation, and `a == spp` on owned rows only
        g = hip.Renderer(W1080, H1080, N2M, rank=rank, nranks=R, flags=1)
        g.load_scene(sc, nodes, prims)
        g.render(spp)
        k = g.counters()
        local = W1080 * (H1080 // R)
        assert k["device_error"] == 0 and k["total_primary_rays"] == spp * local, tag
        assert k["total_extend_rays"] == k["total_primary_rays"] + k["n_survive"], tag
        b = g.blit_buffer().reshape(H1080, W1080, 4)
        assert np.all(b[rank::R, :, 3] == spp), tag + ": sample counts on owned rows"
        mask = np.ones(H1080, dtype=bool)
        mask[rank::R] = False
        assert not b[mask].any(), tag + ": something was written outside the rank's rows"
        assert np.all(np.isfinite(b)) and np.all(b[..., :3] >= 0), tag
        g.close()


def test_c5_ten_million_triangles_4k(orc, hip):
    """BASELINE config C5's scene and resolution on one GPU; the tree comes from the product's own builder"""
    from tyrant_amd import scenes

    W, H = 3840, 2160
    sc = scenes.glass_dof_scene(2236)
    assert sc.triangles.shape[0] == 9999402 and sc.triangle_materials and sc.camera.lensRadius > 0
    hip.set_build_threads(16)
    nodes, prims = hip.bvh_build(sc.triangles)  # tyr_bvh_build: 12.6 M nodes
    hip.set_build_threads(0)
    assert nodes.shape[0] > 12_000_000 and int(nodes["primitiveCount"].max()) <= 4
    g = hip.Renderer(W, H, N2M, flags=1)
    g.load_scene(sc, nodes, prims)
    info = g.scene_info()
    assert info["n_prims"] == 9999402
    assert 0 < info["n_quad_nodes"] < info["max_quad_nodes"] == 1 << 25, info   # the 25-bit quad index of an interior reference
    leaves = nodes["primitiveCount"] > 0
    assert int((nodes["offset"][leaves].astype(np.int64) + nodes["primitiveCount"][leaves]).max()) <= info["max_prim_offset"] == 1 << 26, info  # 26-bit primitive offset
    assert info["n_pair_nodes"] == 0  # (a ctx without the counting flags neither lays out nor uploads the pair nodes: 0.4 GB here)
    assert hip.layout_probe(nodes, prims, True)["n_pair_nodes"] == int((~leaves).sum())  # ... which would be one per interior node
    print("C5 upload: layout %.3f s, allocation + copies %.3f s, %.2f GB resident" % (info["upload_layout_s"], info["upload_copy_s"], info["device_bytes"] / 1e9))
    assert info["upload_layout_s"] < 1.0, info  # (round 4's serial layout passes took ~2 s here; 16 threads: ~0.2 s)
    assert 16 <= info["quad_max_stack"] <= 48, info  # no traversal of this tree can need more than the wide drain's 48 stack entries (nor the 64 of bvh.h:124)
    print("C5 quad_max_stack", info["quad_max_stack"])
    o = orc.Oracle(W, H, N2M, flags=1)
    o.load_scene(sc, nodes, prims)
    # the scan-line cursor walks down the frame: the first 2 Mi slots are 546 rows of sky, the mesh comes into view in the
    # next wavefronts (which also carry the first bounce rays): three wavefronts, every stage of each bit-exact
    shadow = 0
    for it in range(3):
        k = first_wavefront_bit_exact(o, g, f"C5 wavefront {it}", min_hit=0.0)
        shadow += k["shadow_ray_cnt"]
        o.stage("end"), g.stage("end")
    ko, kg = o.counters(), g.counters()
    assert shadow > 100000 and ko["n_survive"] == kg["n_survive"] > 100000, (shadow, ko["n_survive"], kg["n_survive"])
    o.close(), g.close()
    # a WHOLE render at 4K against `orc_render` (round 6; rounds 1-5 checked conservation only): 1 spp, a GPU-sized queue (every primary in
    # flight: 8.3 M camera rays through the 12.6 M-node tree with the thin lens, then the drain), default tuning -- iterations, every
    # counter, one finished path per pixel, radiance <= 1e-5.  ~45 s of one host core for the oracle's side.
    from test_gpu_parity import assert_accum_close

    o = orc.Oracle(W, H, W * H, flags=1)
    g = hip.Renderer(W, H, W * H, flags=1)
    o.load_scene(sc, nodes, prims), g.load_scene(sc, nodes, prims)
    it_o, it_g = o.render(1), g.render(1)
    ko, k = o.counters(), g.counters()
    assert k["device_error"] == 0, k  # bit 1 = the 64-entry traversal stack (bvh.h:124) overflowed
    assert it_o == it_g, (it_o, it_g)
    for f in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible", "start_position", "frame", "primary_ray_cnt", "shadow_ray_cnt"):
        assert ko[f] == k[f], (f, ko[f], k[f])
    assert k["total_primary_rays"] == W * H and k["total_extend_rays"] == k["total_primary_rays"] + k["n_survive"]
    b = g.blit_buffer()
    assert np.all(b[:, 3] == 1) and np.all(np.isfinite(b)) and np.all(b[:, :3] >= 0)
    assert_accum_close(o.blit_buffer(), b, "C5 render at 4K, queue W x H, 1 spp")


def test_resolve_is_bit_exact(orc, hip):
    """blit_onto_framebuffer (kernel.cu:648-662): rgb / a -> c / (c + 1) -> pow(1 / 2.2), on the SAME accumulation buffer on
    both sides (the oracle's, copied into the caller-owned device buffer), pixels without a finished path (0 / 0) included"""
    import torch

    W, H, N = 96, 64, 2048  # N < W * H: after two iterations most pixels have no finished path yet
    sc, nodes, prims = built_scene("tyrant_default")
    o = orc.Oracle(W, H, N)
    o.load_scene(sc, nodes, prims)
    for _ in range(3):
        o.launch_kernels()
    acc = o.blit_buffer()
    assert (acc[:, 3] == 0).any() and (acc[:, 3] > 0).any()
    acc[5] = (0.0, 0.0, 0.0, 2.0)       # finished paths that found nothing: 0 / 2
    acc[6] = (1e30, 4.0, 1e-30, 1.0)    # c / (c + 1) at both ends of the range
    expect = np.zeros_like(acc)
    L = orc.lib()
    src = np.ascontiguousarray(acc)
    # the oracle resolves its own buffer: write the edited one back through the pointer it hands out
    C.memmove(L.orc_blit_buffer(o.h), src.ctypes.data, src.nbytes)
    L.orc_resolve(o.h, expect.ctypes.data)
    dev_acc = torch.from_numpy(src.reshape(-1)).cuda()
    dev_out = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    g = hip.Renderer(W, H, N, blit_buffer=dev_acc.data_ptr())
    g.resolve_into(dev_out.data_ptr())
    got = dev_out.cpu().numpy().reshape(-1, 4)
    assert np.array_equal(bits(got), bits(expect)), f"{np.count_nonzero(bits(got) != bits(expect))} components differ"
    assert np.isnan(got[acc[:, 3] == 0][:, :3]).all()  # 0 / 0, as in the reference
    g.close()


@pytest.mark.parametrize("name", ["cornell36", "soup2k", "mesh32"])
def test_reference_anyhit_fixture_through_the_connect_kernel(hip, name):
    """the committed answers of the reference's CachedBVH::intersectSimple (bvh.h:213-256), reproduced by the HIP connect
    kernel: the fixture's rays go in as ShadowQueue records with colour (1, 1, 1) and one pixel each; a pixel receives
    its colour exactly when the reference found no occluder"""
    from tyrant_amd import scenes

    z = np.load(os.path.join(GOLDEN, f"ref_traverse_{name}.npz"))
    nodes = np.ascontiguousarray(z["nodes"]).view(scenes.NODE_DTYPE).reshape(-1)
    prims = np.ascontiguousarray(z["prims"]).view(scenes.TRIANGLE_DTYPE).reshape(-1)
    n = z["origin"].shape[0]
    W, H = 64, n // 64
    g = hip.Renderer(W, H, n)
    g.upload(nodes, prims)
    s = scenes.cornell_spheres()
    s["position"] = np.array([0.0, 1e6, -1e6], dtype=np.float32)  # only the BVH answers
    s["radius"] = 1.0
    g.set_spheres(s)
    sh = np.zeros(n, dtype=scenes.SHADOW_DTYPE)
    sh["origin"], sh["direction"], sh["closestDistance"] = z["origin"], z["direction"], z["closest"]
    sh["color"] = 1.0
    sh["buffer_index"] = np.arange(n, dtype=np.int32)
    g.stage("begin")
    g.import_shadow_queue(sh)
    g.stage("connect")
    k = g.counters()
    occluded = z["anyhit"].astype(bool)
    assert k["device_error"] == 0 and k["n_shadow_visible"] == int((~occluded).sum()), name
    b = g.blit_buffer()
    assert np.array_equal(b[:, 0] == 1.0, ~occluded) and np.array_equal(b[:, :3].sum(axis=1) == 0.0, occluded), name


def test_product_builder_builds_c3_on_this_host(orc, hip):
    """tyr_bvh_build (task-parallel, 16 threads) on the GPU box's host cores: C3's 996,882 triangles,
    node and primitive bytes identical to the oracle's restatement of bvh.cpp:3-225 (and to the serial build)"""
    sc, nodes_o, prims_o = built_scene("mesh706")
    hip.set_build_threads(16)
    nodes, prims = hip.bvh_build(sc.triangles)
    hip.set_build_threads(1)
    nodes1, prims1 = hip.bvh_build(sc.triangles)
    hip.set_build_threads(0)
    assert nodes.tobytes() == nodes_o.tobytes() and prims.tobytes() == prims_o.tobytes()
    assert nodes1.tobytes() == nodes_o.tobytes() and prims1.tobytes() == prims_o.tobytes()


def test_stack_bound_of_the_tree_gates_the_wide_drain(orc, hip):
    """The layout pass computes the most stack entries ANY traversal of the quad tree can need (tyr_scene_info.quad_max_stack);
    the four-lanes-to-a-ray drain holds a ray's stack in 48 LDS entries and is entered only on trees that cannot need more.
    The benchmark trees are far below; degenerate chains (triangles at geometrically growing distances: every SAH split
    peels a few off) are above -- there the rays stay one to a lane (64 entries, as the reference's nodesToVisit[64],
    bvh.h:124) and the render still equals the oracle's, with no overflow reported."""
    from tyrant_amd import scenes

    for name, lo, hi in (("mesh128", 8, 48), ("cornell_soup10k", 8, 48), ("mesh706", 16, 48)):
        sc, nodes, prims = built_scene(name)
        g = hip.Renderer(64, 48, 4096, flags=1 if sc.triangle_materials else 0)
        g.load_scene(sc, nodes, prims)
        assert lo <= g.scene_info()["quad_max_stack"] <= hi, (name, g.scene_info())
        print(name, "quad_max_stack", g.scene_info()["quad_max_stack"])
        g.close()
    # six chains of triangles at +-2^1 .. +-2^40 along the three axes: every SAH split peels a few far ones off, the tree
    # is 37 levels deep and a traversal could (for some ray) hold 55 entries -- more than the wide drain's 48, fewer than 64
    n = 40
    chains = []
    for axis in range(3):
        for sign in (1.0, -1.0):
            v = np.zeros((n, 3), np.float32)
            v[:, axis] = (2.0 ** np.arange(1, n + 1)).astype(np.float32) * np.float32(sign)
            chains.append(v)
    v0 = np.concatenate(chains)
    tris = scenes.make_triangles(v0, v0 + np.float32([0.3, 0.9, 0.1]), v0 + np.float32([0.1, 0.4, 1.1]))
    tris = np.concatenate([tris, scenes.make_triangles(v0, v0 + np.float32([0.1, 0.4, 1.1]), v0 + np.float32([0.3, 0.9, 0.1]))])  # both faces
    nodes, prims = orc.bvh_build(tris, scenes.triangle_bboxes(tris))
    sc = scenes.SceneData("chains", tris, scenes.cornell_spheres(), scenes.Camera(position=(-40.0, -30.0, 14.0), direction=(0.76, 0.57, -0.3), up=(0.0, 0.0, 1.0)))
    W, H, N, spp = 96, 64, 6144, 3
    o = orc.Oracle(W, H, N)
    g = hip.Renderer(W, H, N)
    for r in (o, g):
        r.load_scene(sc, nodes, prims)
    info = g.scene_info()
    assert info["quad_max_stack"] > 48, info  # the wide drain is off for this tree
    assert o.render(spp) == g.render(spp)
    ko, kg = o.counters(), g.counters()
    assert kg["device_error"] == 0
    for f in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible"):
        assert ko[f] == kg[f], f
    bo, bg = o.blit_buffer(), g.blit_buffer()
    assert np.array_equal(bo[:, 3], bg[:, 3]) and np.allclose(bg[:, :3], bo[:, :3], rtol=1e-5, atol=1e-6)
    assert kg["total_extend_rays"] > W * H * spp  # (rays did reach the chains and bounce)


BENCH_N, BENCH_SPP = 8 * W1080 * H1080, 8  # bench.py's default job: every primary ray of an 8-spp render in flight (job_shape)


@pytest.mark.parametrize("N,spp,knobs,scene", [(BENCH_N, BENCH_SPP, {}, "mesh706"), (W1080 * H1080, 1, {}, "mesh706"), (N2M, 2, {}, "mesh706"), (BENCH_N, BENCH_SPP, {}, "mesh706_framed")],
                         ids=["bench_shape_16M_8spp", "queue_WxH_1spp", "queue_2Mi_2spp", "framed_16M_8spp"])
def test_benchmarked_render_path_matches_oracle_at_full_size(orc, hip, N, spp, knobs, scene):
    """The code path bench.py times -- `tyr_render` with DEFAULT tuning: merged extend(i+1) + connect(i) launches
    (`k_trace_flat`), run-ahead, the four-lanes-per-ray drain -- on C3 (996,882 triangles) at 1920x1080 against `orc_render`
    (main.cpp:164-170 looping kernel.cu:664-748): same iteration count, every counter equal, every pixel exactly `spp`
    finished paths, radiance within 1e-5 relative.
    `bench_shape_16M_8spp` IS the job the driver times (C3, queue 8 x W x H = 16,588,800, 8 spp): its second wavefront is
    ~10 M rays, so `k_trace_flat<12, 768u>` -- the six-waves-per-SIMD form chosen from TYR_TUNE_WIDE_BLOCK_MIN_ITEMS = 3 Mi
    items -- meets the oracle on a FULL persistent grid here, not only on partial blocks; its counters are also held against
    tests/golden/bench_c3_counters.json (what bench.py checks its timed renders with).  The other two stay below 3 Mi rays
    per launch (the 256-thread form): one ray per pixel in flight, and the reference's own 2 Mi slots.
    `framed_16M_8spp` (round 6) is bench.py's secondary workload `c3_framed`: the same scene and job from scenes.FRAMED_CAMERA, where
    the room's opening fills the frame -- 100.5 M rays per render instead of 42.4 M, 77 % of the extend rays in the tree instead of 39 %
    -- held against tests/golden/bench_c3_framed_counters.json likewise (~2.5 min of one host core for the oracle's side)."""
    from test_gpu_parity import assert_accum_close

    sc, nodes, prims = built_scene("mesh706")
    if scene == "mesh706_framed":  # (the same triangles and tree, another camera)
        from tyrant_amd import scenes

        sc = scenes.mesh_scene_framed(706)
    o = orc.Oracle(W1080, H1080, N, flags=1)
    g = hip.Renderer(W1080, H1080, N, flags=1)
    o.load_scene(sc, nodes, prims), g.load_scene(sc, nodes, prims)
    g.set_tuning(**knobs)  # {}: the defaults bench.py times
    it_o, it_g = o.render(spp), g.render(spp)
    ko, kg = o.counters(), g.counters()
    assert kg["device_error"] == 0
    assert it_o == it_g, (it_o, it_g)
    for f in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible", "start_position", "frame", "primary_ray_cnt", "shadow_ray_cnt"):
        assert ko[f] == kg[f], (f, ko[f], kg[f])
    assert ko["total_primary_rays"] == spp * W1080 * H1080
    bo, bg = o.blit_buffer(), g.blit_buffer()
    assert np.all(bg[:, 3] == spp)
    assert_accum_close(bo, bg, f"C3 render, queue {N}, {spp} spp")
    if (N, spp) == (BENCH_N, BENCH_SPP):
        with open(os.path.join(GOLDEN, "bench_c3_framed_counters.json" if scene == "mesh706_framed" else "bench_c3_counters.json")) as f:
            gold = json.load(f)
        assert gold["job"]["queue_size"] == N and gold["job"]["spp"] == spp and gold["job"]["triangles"] == prims.shape[0]
        assert gold["per_render"]["iterations"] == it_g
        for f_ in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible"):
            assert gold["per_render"][f_] == kg[f_] == ko[f_], f_
        # the same job again from the same frame counter (what every timed step of bench.py is): the same counters
        g.set_frame(1)
        g.reset_accum()
        assert g.render(spp) == it_g
        k2 = g.counters()
        for f_ in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible"):
            assert k2[f_] == 2 * kg[f_], f_
        assert_accum_close(bo, g.blit_buffer(), "the bench job rendered again from frame 1")


def test_c2_render_matches_oracle_at_full_size(orc, hip):
    """BASELINE config C2 as it is quoted: Cornell box + 10,000 random diffuse triangles at 1920x1080, the reference's 2 Mi-slot
    queue, 8 spp (round 5 ran 2): `tyr_render` with default tuning against `orc_render` -- iteration count (20), every counter,
    every pixel's path count, radiance <= 1e-5.  The oracle's render is ~35 s of one host core."""
    from test_gpu_parity import assert_accum_close

    sc, nodes, prims = built_scene("cornell_soup10k")
    flags = 1 if sc.triangle_materials else 0
    o = orc.Oracle(W1080, H1080, N2M, flags=flags)
    g = hip.Renderer(W1080, H1080, N2M, flags=flags)
    o.load_scene(sc, nodes, prims), g.load_scene(sc, nodes, prims)
    spp = 8
    it_o, it_g = o.render(spp), g.render(spp)
    ko, kg = o.counters(), g.counters()
    assert kg["device_error"] == 0
    assert it_o == it_g, (it_o, it_g)
    for f in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible", "start_position", "frame", "primary_ray_cnt", "shadow_ray_cnt"):
        assert ko[f] == kg[f], (f, ko[f], kg[f])
    assert ko["total_primary_rays"] == spp * W1080 * H1080
    bo, bg = o.blit_buffer(), g.blit_buffer()
    assert np.all(bg[:, 3] == spp)
    assert_accum_close(bo, bg, "C2 render, queue 2 Mi, 8 spp")


def test_set_frame_restarts_the_seed_sequence(orc, hip):
    """tyr_set_frame (kernel.cu:667's `frame`, which the reference can only count on): after it, tyr_reset_accum and a budget of
    whole passes over the pixels, a render repeats the render that began at that frame -- counters and radiance; 0 is refused."""
    from test_gpu_parity import assert_accum_close

    sc, nodes, prims = built_scene("mesh128")
    W, H, N, spp = 160, 96, 8192, 3
    g = hip.Renderer(W, H, N, flags=1)
    g.load_scene(sc, nodes, prims)
    it1 = g.render(spp)
    k1, b1 = g.counters(), g.blit_buffer()
    g.reset_accum()
    assert g.render(spp) > 0
    k2 = g.counters()
    assert k2["total_extend_rays"] - k1["total_extend_rays"] != k1["total_extend_rays"]  # other frames, other random numbers
    g.set_frame(1)
    g.reset_accum()
    assert g.render(spp) == it1
    k3 = g.counters()
    for f in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible"):
        assert k3[f] - k2[f] == k1[f], f
    assert k3["frame"] == k1["frame"]
    assert_accum_close(b1, g.blit_buffer(), "render repeated from frame 1")
    with pytest.raises(hip.TyrError):
        g.set_frame(0)
