"""tyr_bvh_build_device (hip/bvh_build_dev.hip): the reference's binned-SAH build (bvh.cpp:3-225) on the GPU.  The gate is the
one the host builders passed (tests/test_ref_pins.py): BYTES -- every node and the primitive order equal the product's host
builder's, which equal the oracle's and the reference's own bvh.cpp (tests/golden/ref_build_*.npz; the C3 and C5 trees by
SHA-256, tests/golden/ref_build_hashes.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _same(hip, tris, what, bboxes=None):
    hn, hp = hip.bvh_build(tris, bboxes)
    dn, dp, sec = hip.bvh_build_device(tris, bboxes)
    assert dn.shape == hn.shape, (what, dn.shape, hn.shape)
    assert dn.tobytes() == hn.tobytes(), f"{what}: {np.count_nonzero(dn.view(np.uint8).reshape(len(dn), 32) != hn.view(np.uint8).reshape(len(hn), 32))} node bytes differ, first node {np.flatnonzero((dn.view(np.uint8).reshape(len(dn), 32) != hn.view(np.uint8).reshape(len(hn), 32)).any(1))[:5]}"
    assert dp.tobytes() == hp.tobytes(), f"{what}: primitive order differs"
    return dn, dp, sec


def _cases():
    from tyrant_amd import scenes

    box = scenes.cornell_box().triangles
    soup = scenes.cornell_soup(500).triangles
    cases = {
        "cornell36": box,
        "one": box[:1],
        "two": box[:2],
        "five": box[:5],
        "soup64": scenes.random_soup(64, seed=3),
        "soup65": scenes.random_soup(65, seed=4),
        "soup2k": scenes.cornell_soup(2000).triangles,
        "soup10k": scenes.cornell_soup(10000).triangles,
        "mesh128": scenes.mesh_scene(128).triangles,
        "tyrant_default": scenes.tyrant_default().triangles,
        "glass_dof48": scenes.glass_dof_scene(48).triangles,
        # identical centroids: leaves far beyond four primitives (bvh.cpp:103-111), small and beyond a task thread's range
        "longleaf": np.concatenate([soup, np.repeat(soup[:3], 100, axis=0)]),
        "only_long": np.repeat(box[:1], 150, axis=0),
        "big_identical": np.concatenate([scenes.random_soup(3000, seed=9), np.repeat(soup[7:8], 5000, axis=0)]),
    }
    # flat geometry: every box has zero extent along z -- node boxes of zero surface area reach the SAH's division (bvh.cpp:150)
    flat = scenes.mesh_scene(40).triangles.copy()
    flat["vert"][:, 2] = 0.0
    flat["e1"][:, 2] = 0.0
    flat["e2"][:, 2] = 0.0
    cases["flat_z0"] = flat
    line = scenes.random_soup(300, seed=12).copy()  # all on one line: zero-area boxes everywhere
    line["vert"][:, 1:] = 0.0
    line["e1"][:, 1:] = 0.0
    line["e2"][:, 1:] = 0.0
    cases["on_a_line"] = line
    neg = scenes.cornell_soup(900).triangles.copy()  # signed zeros among the coordinates
    neg["vert"][::3, 0] = -0.0
    neg["vert"][1::3, 0] = 0.0
    neg["e1"][::2, 0] = 0.0
    neg["e2"][::2, 0] = -0.0
    cases["signed_zeros"] = neg
    # six chains of triangles at +-2^1 .. +-2^40 along the axes: every SAH split peels a few far ones off -- a tree 37 levels deep
    chains = []
    for axis in range(3):
        for sign in (1.0, -1.0):
            v = np.zeros((40, 3), np.float32)
            v[:, axis] = (2.0 ** np.arange(1, 41)).astype(np.float32) * np.float32(sign)
            chains.append(v)
    v0 = np.concatenate(chains)
    cases["chains"] = scenes.make_triangles(v0, v0 + np.float32([0.3, 0.9, 0.1]), v0 + np.float32([0.1, 0.4, 1.1]))
    return {k: np.ascontiguousarray(v) for k, v in cases.items()}


def test_device_build_emits_the_host_builders_bytes(hip):
    from tyrant_amd import scenes

    for name, tris in _cases().items():
        _same(hip, tris, name)
    rng = np.random.default_rng(21)
    for case in range(25):
        n = int(rng.integers(1, 40000))
        t = scenes.random_soup(n, seed=int(rng.integers(1, 1 << 30)))
        if case % 3 == 0:
            k = int(rng.integers(0, n))
            t = np.concatenate([t, np.repeat(t[k:k + 1], int(rng.integers(5, 300)), axis=0)])
            t = t[rng.permutation(len(t))]
        _same(hip, t, f"fuzz {case} ({len(t)} triangles)")
    # a tree built over caller-supplied boxes (not the triangles' own)
    t = scenes.cornell_soup(3000).triangles
    bb = scenes.triangle_bboxes(t)
    bb["bounds"][:, 0, :] -= np.float32(0.25)
    _same(hip, t, "grown boxes", bb)


def test_device_build_refuses_boxes_that_are_not_finite(hip):
    """the check runs in the first kernel that reads the boxes (a host loop over C5's 60 M floats cost as much as the build):
    TYR_ERR_INVALID from both entry points, for a small range (one task thread) and a large one (the level loop)"""
    from tyrant_amd import scenes

    r = hip.Renderer(64, 48, 64 * 48)
    for n in (20, 5000):
        t = scenes.random_soup(n, seed=77)
        for bad in (np.nan, np.inf, -np.inf):
            bb = scenes.triangle_bboxes(t)
            bb["bounds"][n // 2, 1, 2] = bad
            with pytest.raises(hip.TyrError) as e:
                hip.bvh_build_device(t, bb)
            assert e.value.status == -1, (n, bad, e.value.status)  # TYR_ERR_INVALID
            with pytest.raises(hip.TyrError) as e:
                r.build_upload(t, bb)
            assert e.value.status == -1, (n, bad, e.value.status)
        _same(hip, t, f"after the refusals ({n})")  # the library is fine afterwards


def test_device_build_of_the_benchmark_trees_equals_the_reference_bvh_cpp(hip):
    """C3 (1,097,453 nodes) and C5 (12,614,891 nodes): the SHA-256 the reference's own bvh.cpp produced in the authoring container
    (tests/golden/ref_build_hashes.json) -- and how long the device took"""
    from tyrant_amd import scenes

    with open(os.path.join(GOLDEN, "ref_build_hashes.json")) as f:
        gold = json.load(f)
    for key, tris in (("mesh706", scenes.mesh_scene(706).triangles), ("glass2236", scenes.glass_dof_scene(2236).triangles)):
        hip.bvh_build_device(tris[:1000])  # (code objects loaded, context warm)
        nodes, prims, sec = hip.bvh_build_device(tris)
        g = gold[key]
        assert nodes.shape[0] == g["nodes"], (key, nodes.shape[0], g["nodes"])
        assert hashlib.sha256(nodes.tobytes()).hexdigest() == g["nodes_sha256"], key
        assert hashlib.sha256(np.ascontiguousarray(prims.view(np.uint8).reshape(-1, 40)[:, :37]).tobytes()).hexdigest() == g["prims_sha256"], key  # (bytes 37-39 of a Triangle are padding)
        print(f"{key}: {len(tris)} triangles -> {len(nodes)} nodes in {sec[0] * 1e3:.1f} ms on the device (+ {sec[1] * 1e3:.1f} ms of copies in and out)")


# ---- hip/bvh_layout_dev.hip: the upload's layout pass on the device (TYR_TUNE_LAYOUT_ON_DEVICE), and both halves in one call ----
_LAYOUT_KEYS = ("n_pair_nodes", "n_quad_nodes", "n_staged_nodes", "quad_max_stack", "root_ref", "quad_root_ref", "hash_quads", "hash_tris")


def _layout_matches_host(hip, r, nodes, prims, what, expect_on_device=None):
    """the scene `r` holds in HBM, read back, against the host pass on the same arrays (tyr_scene_hash vs tyr_layout_probe)"""
    want = hip.layout_probe(nodes, prims, want_pairs=False)
    got = r.scene_hash()
    for k in _LAYOUT_KEYS:
        assert got[k] == want[k], (what, k, got[k], want[k])
    if expect_on_device is not None:
        assert r.scene_info()["layout_on_device"] == int(expect_on_device), (what, r.scene_info())


def test_layout_on_the_device_writes_the_host_passs_bytes(hip):
    """every tree of the builder's test list uploaded twice -- layout on the device, layout on the host -- and read back"""
    from tyrant_amd import scenes

    r = hip.Renderer(64, 48, 64 * 48)
    host_only = {"one", "two", "longleaf", "only_long", "big_identical"}  # one-leaf trees, leaves of more than 31 primitives
    for name, tris in _cases().items():
        nodes, prims = hip.bvh_build(tris)
        r.set_tuning(layout_on_device=1)
        r.upload(nodes, prims)
        on_device = r.scene_info()["layout_on_device"]
        if name in host_only:
            assert on_device == 0, name
        elif len(nodes) >= 3 and int(nodes["primitiveCount"].max()) <= 31:
            assert on_device == 1, name
        _layout_matches_host(hip, r, nodes, prims, name)
        r.set_tuning(layout_on_device=0)
        r.upload(nodes, prims)
        _layout_matches_host(hip, r, nodes, prims, name + " (host)", expect_on_device=False)
    rng = np.random.default_rng(5)
    r.set_tuning(layout_on_device=1)
    for case in range(20):
        t = scenes.random_soup(int(rng.integers(3, 60000)), seed=int(rng.integers(1, 1 << 30)))
        nodes, prims = hip.bvh_build(t)
        r.upload(nodes, prims)
        _layout_matches_host(hip, r, nodes, prims, f"fuzz {case}", expect_on_device=len(nodes) >= 3)


def test_layout_on_the_device_leaves_malformed_trees_to_the_host_pass(hip):
    """the error is the host pass's (TYR_ERR_INVALID), and the scene the ctx held stays"""
    from tyrant_amd import scenes

    r = hip.Renderer(64, 48, 64 * 48)
    nodes, prims = hip.bvh_build(scenes.cornell_soup(300).triangles)
    r.upload(nodes, prims)
    before = r.scene_hash()
    interior = np.flatnonzero(nodes["primitiveCount"] == 0)
    bad = {}
    b = nodes.copy()
    b["offset"][interior[3]] = int(interior[3])  # a second child in front of its parent
    bad["backward child"] = b
    b = nodes.copy()
    b["offset"][interior[5]] = b["offset"][interior[2]]  # two parents name one child
    bad["shared child"] = b
    b = nodes.copy()
    b["bounds"][7, 0, 1] = np.nan
    bad["nan box"] = b
    b = nodes.copy()
    leaf = np.flatnonzero(nodes["primitiveCount"] > 0)[4]
    b["offset"][leaf] = len(prims)  # a leaf past the primitives
    bad["leaf out of range"] = b
    for what, arr in bad.items():
        with pytest.raises(hip.TyrError) as e:
            r.upload(arr, prims)
        assert e.value.status == -1, (what, e.value.status)  # TYR_ERR_INVALID
        assert r.scene_hash() == before, what


def test_scene_build_upload_is_build_plus_upload(hip):
    """tyr_scene_build_upload: nodes, primitive order and the scene in HBM equal tyr_bvh_build + tyr_scene_upload's, whichever
    half had to fall back to the host; a render on it equals a render on the two-call scene"""
    from tyrant_amd import scenes

    r = hip.Renderer(64, 48, 64 * 48)
    for name, tris in _cases().items():
        hn, hp = hip.bvh_build(tris)
        nodes, prims, sec = r.build_upload(tris)
        assert nodes.tobytes() == hn.tobytes() and prims.tobytes() == hp.tobytes(), name
        _layout_matches_host(hip, r, hn, hp, name)
        n2, p2, _ = r.build_upload(tris, want_nodes=False)
        assert n2 is None and p2.tobytes() == hp.tobytes(), name
        _layout_matches_host(hip, r, hn, hp, name + " (no nodes asked for)")
    # a counting ctx wants pair nodes: both halves on the host, the same scene as its own upload
    rc = hip.Renderer(64, 48, 64 * 48, flags=hip.TYR_FLAG_COUNT_VISITS)
    t = scenes.cornell_soup(700).triangles
    hn, hp = hip.bvh_build(t)
    nodes, prims, _ = rc.build_upload(t)
    assert nodes.tobytes() == hn.tobytes() and prims.tobytes() == hp.tobytes()
    got = rc.scene_hash()
    want = hip.layout_probe(hn, hp, want_pairs=True)
    for k in _LAYOUT_KEYS + ("hash_pairs",):
        assert got[k] == want[k], k
    # and a picture: the fused scene renders what the two-call scene renders
    sc = scenes.cornell_soup(2000)
    a = hip.Renderer(128, 72, 128 * 72 * 2)
    b = hip.Renderer(128, 72, 128 * 72 * 2)
    hn, hp = hip.bvh_build(sc.triangles)
    a.load_scene(sc, hn, hp)
    b.load_scene(sc, hn, hp)
    b.build_upload(sc.triangles)
    assert a.render(2) == b.render(2)
    assert np.array_equal(a.blit_buffer()[..., 3], b.blit_buffer()[..., 3])
    ka, kb = a.counters(), b.counters()
    for k in ("total_extend_rays", "total_shadow_rays", "n_shadow_visible"):
        assert ka[k] == kb[k], k


def test_layout_of_the_benchmark_trees_on_the_device(hip):
    """C3 and C5: both halves on the device in one call against the host pass's hashes, and what each way costs"""
    from tyrant_amd import scenes

    for key, sc in (("c3", scenes.mesh_scene(706)), ("c5", scenes.glass_dof_scene(2236))):
        hn, hp = hip.bvh_build(sc.triangles)
        r = hip.Renderer(64, 48, 64 * 48)
        r.build_upload(sc.triangles[:2000])  # (code objects loaded)
        nodes, prims, sec = r.build_upload(sc.triangles, want_nodes=False)
        assert prims.tobytes() == hp.tobytes(), key
        _layout_matches_host(hip, r, hn, hp, key, expect_on_device=True)
        r.set_tuning(layout_on_device=1)
        r.upload(hn, hp)
        dev = r.scene_info()
        _layout_matches_host(hip, r, hn, hp, key + " upload", expect_on_device=True)
        r.set_tuning(layout_on_device=0)
        r.upload(hn, hp)
        host = r.scene_info()
        print(f"{key}: {len(hn)} nodes -> {dev['n_quad_nodes']} records.  tyr_scene_build_upload: build {sec[0] * 1e3:.1f} + layout {sec[1] * 1e3:.1f} + copies {sec[2] * 1e3:.1f} ms; "
              f"tyr_scene_upload, layout on the device: copies {dev['upload_copy_s'] * 1e3:.1f} + layout {dev['upload_layout_s'] * 1e3:.1f} ms; on the host: layout {host['upload_layout_s'] * 1e3:.1f} + copies {host['upload_copy_s'] * 1e3:.1f} ms")
