"""Host side of the hot path (BVH builder, bboxes, camera) and the C-ABI surface.  CPU only:
the library loads and exports every symbol include/tyr_c.h declares; no compute call needs a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, built_scene


def test_library_exports_every_declared_symbol(hip):
    """every function declared in include/tyr_c.h is exported, and the binding lists exactly those"""
    hdr = open(os.path.join(ROOT, "include", "tyr_c.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tyr_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = hip.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in tyr_c.h but not exported"
    assert declared == set(hip.SYMBOLS), (declared ^ set(hip.SYMBOLS))
    assert L.tyr_abi_version() == 5


def test_abi_struct_sizes(hip):
    from tyrant_amd import scenes

    assert C.sizeof(hip.Config) == 40
    assert C.sizeof(hip.CameraC) == 44
    assert C.sizeof(hip.Counters) == 24 + 28 * 8
    assert scenes.TRIANGLE_DTYPE.itemsize == 40 and scenes.NODE_DTYPE.itemsize == 32
    assert scenes.RAY_DTYPE.itemsize == 60 and scenes.SHADOW_DTYPE.itemsize == 44 and scenes.SPHERE_DTYPE.itemsize == 44


def test_error_paths_without_compute(hip):
    L = hip.lib()
    assert L.tyr_status_string(0) == b"ok"
    assert b"no HIP device" in L.tyr_status_string(-2)
    h = C.c_void_p()
    assert L.tyr_create(C.byref(h), None) == -1
    bad = hip.Config(0, 64, 1024, 0, 0, 1, 0, None)
    assert L.tyr_create(C.byref(h), C.byref(bad)) == -1
    bad = hip.Config(64, 64, 1024, 0, 3, 2, 0, None)  # rank >= nranks
    assert L.tyr_create(C.byref(h), C.byref(bad)) == -1
    bad = hip.Config(64, 63, 1024, 0, 0, 2, 0, None)  # rows not divisible by nranks
    assert L.tyr_create(C.byref(h), C.byref(bad)) == -1
    assert L.tyr_launch_kernels(None) == -1 and L.tyr_destroy(None) == 0
    assert L.tyr_bvh_build(None, -1, None, None, 2) == -1
    assert L.tyr_bvh_build(None, 0, None, None, 2) == 0  # bvh.cpp:8-10: empty input, no nodes
    assert L.tyr_bvh_build(None, 0, None, None, 0) == -1  # PartitionAlgorithm::Middle is unimplemented (bvh.cpp:190-193)


@pytest.mark.parametrize("name", ["cornell36", "cornell_soup2k", "mesh32", "mesh128", "tyrant_default"])
def test_product_builder_matches_oracle_bytes(hip, orc, name):
    """tyr_bvh_build emits the same node array and the same reordered triangles as the oracle's restatement"""
    from tyrant_amd import scenes

    sc, nodes_o, prims_o = built_scene(name)
    bb = hip.triangle_bboxes(sc.triangles)
    assert bb.tobytes() == scenes.triangle_bboxes(sc.triangles).tobytes()
    nodes_p, prims_p = hip.bvh_build(sc.triangles, bb)
    assert nodes_p.tobytes() == nodes_o.tobytes()
    assert prims_p.tobytes() == prims_o.tobytes()


@pytest.mark.parametrize("threads", [1, 2, 3, 8])
def test_parallel_builder_is_byte_identical(hip, orc, threads):
    """the task-parallel build emits the serial builder's bytes for every thread count, both on a
    scene large enough to fan out over several levels and on one below the task grain"""
    from tyrant_amd import scenes

    try:
        for sc in (scenes.cornell_soup(60000, seed=5), scenes.mesh_scene(96), scenes.cornell_soup(3000, seed=2)):
            bb = scenes.triangle_bboxes(sc.triangles)
            nodes_o, prims_o = orc.bvh_build(sc.triangles, bb)
            hip.set_build_threads(threads)
            nodes_p, prims_p = hip.bvh_build(sc.triangles, bb)
            assert nodes_p.tobytes() == nodes_o.tobytes() and prims_p.tobytes() == prims_o.tobytes()
            nodes_e, prims_e = hip.bvh_build(sc.triangles, bb, algo=1)  # EqualCounts: same bytes as its own serial build
            hip.set_build_threads(1)
            nodes_s, prims_s = hip.bvh_build(sc.triangles, bb, algo=1)
            assert nodes_e.tobytes() == nodes_s.tobytes() and prims_e.tobytes() == prims_s.tobytes()
    finally:
        hip.set_build_threads(0)
    with pytest.raises(hip.TyrError):
        hip.set_build_threads(-1)


def test_builder_structure(hip):
    sc, nodes, prims = built_scene("cornell_soup2k")
    n = prims.shape[0]
    leaves = nodes[nodes["primitiveCount"] > 0]
    inner = nodes[nodes["primitiveCount"] == 0]
    assert len(nodes) == 2 * len(leaves) - 1  # full binary tree
    assert leaves["primitiveCount"].sum() == n and leaves["primitiveCount"].max() <= 4  # bvh.h:78
    assert np.all(leaves["splitAxis"] == 0) and np.all(nodes["pad"] == 0)  # value-initialised bytes (bvh.cpp:11)
    assert np.all(inner["splitAxis"] <= 2)
    # leaves partition [0, n) in depth-first order
    offs = leaves["offset"].astype(np.int64)
    assert offs[0] == 0 and np.all(np.diff(offs) == leaves["primitiveCount"][:-1])
    # every child box is inside its parent's box; second child index is in range
    idx = np.nonzero(nodes["primitiveCount"] == 0)[0]
    for i in idx[:: max(1, len(idx) // 500)]:
        for c in (i + 1, nodes["offset"][i]):
            assert i < c < len(nodes)
            assert np.all(nodes["bounds"][c, 0] >= nodes["bounds"][i, 0]) and np.all(nodes["bounds"][c, 1] <= nodes["bounds"][i, 1])
    # the reordered triangles are a permutation of the input
    key = lambda t: np.sort(t.view(np.uint8).reshape(-1, 40)[:, :37].copy().view(np.dtype((np.void, 37))).ravel())  # noqa: E731
    assert np.array_equal(key(np.ascontiguousarray(sc.triangles)), key(np.ascontiguousarray(prims)))


def test_builder_edge_cases(hip):
    from tyrant_amd import scenes

    # one triangle: the root is a leaf
    t = scenes.make_triangles([[0, 0, 0]], [[1, 0, 0]], [[0, 1, 0]])
    nodes, prims = hip.bvh_build(t)
    assert len(nodes) == 1 and nodes["primitiveCount"][0] == 1 and nodes["offset"][0] == 0
    # identical centroids: one leaf holds them all (bvh.cpp:103-111), also beyond 4 primitives
    t = scenes.make_triangles(np.zeros((9, 3)), np.tile([1, 0, 0], (9, 1)), np.tile([0, 1, 0], (9, 1)))
    nodes, prims = hip.bvh_build(t)
    assert len(nodes) == 1 and nodes["primitiveCount"][0] == 9
    # non-finite geometry is rejected instead of producing a broken tree
    bad = scenes.make_triangles([[0, 0, 0]], [[np.inf, 0, 0]], [[0, 1, 0]])
    with pytest.raises(hip.TyrError):
        hip.bvh_build(np.concatenate([t, bad]))
    # EqualCounts splits at the median
    t = scenes.random_soup(257, seed=3)
    nodes, prims = hip.bvh_build(t, algo=1)
    leaves = nodes[nodes["primitiveCount"] > 0]
    assert leaves["primitiveCount"].sum() == 257 and len(nodes) == 2 * len(leaves) - 1


def test_camera_update(hip):
    """Camera::update, camera.cpp:46-52"""
    for h, v in ((0.0, 0.0), (0.3, -0.2), (-1.476, -0.398), (3.0, 1.5)):
        d = hip.camera_update(h, v)
        want = np.array([np.cos(v) * np.sin(h), np.cos(v) * np.cos(h), np.sin(v)])
        assert np.allclose(d, want / np.linalg.norm(want), atol=2e-7)
    assert list(hip.camera_update(0.0, 0.0)) == [0.0, 1.0, 0.0]


def test_default_spheres_match_oracle(hip, orc):
    from tyrant_amd import scenes

    s = hip.default_spheres()
    o = np.zeros(7, dtype=scenes.SPHERE_DTYPE)
    orc.lib().orc_default_spheres(o.ctypes.data)
    assert s.tobytes() == o.tobytes() == scenes.reference_spheres().tobytes()
    assert s["refl"][6] == scenes.LIGHT and s["radius"][4] == np.float32(1e4)  # kernel.cu:678, 680


def test_hand_declared_rccl_prototypes_are_checked_against_rccl_h(tmp_path):
    """host/dist.cpp binds RCCL with dlsym through prototypes written by hand (host/rccl_slice.hpp); host/rccl_check.cpp -- part
    of every build of the library -- static_asserts each of them against <rccl/rccl.h>.  Here: the check compiles as it
    stands, and a prototype that drifts (ncclSend without its `peer` argument; a 64-byte id) does NOT."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("hipcc or <rccl/rccl.h> not installed")
    host = os.path.join(ROOT, "tyrant_amd", "csrc", "host")
    assert "host/rccl_check.cpp" in open(os.path.join(ROOT, "tyrant_amd", "csrc", "Makefile")).read()  # a drift fails the BUILD
    check = open(os.path.join(host, "rccl_check.cpp")).read()
    slice_ = open(os.path.join(host, "rccl_slice.hpp")).read().replace('"../../../include/tyr_c.h"', f'"{ROOT}/include/tyr_c.h"')

    def compiles(slice_text):
        (tmp_path / "rccl_slice.hpp").write_text(slice_text)
        (tmp_path / "rccl_check.cpp").write_text(check)
        p = subprocess.run([hipcc, "-std=c++17", "-fsyntax-only", str(tmp_path / "rccl_check.cpp")], capture_output=True, text=True, timeout=300)
        return p.returncode == 0, p.stderr

    ok, err = compiles(slice_)
    assert ok, err[-2000:]
    assert "NOT checked" not in err  # (<rccl/rccl.h> was found: the assertions were evaluated)
    drifted = slice_.replace("int (*Send)(const void*, size_t, int, int, nccl_comm, hipStream_t)", "int (*Send)(const void*, size_t, int, nccl_comm, hipStream_t)")
    assert drifted != slice_
    ok, err = compiles(drifted)
    assert not ok and "Rccl::Send no longer matches ncclSend" in err
    ok, err = compiles(slice_.replace("char internal[TYR_DIST_ID_BYTES];", "char internal[64];"))
    assert not ok and "ncclUniqueId" in err


def _layout_cases():
    from tyrant_amd import scenes

    box = scenes.cornell_box().triangles
    base = scenes.cornell_soup(500).triangles
    return {
        "cornell36": box,
        "soup10k": scenes.cornell_soup(10000).triangles,
        "mesh128": scenes.mesh_scene(128).triangles,
        "longleaf": np.concatenate([base, np.repeat(base[:3], 100, axis=0)]),  # three leaves of 101 identical triangles: chains of synthetic records
        "one": box[:1].copy(),
        "two": box[:2].copy(),
        "only_long": np.repeat(box[:1], 150, axis=0),  # the whole tree is one over-long leaf
        "mesh706": scenes.mesh_scene(706).triangles,  # C3
    }


def test_device_layout_is_byte_identical_to_the_serial_one_at_every_thread_count(hip):
    """tyr_scene_upload's host half (host/bvh_layout.cpp; tyr_layout_probe runs it without a device): round 5 made its passes
    parallel.  The quad-node, pair-node and triangle arrays must equal, byte for byte, what the serial passes of rounds 1-4
    produced (tests/golden/layout_hashes.json, recorded from that code) -- at 1, 2, 3, 8 and 16 threads, with and without the
    pair layout, over-long leaves included."""
    import json

    from conftest import GOLDEN

    with open(os.path.join(GOLDEN, "layout_hashes.json")) as f:
        gold = json.load(f)["cases"]
    fields = ("n_quad_nodes", "n_staged_nodes", "quad_max_stack", "root_ref", "quad_root_ref", "hash_quads", "hash_tris")
    try:
        for name, tris in _layout_cases().items():
            hip.set_build_threads(0)
            nodes, prims = hip.bvh_build(tris.copy())
            for threads in (1, 2, 3, 8, 16):
                hip.set_build_threads(threads)
                st = hip.layout_probe(nodes, prims, True)
                for f_ in fields + ("n_pair_nodes", "hash_pairs"):
                    assert st[f_] == gold[name][f_], (name, threads, f_)
                st = hip.layout_probe(nodes, prims, False)  # what a ctx without the counting flags uploads: no pair nodes, the same quads
                for f_ in fields:
                    assert st[f_] == gold[name][f_], (name, threads, f_, "no pairs")
                assert st["n_pair_nodes"] == 0
    finally:
        hip.set_build_threads(0)


def test_device_layout_refuses_arrays_that_are_not_a_depth_first_tree(hip):
    """the layout's parallel passes rely on the reference's array order (bvh.cpp:195-202: first child = index + 1, a subtree is
    a contiguous range).  Arrays that break it -- a second child outside its parent's range, a node named twice or never, an
    offset past the end, a non-finite box, a leaf reaching past the primitives -- are refused (TYR_ERR_INVALID), at any thread
    count, never followed."""
    from tyrant_amd import scenes

    nodes, prims = hip.bvh_build(scenes.cornell_soup(20000).triangles)
    interior = np.flatnonzero(nodes["primitiveCount"] == 0)
    leaves = np.flatnonzero(nodes["primitiveCount"] > 0)

    def refused(mutate):
        n = nodes.copy()
        mutate(n)
        for threads in (1, 8):
            hip.set_build_threads(threads)
            try:
                hip.layout_probe(n, prims)
            except hip.TyrError as e:
                assert e.status == -1, e  # TYR_ERR_INVALID
            else:
                return False
        return True

    try:
        hip.set_build_threads(8)
        assert hip.layout_probe(nodes, prims)["n_quad_nodes"] > 0  # the array as built is accepted
        deep = interior[len(interior) // 2]
        assert refused(lambda n: n["offset"].__setitem__(deep, len(n) - 1))            # second child far outside the parent's subtree (and named twice)
        assert refused(lambda n: n["offset"].__setitem__(deep, n["offset"][deep] + 1))  # the real second child is never named, its neighbour twice
        assert refused(lambda n: n["offset"].__setitem__(interior[3], len(n) + 5))      # past the end
        assert refused(lambda n: n["offset"].__setitem__(interior[3], interior[3] + 1))  # second child == first child
        assert refused(lambda n: n["bounds"].__setitem__((leaves[7], 1, 2), np.inf))
        assert refused(lambda n: n["offset"].__setitem__(leaves[5], len(prims)))        # a leaf's primitives past the array
        assert refused(lambda n: n["splitAxis"].__setitem__(interior[9], 3))
        t = prims.copy()
        t["e1"][11, 0] = np.nan
        with pytest.raises(hip.TyrError):
            hip.layout_probe(nodes, t)
    finally:
        hip.set_build_threads(0)
