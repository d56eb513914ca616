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


def test_device_build_emits_the_host_builders_bytes(hip):
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
    for name, tris in cases.items():
        _same(hip, np.ascontiguousarray(tris), name)
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
