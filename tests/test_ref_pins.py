"""The oracle AND the product pinned to outputs of the reference's own host sources.

tests/golden/ref_build_*.npz, ref_build_hashes.json and ref_sunsky.npz were produced by the reference's bvh.cpp, Bbox.cpp
and sunsky.cu, compiled unmodified where they lie (oracle/Makefile `ref`, oracle/ref_host_harness.cpp; generator
tests/golden/make_ref_build_golden.py).

  CPU   oracle builder == fixture bytes (SAH); product builder (tyr_bvh_build, host code) == fixture bytes; C3's
        1.1 M-node tree by SHA-256; the oracle's sun / sky / sunsky within SUNSKY_MAX_ULP of the reference's and bit-equal
        on >= 99 % of the components; setup constants and the sun-cone samples; when oracle/_ref is present the
        fixtures are re-derived live.
  GPU   the product builder on the GPU box's host for C3 and C5 (hash); the DEVICE's sun / sky / sunsky / cone sample
        through tyr_sunsky_probe within the same bound.

Tolerance, stated: the reference's atmosphere goes through glibc's expf / powf / pow / acosf here (CUDA's libm on the
original); the numeric contract of this project evaluates the same expressions with correctly rounded fixed-sequence
kernels.  Composed through sky() that is at most 16 ulp (measured: 10) on at most 1 % of the
components (measured: 0.06 %), i.e. <= 2e-6 relative -- two orders inside north_star's 1e-4.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

SUNSKY_MAX_ULP = 16
SUNSKY_MIN_BIT_EQUAL = 0.99
SMALL = ("cornell36", "soup10k", "mesh64")


def ulp_distance(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """distance in units in the last place between two float32 arrays (monotone integer mapping of the bit patterns)"""
    ai = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    bi = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def defined_triangle_bytes(prims: np.ndarray) -> bytes:
    return np.ascontiguousarray(prims.view(np.uint8).reshape(-1, 40)[:, :37]).tobytes()


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


@pytest.fixture(scope="module")
def hashes():
    with open(os.path.join(GOLDEN, "ref_build_hashes.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def sunsky_gold():
    return np.load(os.path.join(GOLDEN, "ref_sunsky.npz"))


def load_build(name):
    from tyrant_amd import scenes

    g = np.load(os.path.join(GOLDEN, f"ref_build_{name}.npz"))
    tris = np.ascontiguousarray(g["triangles"]).view(scenes.TRIANGLE_DTYPE).reshape(-1)
    bb = np.ascontiguousarray(g["bboxes"]).view(scenes.BBOX_DTYPE).reshape(-1)
    return tris, bb, g["nodes"], g["prims"]


def big_scene(name):
    from tyrant_amd import scenes

    return {"mesh706": lambda: scenes.mesh_scene(706), "glass2236": lambda: scenes.glass_dof_scene(2236)}[name]()


# ---- a17 / a18: the builder ---------------------------------------------------------------------------------------


@pytest.mark.parametrize("name", SMALL)
def test_oracle_builder_equals_reference_builder(orc, hashes, name):
    """bvh.cpp:3-225 run by the reference itself vs oracle/orc_bvh.c: every byte of every node, the defined bytes of every triangle"""
    tris, bb, nodes_ref, prims_ref = load_build(name)
    nodes, prims = orc.bvh_build(tris, bb)
    assert nodes.shape[0] == hashes[name]["nodes"] == nodes_ref.shape[0]
    assert nodes.tobytes() == nodes_ref.tobytes()
    assert defined_triangle_bytes(prims) == np.ascontiguousarray(prims_ref[:, :37]).tobytes()
    assert sha(nodes.tobytes()) == hashes[name]["nodes_sha256"] and sha(defined_triangle_bytes(prims)) == hashes[name]["prims_sha256"]


@pytest.mark.parametrize("threads", [1, 8])
@pytest.mark.parametrize("name", SMALL)
def test_product_builder_equals_reference_builder(hip, name, threads):
    """tyr_bvh_build (host/bvh_build.cpp, serial and task-parallel) vs the reference's own output"""
    tris, bb, nodes_ref, prims_ref = load_build(name)
    hip.set_build_threads(threads)
    try:
        nodes, prims = hip.bvh_build(tris, bb)
    finally:
        hip.set_build_threads(0)
    assert nodes.tobytes() == nodes_ref.tobytes()
    assert defined_triangle_bytes(prims) == np.ascontiguousarray(prims_ref[:, :37]).tobytes()


def test_scene_generators_still_produce_the_fixture_inputs(hashes):
    """the hashes of the big scenes are only meaningful while the generators emit the triangles they were made from"""
    from tyrant_amd import scenes

    for name, sc in (("cornell36", scenes.cornell_box()), ("soup10k", scenes.cornell_soup(10000)), ("mesh64", scenes.mesh_scene(64)), ("mesh706", scenes.mesh_scene(706))):
        assert sc.triangles.shape[0] == hashes[name]["triangles"]
        assert sha(defined_triangle_bytes(sc.triangles)) == hashes[name]["input_sha256"], name
        if name in SMALL:
            assert load_build(name)[0].tobytes() == sc.triangles.tobytes()


def test_c3_tree_of_oracle_and_product_equals_the_reference_tree(orc, hip, hashes):
    """the 1.1 M-node tree of the benchmarked scene (C3): SHA-256 of what the reference's builder emitted for it"""
    from tyrant_amd import scenes

    sc = big_scene("mesh706")
    bb = scenes.triangle_bboxes(sc.triangles)
    want = hashes["mesh706"]
    for build in (orc.bvh_build, hip.bvh_build):
        nodes, prims = build(sc.triangles, bb)
        assert nodes.shape[0] == want["nodes"]
        assert sha(nodes.tobytes()) == want["nodes_sha256"]
        assert sha(defined_triangle_bytes(prims)) == want["prims_sha256"]


def test_fixtures_rederive_from_the_reference_when_it_is_present(ref, hashes):
    """authoring container only (skipped where oracle/_ref is absent or older): the committed fixtures are what the
    reference's builder emits now"""
    if not hasattr(ref, "ref_bvh_build"):
        pytest.skip("oracle/_ref predates the builder exports")
    from tyrant_amd import scenes

    for name in SMALL:
        tris, bb, nodes_ref, prims_ref = load_build(name)
        prims = np.ascontiguousarray(tris.copy())
        nodes = np.zeros(2 * tris.shape[0] - 1, dtype=scenes.NODE_DTYPE)
        nn = ref.ref_bvh_build(prims.ctypes.data, tris.shape[0], bb.ctypes.data, nodes.ctypes.data, 2)
        assert nn == nodes_ref.shape[0] and nodes[:nn].tobytes() == nodes_ref.tobytes()
        assert defined_triangle_bytes(prims) == np.ascontiguousarray(prims_ref[:, :37]).tobytes()


def test_bbox_union_matches_reference(ref, orc):
    """Bbox.cpp:3-14 (fmin / fmax per component) against the oracle's union as the builder uses it"""
    if not hasattr(ref, "ref_bbox_union"):
        pytest.skip("oracle/_ref predates the builder exports")
    rng = np.random.default_rng(3)
    a = rng.normal(size=(4096, 2, 3)).astype(np.float32)
    b = rng.normal(size=(4096, 2, 3)).astype(np.float32)
    a[:8, 0] = 1e10  # Bbox.h:4 initial bounds
    a[:8, 1] = -1e10
    out = np.zeros_like(a)
    ref.ref_bbox_union(a.ctypes.data, b.ctypes.data, a.shape[0], out.ctypes.data)
    want = np.stack([np.minimum(a[:, 0], b[:, 0]), np.maximum(a[:, 1], b[:, 1])], axis=1)
    assert out.tobytes() == want.tobytes()


def test_equal_counts_is_not_byte_pinned(hashes):
    """PartitionAlgorithm::EqualCounts (bvh.cpp:113-120) partitions with std::nth_element, whose permutation is
    implementation-defined (MSVC's STL in the original, libstdc++ in oracle/_ref, a hand-written selection here): the
    oracle and the product agree with each other (tests/test_host_and_abi.py) and are NOT claimed to reproduce either
    library's bytes.  The fixture records libstdc++'s result so that the difference stays visible."""
    assert "_equalcounts_cornell36_libstdcxx" in hashes


# ---- a13 / a14: the atmosphere --------------------------------------------------------------------------------------


def check_atmosphere(got: np.ndarray, want: np.ndarray, what: str):
    assert not np.isnan(got).any() and np.isfinite(got).all(), what
    u = ulp_distance(got, want)
    assert u.max() <= SUNSKY_MAX_ULP, f"{what}: worst {u.max()} ulp"
    assert (u == 0).mean() >= SUNSKY_MIN_BIT_EQUAL, f"{what}: only {(u == 0).mean():.4f} bit-equal"


@pytest.mark.parametrize("k", [0, 1, 2])
def test_oracle_sun_setup_equals_reference(orc, sunsky_gold, k):
    """kernel.cu:683-709 + the hoisted SunIntensity / totalMie: bit for bit.  (Sun position 1 tells a binary32
    fromSpherical, which the reference has, from a binary64 one by one ulp in every component.)"""
    S = orc.sun_setup(tuple(sunsky_gold[f"sun_position{k}"]))
    st = sunsky_gold[f"setup{k}"]
    assert np.array(S.sunDirection[:], dtype=np.float32).tobytes() == st[:3].tobytes()
    assert np.float32(S.sunAngularDiameterCos) == st[3]
    assert np.float32(S.sunE) == st[7]
    assert np.array(S.mieAtX[:], dtype=np.float32).tobytes() == sunsky_gold["mie_at_x"].tobytes()


@pytest.mark.parametrize("k", [0, 1, 2])
def test_product_sun_setup_equals_reference(hip, orc, sunsky_gold, k):
    """host/sun_setup.cpp (no GPU needed) against the reference's values and, field by field, the oracle's"""
    sp = tuple(float(x) for x in sunsky_gold[f"sun_position{k}"])
    P = hip.sun_setup(*sp)
    st = sunsky_gold[f"setup{k}"]
    assert P["sunDirection"].tobytes() == st[:3].tobytes() and P["sunAngularDiameterCos"] == st[3] and P["sunE"] == st[7]
    assert P["mieAtX"].tobytes() == sunsky_gold["mie_at_x"].tobytes()
    S = orc.sun_setup(sp)
    for name, n in hip.SUN_PARAM_FIELDS:
        o = np.array(getattr(S, name)[:], dtype=np.float32) if n > 1 else np.float32(getattr(S, name))
        assert np.asarray(P[name]).tobytes() == np.asarray(o).tobytes(), name


@pytest.mark.parametrize("k", [0, 1, 2])
def test_oracle_atmosphere_within_stated_bound_of_reference(orc, sunsky_gold, k):
    """sun / sky / sunsky (sunsky.cu:32-161) over 10 k directions per sun position, 2 k of them inside 2 degrees of the sun"""
    S = orc.sun_setup(tuple(sunsky_gold[f"sun_position{k}"]))
    dirs = sunsky_gold[f"dirs{k}"]
    assert dirs.shape[0] >= 10000
    for name in ("sun", "sky", "sunsky"):
        got = np.stack([getattr(orc, name)(S, d) for d in dirs])
        check_atmosphere(got, sunsky_gold[f"{name}{k}"], f"oracle {name}, sun position {k}")


@pytest.mark.parametrize("k", [0, 1, 2])
def test_oracle_cone_samples_follow_reference(orc, sunsky_gold, k):
    """getConeSample (sunsky.cu:170-185) along four xorshift streams: the stream itself bit-exact, the directions within 2 ulp"""
    import ctypes as C

    S = orc.sun_setup(tuple(sunsky_gold[f"sun_position{k}"]))
    L = orc.lib()
    for si, seed in enumerate(sunsky_gold[f"cone_seeds{k}"]):
        st = C.c_uint32(int(seed))
        got = np.zeros((64, 3), dtype=np.float32)
        for i in range(64):
            o = (C.c_float * 3)()
            L.orc_cone_sample(C.byref(S), C.byref(st), o)
            got[i] = o[:]
        assert st.value == int(sunsky_gold[f"cone_seed_after{k}"][si])
        assert ulp_distance(got, sunsky_gold[f"cone{k}"][si]).max() <= 2


def test_oracle_scalar_helpers_follow_reference(orc, sunsky_gold):
    """SunIntensity (sunsky.cu:24-26) enters through sunE; RayleighPhase / hgPhase (10-12, 20-22) are inlined in the
    oracle's atmosphere: checked here through sky() being within bound, and directly for SunIntensity at the setups."""
    for k in range(3):
        S = orc.sun_setup(tuple(sunsky_gold[f"sun_position{k}"]))
        assert np.float32(S.sunE) == sunsky_gold[f"setup{k}"][7]
    h = sunsky_gold["helpers"]
    assert np.isfinite(h).all() and (h[:, 0] > 0).all() and (h[:, 1] > 0).all() and (h[:, 2] >= 0).all()


def test_sunsky_fixture_rederives_from_the_reference_when_it_is_present(ref, sunsky_gold):
    import ctypes as C

    if not hasattr(ref, "ref_atmosphere"):
        pytest.skip("oracle/_ref predates the atmosphere exports")
    for k in range(3):
        setup = (C.c_float * 8)()
        ref.ref_sun_setup((C.c_float * 2)(*[float(x) for x in sunsky_gold[f"sun_position{k}"]]), setup)
        assert np.array(setup[:], dtype=np.float32).tobytes() == sunsky_gold[f"setup{k}"].tobytes()
        dirs = np.ascontiguousarray(sunsky_gold[f"dirs{k}"])
        for which, name in enumerate(("sun", "sky", "sunsky")):
            out = np.zeros_like(dirs)
            assert ref.ref_atmosphere(which, dirs.ctypes.data, dirs.shape[0], out.ctypes.data) == 0
            assert out.tobytes() == sunsky_gold[f"{name}{k}"].tobytes()


# ---- on the GPU box ---------------------------------------------------------------------------------------------------


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mesh706", "glass2236"])
def test_product_builder_on_this_host_equals_reference_tree(hip, hashes, name):
    """C3 (1.1 M nodes) and C5 (12.6 M nodes) built by tyr_bvh_build with the GPU box's host threads: the reference
    builder's bytes, by SHA-256"""
    from tyrant_amd import scenes

    sc = big_scene(name)
    want = hashes[name]
    assert sc.triangles.shape[0] == want["triangles"] and sha(defined_triangle_bytes(sc.triangles)) == want["input_sha256"]
    hip.set_build_threads(16)
    try:
        nodes, prims = hip.bvh_build(sc.triangles, scenes.triangle_bboxes(sc.triangles))
    finally:
        hip.set_build_threads(0)
    assert nodes.shape[0] == want["nodes"]
    assert sha(nodes.tobytes()) == want["nodes_sha256"]
    assert sha(defined_triangle_bytes(prims)) == want["prims_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 1, 2])
def test_device_atmosphere_within_stated_bound_of_reference(hip, orc, sunsky_gold, k):
    """hip/sunsky.hpp as k_shade evaluates it, on the device: within the stated bound of the reference's sunsky.cu and
    bit-identical to the oracle (the numeric contract)"""
    sp = tuple(float(x) for x in sunsky_gold[f"sun_position{k}"])
    S = orc.sun_setup(sp)
    dirs = sunsky_gold[f"dirs{k}"]
    for which, name in enumerate(("sun", "sky", "sunsky")):
        got = hip.sunsky_probe(which, sp, dirs)
        check_atmosphere(got, sunsky_gold[f"{name}{k}"], f"device {name}, sun position {k}")
        want = np.stack([getattr(orc, name)(S, d) for d in dirs])
        assert got.tobytes() == want.tobytes(), f"device {name} != oracle"


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 1, 2])
def test_device_cone_samples_follow_reference(hip, sunsky_gold, k):
    sp = tuple(float(x) for x in sunsky_gold[f"sun_position{k}"])
    for si, seed in enumerate(sunsky_gold[f"cone_seeds{k}"]):
        got, after = hip.cone_probe(sp, int(seed), 64)
        assert after == int(sunsky_gold[f"cone_seed_after{k}"][si])
        assert ulp_distance(got, sunsky_gold[f"cone{k}"][si]).max() <= 2
