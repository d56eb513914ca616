"""The vector arithmetic every stage is built from, pinned to the reference's vendored glm itself
(Dependencies/glm-0.9.9.3, run through oracle/_ref -> tests/golden/ref_glm.npz) and the BBox host operations pinned to
the reference's Bbox.h (-> tests/golden/ref_bbox_ops.npz):

  CPU   the oracle's v3* helpers (oracle/orc_internal.h, via orc_glm) and its BBox ops reproduce the fixtures bit for
        bit; when oracle/_ref is present (authoring container) the fixtures are re-derived from the live library too.
  GPU   hip/vecmath.hpp evaluated on the device (tyr_vecmath_probe) reproduces the same fixtures bit for bit.

pow and exp go through <cmath> inside glm; the path's versions are the deterministic polynomial layer (DESIGN.md
"Numeric contract" item 2), so those two ops are held to <= 1 ulp of glm's answers instead of equality -- except
pow(x, 0.5), which the path evaluates as sqrt(x) and which must agree exactly on the path's range of arguments."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits

N_OPS = 20
NAMES = ["dot", "cross", "normalize", "length", "reflect", "min", "max", "clamp", "mix", "smoothstep", "pow", "vec/scalar", "vec*scalar", "scalar*vec", "exp", "vec*vec", "vec/vec", "-vec", "vec+vec",
         "vec-vec"]
APPROX = {10, 14}  # <cmath> inside glm vs the deterministic layer


@pytest.fixture(scope="module")
def glm():
    return np.load(os.path.join(GOLDEN, "ref_glm.npz"))


def ulp_distance(x, y):
    a, b = bits(x).astype(np.int64), bits(y).astype(np.int64)
    a = np.where(a < 0x80000000, a, 0x80000000 - a)
    b = np.where(b < 0x80000000, b, 0x80000000 - b)
    return np.abs(a - b)


def check(op, got, z):
    want = z[f"out{op}"]
    if op in APPROX:
        d = ulp_distance(got, want)
        assert d.max() <= 1, f"{NAMES[op]}: {d.max()} ulp from glm"
        if op == 10:
            # pow(x, 0.5) is evaluated as sqrt(x) (correctly rounded); glibc's powf is not correctly rounded everywhere (one
            # of the 1536 samples, x = 7.5e14, comes out 1 ulp high), so equality is demanded on the path's range of x
            half = (z["b10"] == np.float32(0.5)) & (z["a10"] < np.float32(1e6))
            assert half.sum() > 1000 and np.array_equal(bits(got[half]), bits(want[half])), "pow(x, 0.5) evaluated as sqrt(x) must equal glm::pow on [1e-6, 1e6]"
    else:
        bad = np.count_nonzero(np.any(bits(got) != bits(want), axis=-1))
        assert bad == 0, f"{NAMES[op]} differs from glm in {bad} of {len(want)} elements"


@pytest.mark.parametrize("op", range(N_OPS))
def test_oracle_helpers_equal_glm(orc, glm, op):
    L = orc.lib()
    a, b, c = (np.ascontiguousarray(glm[f"{k}{op}"]) for k in "abc")
    out = np.zeros_like(a)
    assert L.orc_glm(op, a.ctypes.data, b.ctypes.data, c.ctypes.data, a.shape[0], out.ctypes.data) == 0
    check(op, out, glm)


def test_fixture_is_what_the_live_glm_says(ref, glm):
    """authoring container only: the committed answers are the vendored library's, byte for byte"""
    if not hasattr(ref, "ref_glm"):
        pytest.skip("oracle/_ref predates ref_glm: make -C oracle ref")
    for op in range(N_OPS):
        a, b, c = (np.ascontiguousarray(glm[f"{k}{op}"]) for k in "abc")
        out = np.zeros_like(a)
        assert ref.ref_glm(op, a.ctypes.data, b.ctypes.data, c.ctypes.data, a.shape[0], out.ctypes.data) == 0
        assert np.array_equal(bits(out), bits(glm[f"out{op}"])), NAMES[op]


def test_bbox_host_ops_equal_reference(orc, hip):
    """BBox::addVertex / surfaceArea / largestExtent (Bbox.h:8-36, incl. the ties -> y then z rule): the oracle's, and --
    for the three-vertex cases -- the product's per-triangle boxes (tyr_triangle_bboxes, Scene.cpp:22-33)"""
    from tyrant_amd import scenes

    z = np.load(os.path.join(GOLDEN, "ref_bbox_ops.npz"))
    L = orc.lib()
    for v, k, box, ae in zip(z["verts"], z["counts"], z["boxes"], z["area_extent"]):
        vv = np.ascontiguousarray(v[:k])
        b = np.zeros(6, dtype=np.float32)
        o2 = np.zeros(2, dtype=np.float32)
        L.orc_bbox_host_ops(vv.ctypes.data, int(k), b.ctypes.data, o2.ctypes.data)
        assert np.array_equal(bits(b), bits(box)) and np.array_equal(bits(o2), bits(ae)), (vv, b, box, o2, ae)
    tri = z["counts"] == 3
    t = scenes.make_triangles(z["verts"][tri, 0], z["verts"][tri, 1], z["verts"][tri, 2])
    # Triangle stores e1 = v1 - v0: keep the cases whose vertices survive the round trip vert + e exactly
    exact = np.all(t["vert"] + t["e1"] == z["verts"][tri, 1], axis=1) & np.all(t["vert"] + t["e2"] == z["verts"][tri, 2], axis=1)
    assert exact.sum() >= 10
    bb = hip.triangle_bboxes(t)
    assert np.array_equal(bits(bb["bounds"].reshape(-1, 6)[exact]), bits(z["boxes"][tri][exact]))


def test_ref_bbox_fixture_is_live(ref):
    z = np.load(os.path.join(GOLDEN, "ref_bbox_ops.npz"))
    import ctypes as C

    for v, k, box, ae in list(zip(z["verts"], z["counts"], z["boxes"], z["area_extent"]))[::7]:
        vv = np.ascontiguousarray(v[:k])
        b = np.zeros(6, dtype=np.float32)
        o2 = np.zeros(2, dtype=np.float32)
        ref.ref_bbox_host_ops(vv.ctypes.data_as(C.POINTER(C.c_float)), int(k), b.ctypes.data, o2.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(bits(b), bits(box)) and np.array_equal(bits(o2), bits(ae))


@pytest.mark.gpu
@pytest.mark.parametrize("op", range(N_OPS))
def test_device_vecmath_equals_glm(hip, glm, op):
    """hip/vecmath.hpp on the MI355X against the vendored glm's answers"""
    got = hip.vecmath_probe(op, glm[f"a{op}"], glm[f"b{op}"], glm[f"c{op}"])
    check(op, got, glm)
