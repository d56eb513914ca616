"""Regenerates tests/golden/ref_glm.npz from the REFERENCE's vendored glm (Dependencies/glm-0.9.9.3) and
tests/golden/ref_bbox_ops.npz from its BBox host operations (Bbox.h:8-36), both through oracle/_ref (ref_harness.cpp).

Run in the authoring container only (needs /root/reference):
    make -C oracle ref && python tests/golden/make_glm_golden.py

Data only: seeded inputs and the library's outputs per op code (oracle/ref_harness.cpp ref_glm)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pyorc  # noqa: E402

N_OPS = 20


def inputs(n=1024, seed=20261004):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, 3)).astype(np.float32) * np.float32(50.0)
    b = rng.normal(size=(n, 3)).astype(np.float32)
    c = rng.uniform(-2.0, 2.0, size=(n, 3)).astype(np.float32)
    # ray-like data: unit directions, scene-scale positions, tiny and huge magnitudes, signed zeros, equal operands
    a[:256] = rng.normal(size=(256, 3)).astype(np.float32)
    a[:256] /= np.linalg.norm(a[:256], axis=1, keepdims=True).astype(np.float32)
    b[:256] = rng.normal(size=(256, 3)).astype(np.float32)
    b[:256] /= np.linalg.norm(b[:256], axis=1, keepdims=True).astype(np.float32)
    a[256:320] *= np.float32(1e-18)
    a[320:384] *= np.float32(1e15)
    a[384:392] = np.array([[0.0, 0.0, 1.0], [0.0, -0.0, 1.0], [1.0, 0.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [3.0, 4.0, 0.0], [1e-3, 1e-3, 1e-3], [50.0, -50.0, 100.0]], dtype=np.float32)
    b[392:456] = a[392:456]  # min / max / clamp ties
    return a, b, c


def inputs_for(op, a, b, c):
    """per-op domains (positive bases for pow, ordered bounds for clamp / smoothstep, non-zero divisors)"""
    a, b, c = a.copy(), b.copy(), c.copy()
    if op == 7:  # clamp(a, lo, hi): lo <= hi
        lo, hi = np.minimum(b[:, 0], b[:, 1]), np.maximum(b[:, 0], b[:, 1])
        b[:, 0], b[:, 1] = lo, hi
        a = (a / np.float32(25.0)).astype(np.float32)
    if op == 8:
        c[:, 0] = np.abs(c[:, 0]) / np.float32(2.0)  # mix factor in [0, 1]
    if op == 9:  # smoothstep(e0, e1, x): e0 < e1, x around the edge interval (sunsky.cu:156: the sun's disk edge)
        c[:, 1] = c[:, 0] + np.abs(c[:, 1]) * np.float32(1e-3) + np.float32(2e-5)
        a[:, 0] = c[:, 0] + (a[:, 0] / np.float32(50.0)) * (c[:, 1] - c[:, 0])
    if op == 10:  # pow(a, b): positive bases; exponents the path uses (0.5: sunsky.cu:66) and general ones
        a = np.abs(a) / np.float32(40.0) + np.float32(1e-6)
        b = np.abs(b) * np.float32(3.0)
        b[::2] = np.float32(0.5)
    if op in (11, 16):
        c[np.abs(c) < 1e-3] = np.float32(0.5)
        b[np.abs(b) < 1e-3] = np.float32(0.5)
    if op == 14:
        a = (a / np.float32(10.0)).astype(np.float32)
    return a, b, c


def main():
    R = pyorc.ref()
    assert R is not None and hasattr(R, "ref_glm"), "build oracle/_ref first (make -C oracle ref)"
    a, b, c = inputs()
    out = {}
    for op in range(N_OPS):
        aa, bb, cc = (np.ascontiguousarray(x) for x in inputs_for(op, a, b, c))
        o = np.zeros_like(aa)
        assert R.ref_glm(op, aa.ctypes.data, bb.ctypes.data, cc.ctypes.data, aa.shape[0], o.ctypes.data) == 0
        out[f"a{op}"], out[f"b{op}"], out[f"c{op}"], out[f"out{op}"] = aa, bb, cc, o
    path = os.path.join(ROOT, "tests", "golden", "ref_glm.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")

    # BBox host ops (Bbox.h:8-36): addVertex over k vertices, surfaceArea, largestExtent (ties -> y, then z)
    rng = np.random.default_rng(7)
    cases = []
    for k in (1, 2, 3, 5, 8):
        for _ in range(200):
            cases.append(rng.normal(size=(k, 3)).astype(np.float32) * np.float32(rng.choice([1e-3, 1.0, 50.0, 1e6])))
    # ties of the largest extent and degenerate (flat / point) boxes
    cases += [np.array([[0, 0, 0], [2, 2, 2]], dtype=np.float32), np.array([[0, 0, 0], [2, 2, 1]], dtype=np.float32), np.array([[0, 0, 0], [1, 2, 2]], dtype=np.float32),
              np.array([[0, 0, 0], [2, 1, 2]], dtype=np.float32), np.array([[1, 1, 1]], dtype=np.float32), np.array([[0, 0, 0], [0, 3, 0]], dtype=np.float32)]
    verts = np.zeros((len(cases), 8, 3), dtype=np.float32)
    counts = np.zeros(len(cases), dtype=np.int32)
    boxes = np.zeros((len(cases), 6), dtype=np.float32)
    res = np.zeros((len(cases), 2), dtype=np.float32)
    for i, v in enumerate(cases):
        v = np.ascontiguousarray(v)
        verts[i, : len(v)] = v
        counts[i] = len(v)
        R.ref_bbox_host_ops(v.ctypes.data_as(pyorc.C.POINTER(pyorc.c_f)), len(v), boxes[i].ctypes.data, res[i].ctypes.data_as(pyorc.C.POINTER(pyorc.c_f)))
    path = os.path.join(ROOT, "tests", "golden", "ref_bbox_ops.npz")
    np.savez_compressed(path, verts=verts, counts=counts, boxes=boxes, area_extent=res)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
