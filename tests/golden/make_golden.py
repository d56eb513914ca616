"""Regenerates tests/golden/ref_traverse_*.npz from the REFERENCE's own traversal code.

Run in the authoring container only (needs /root/reference):
    make -C oracle ref && python tests/golden/make_golden.py

Each fixture holds a small scene (flat node array + builder-ordered triangles), a seeded
ray set and the answers of the reference functions compiled from bvh.h / Bbox.h /
loader.h where they lie (oracle/ref_harness.cpp):
    hit, identifier, distance          CachedBVH::intersect        bvh.h:118-161
    traversals                         CachedBVH::intersect_debug  bvh.h:164-209
    anyhit                             CachedBVH::intersectSimple  bvh.h:213-256
The fixtures are data only: inputs and expected outputs.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pyorc  # noqa: E402
from tyrant_amd import scenes  # noqa: E402


def ray_set(sc, n, seed):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=scenes.RAY_DTYPE)
    o = np.array(sc.camera.position, dtype=np.float32)
    h = n // 2
    rays["origin"][:h] = o
    d = rng.normal(size=(n, 3))
    d[:h] = np.array(sc.camera.direction) * 1.5 + rng.uniform(-0.75, 0.75, size=(h, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["direction"] = d.astype(np.float32)
    rays["origin"][h:] = rng.uniform(-45, 45, size=(n - h, 3)).astype(np.float32) + np.array([0, 0, 50], dtype=np.float32)
    # a few axis-parallel rays: zero direction components exercise the inf / NaN paths of Bbox.h:39-56
    rays["direction"][h : h + 6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float32)
    rays["distance"] = 1e20
    rays["distance"][::7] = rng.uniform(20, 200, size=len(rays["distance"][::7])).astype(np.float32)  # pre-shortened by a sphere hit
    rays["identifier"] = -7
    return rays, rng.uniform(5, 250, size=n).astype(np.float32)


def main():
    R = pyorc.ref()
    assert R is not None, "build oracle/_ref first (make -C oracle ref)"
    ip = C.POINTER(C.c_int)
    for name, sc, n in (("cornell36", scenes.cornell_box(), 2048), ("soup2k", scenes.cornell_soup(2000), 4096), ("mesh32", scenes.mesh_scene(32), 4096)):
        nodes, prims = pyorc.bvh_build(sc.triangles, scenes.triangle_bboxes(sc.triangles))
        rays, closest = ray_set(sc, n, 20261003)
        out = rays.copy()
        hit = np.zeros(n, dtype=np.int32)
        trav = np.zeros(n, dtype=np.int32)
        R.ref_bvh_intersect(nodes.ctypes.data, prims.ctypes.data, out.ctypes.data, n, hit.ctypes.data_as(ip), trav.ctypes.data_as(ip))
        sh = np.zeros(n, dtype=scenes.SHADOW_DTYPE)
        sh["origin"], sh["direction"], sh["closestDistance"] = rays["origin"], rays["direction"], closest
        anyhit = np.zeros(n, dtype=np.int32)
        R.ref_bvh_intersect_simple(nodes.ctypes.data, prims.ctypes.data, sh.ctypes.data, n, anyhit.ctypes.data_as(ip))
        path = os.path.join(ROOT, "tests", "golden", f"ref_traverse_{name}.npz")
        np.savez_compressed(
            path,
            nodes=nodes.view(np.uint8).reshape(-1, 32),
            prims=prims.view(np.uint8).reshape(-1, 40),
            origin=rays["origin"],
            direction=rays["direction"],
            distance_in=rays["distance"],
            closest=closest,
            hit=hit,
            identifier=out["identifier"],
            distance=out["distance"],
            traversals=trav,
            anyhit=anyhit,
        )
        print(path, os.path.getsize(path), "bytes; hit rate", hit.mean(), "anyhit rate", anyhit.mean())


if __name__ == "__main__":
    main()
