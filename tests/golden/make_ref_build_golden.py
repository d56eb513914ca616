"""Regenerates tests/golden/ref_build_*.npz, ref_build_hashes.json and ref_sunsky.npz from the REFERENCE's own
host builder and atmosphere.

Run in the authoring container only (needs /root/reference):
    make -C oracle ref && python tests/golden/make_ref_build_golden.py

What runs is the reference's bvh.cpp / Bbox.cpp / sunsky.cu, compiled unmodified where they lie (oracle/Makefile `ref`,
oracle/ref_host_harness.cpp).  The fixtures are data only -- inputs and expected outputs:

  ref_build_<scene>.npz     input triangles + boxes, the node array BVH::BVH(SAH) produced (all 32 bytes per node:
                            vector::resize value-initialises them, bvh.cpp:11) and the reordered triangle array
                            (bvh.cpp:24) -- cornell36, soup10k (C2), mesh64
  ref_build_hashes.json     node count + SHA-256 of the node bytes and of the reordered triangles' defined bytes
                            (0..36 of every 40: vert, e1, e2, materialType) for mesh706 (C3) and glass2236 (C5),
                            whose arrays are too large to commit; also for the three small scenes
  ref_sunsky.npz            3 sun positions x >= 10 k view directions (2 k of them inside 2 degrees of the sun):
                            sun / sky / sunsky (sunsky.cu:32-161), the setup values of kernel.cu:683-709, the scalar
                            helpers (sunsky.cu:10-26) and getConeSample streams (sunsky.cu:170-185)
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pyorc  # noqa: E402
from tyrant_amd import scenes  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SUN_POSITIONS = ((0.05, 0.3), (0.3, 0.12), (0.62, 0.45))  # variables.cpp:3 default; a high sun; a sun near the horizon


def defined_triangle_bytes(prims: np.ndarray) -> bytes:
    """bytes 0..36 of every 40-byte Triangle (37-39 are padding, indeterminate in the reference)"""
    return np.ascontiguousarray(prims.view(np.uint8).reshape(-1, 40)[:, :37]).tobytes()


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def ref_build(tris: np.ndarray, algo: int = 2):
    R = pyorc.ref()
    n = tris.shape[0]
    prims = np.ascontiguousarray(tris.copy())
    bb = np.ascontiguousarray(scenes.triangle_bboxes(tris))
    nodes = np.zeros(max(2 * n - 1, 1), dtype=scenes.NODE_DTYPE)
    nn = R.ref_bvh_build(prims.ctypes.data, n, bb.ctypes.data, nodes.ctypes.data, algo)
    return nodes[:nn].copy(), prims, bb


def view_directions(sun_dir: np.ndarray, seed: int, n_sphere=8000, n_near=2000, n_axis=16) -> np.ndarray:
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n_sphere, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # inside 2 degrees of the sun: the smoothstep edge of sunsky.cu:156-157 and the disk term of :70
    s = sun_dir.astype(np.float64)
    a = np.cross(s, [0.0, 0.0, 1.0])
    a /= np.linalg.norm(a)
    b = np.cross(s, a)
    ang = np.radians(rng.uniform(0.0, 2.0, size=n_near))
    phi = rng.uniform(0, 2 * np.pi, size=n_near)
    near = np.cos(ang)[:, None] * s + np.sin(ang)[:, None] * (np.cos(phi)[:, None] * a + np.sin(phi)[:, None] * b)
    axes = np.array(
        [[0, 0, 1], [0, 0, -1], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [1, 1, 1], [1, 1, 0], [0.6, 0, 0.8], [0, 0.6, -0.8], [0, 0, 1e-3], [1, 0, 1e-6], [-1, 1, 0], [0.1, 0.2, 0.97], [0.5, 0.5, 0.7071], [0, 1, 1]],
        dtype=np.float64,
    )[:n_axis]
    axes[6:] /= np.linalg.norm(axes[6:], axis=1, keepdims=True)
    out = np.concatenate([d, near, axes, s[None, :]]).astype(np.float32)
    # what kernel.cu hands sun/sky/sunsky is glm::normalize'd in float32; renormalise in float32 the same way numpy can
    return out


def make_sunsky():
    R = pyorc.ref()
    data = {}
    for k, sp in enumerate(SUN_POSITIONS):
        setup = (C.c_float * 8)()
        R.ref_sun_setup((C.c_float * 2)(*sp), setup)
        setup = np.array(setup[:], dtype=np.float32)
        dirs = view_directions(setup[:3], 1000 + k)
        n = dirs.shape[0]
        data[f"sun_position{k}"] = np.array(sp, dtype=np.float32)
        data[f"setup{k}"] = setup
        data[f"dirs{k}"] = dirs
        for which, name in enumerate(("sun", "sky", "sunsky")):
            out = np.zeros((n, 3), dtype=np.float32)
            assert R.ref_atmosphere(which, dirs.ctypes.data, n, out.ctypes.data) == 0
            data[f"{name}{k}"] = out
        seeds = np.array([0x9E3779B9, 1, 12345, 0xDEADBEEF], dtype=np.uint32)
        cones, after = [], []
        for s in seeds:
            st = C.c_uint32(int(s))
            o = np.zeros((64, 3), dtype=np.float32)
            R.ref_cone_samples(C.byref(st), 64, o.ctypes.data)
            cones.append(o)
            after.append(st.value)
        data[f"cone_seeds{k}"] = seeds
        data[f"cone{k}"] = np.stack(cones)
        data[f"cone_seed_after{k}"] = np.array(after, dtype=np.uint32)
    x = np.concatenate([np.linspace(-1, 1, 1001), np.random.default_rng(7).uniform(-1, 1, 1000)]).astype(np.float32)
    h = np.zeros((x.shape[0], 4), dtype=np.float32)
    R.ref_sun_helpers(x.ctypes.data, x.shape[0], h.ctypes.data)
    mie = (C.c_float * 3)()
    R.ref_mie_at_x(mie)
    data["helper_x"] = x
    data["helpers"] = h[:, :3].copy()
    data["mie_at_x"] = np.array(mie[:], dtype=np.float32)
    path = os.path.join(GOLD, "ref_sunsky.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), "bytes;", sum(data[f"dirs{k}"].shape[0] for k in range(len(SUN_POSITIONS))), "directions")


def make_builds():
    hashes = {}
    small = (("cornell36", scenes.cornell_box()), ("soup10k", scenes.cornell_soup(10000)), ("mesh64", scenes.mesh_scene(64)))
    big = (("mesh706", lambda: scenes.mesh_scene(706)), ("glass2236", lambda: scenes.glass_dof_scene(2236)))
    for name, sc in small:
        nodes, prims, bb = ref_build(sc.triangles)
        path = os.path.join(GOLD, f"ref_build_{name}.npz")
        np.savez_compressed(
            path,
            triangles=sc.triangles.view(np.uint8).reshape(-1, 40),
            bboxes=bb.view(np.uint8).reshape(-1, 24),
            nodes=nodes.view(np.uint8).reshape(-1, 32),
            prims=prims.view(np.uint8).reshape(-1, 40),
        )
        hashes[name] = {"triangles": int(sc.triangles.shape[0]), "nodes": int(nodes.shape[0]), "nodes_sha256": sha(nodes.tobytes()), "prims_sha256": sha(defined_triangle_bytes(prims)), "input_sha256": sha(defined_triangle_bytes(sc.triangles))}
        print(path, os.path.getsize(path), "bytes;", hashes[name])
    for name, mk in big:
        sc = mk()
        nodes, prims, _ = ref_build(sc.triangles)
        hashes[name] = {"triangles": int(sc.triangles.shape[0]), "nodes": int(nodes.shape[0]), "nodes_sha256": sha(nodes.tobytes()), "prims_sha256": sha(defined_triangle_bytes(prims)), "input_sha256": sha(defined_triangle_bytes(sc.triangles))}
        print(name, hashes[name])
        del sc, nodes, prims
    # EqualCounts (bvh.cpp:113-120) goes through std::nth_element, whose permutation is implementation-defined: recorded for
    # the record only (libstdc++ 11 here, MSVC in the original); tests do not pin it
    nodes, prims, _ = ref_build(scenes.cornell_box().triangles, algo=1)
    hashes["_equalcounts_cornell36_libstdcxx"] = {"nodes": int(nodes.shape[0]), "nodes_sha256": sha(nodes.tobytes())}
    with open(os.path.join(GOLD, "ref_build_hashes.json"), "w") as f:
        json.dump(hashes, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    assert pyorc.ref() is not None and hasattr(pyorc.ref(), "ref_bvh_build"), "build oracle/_ref first (make -C oracle ref)"
    make_sunsky()
    make_builds()
