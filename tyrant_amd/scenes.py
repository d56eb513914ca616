"""Seeded synthetic scenes for the configurations of BASELINE.json / SURVEY.md section 8d.

The reference loads one mesh through assimp (Scene.cpp:3-47) and a hard-wired
sphere table (kernel.cu:674-680); its scene file is absent.  BASELINE.json asks
for synthetic triangle scenes instead; this module builds them as arrays in
the reference's own record layouts (loader.h:13-19 Triangle = {vert, e1, e2,
materialType}, kernel.cu:77-81 Sphere) so they can be handed to the C ABI, to the
host BVH builder and to the test oracle alike.

World conventions: z-up, triangles are one-sided and wound so
that e1 x e2 faces the side they are seen from (loader.h:28 culls back faces),
scenes are O(100) units because epsilon = 1e-3 is absolute (variables.h:14).

Randomness is a counter-based integer hash (murmur3 finaliser) so every scene is
a pure function of (index, seed), vectorised, and reproducible anywhere.
"""
from __future__ import annotations

import dataclasses

import numpy as np

# loader.h:13-19 (40 bytes)
TRIANGLE_DTYPE = np.dtype(
    [("vert", "<f4", (3,)), ("e1", "<f4", (3,)), ("e2", "<f4", (3,)), ("materialType", "u1"), ("pad_", "u1", (3,))]
)
# kernel.cu:77-81 (44 bytes)
SPHERE_DTYPE = np.dtype(
    [("radius", "<f4"), ("position", "<f4", (3,)), ("color", "<f4", (3,)), ("emmission", "<f4", (3,)), ("refl", "<i4")]
)
# bvh.h:55-68 (32 bytes)
NODE_DTYPE = np.dtype(
    [("bounds", "<f4", (2, 3)), ("offset", "<i4"), ("primitiveCount", "<u2"), ("splitAxis", "u1"), ("pad", "u1")]
)
# Bbox.h:3-5 (24 bytes)
BBOX_DTYPE = np.dtype([("bounds", "<f4", (2, 3))])
# variables.h:24-34 (60 bytes)
RAY_DTYPE = np.dtype(
    [
        ("origin", "<f4", (3,)),
        ("direction", "<f4", (3,)),
        ("direct", "<f4", (3,)),
        ("distance", "<f4"),
        ("identifier", "<i4"),
        ("bounces", "<i4"),
        ("index", "<i4"),
        ("geometry_type", "<i4"),
        ("lastSpecular", "u1"),
        ("pad_", "u1", (3,)),
    ]
)
# variables.h:36-42 (44 bytes)
SHADOW_DTYPE = np.dtype(
    [("origin", "<f4", (3,)), ("direction", "<f4", (3,)), ("color", "<f4", (3,)), ("buffer_index", "<i4"), ("closestDistance", "<f4")]
)
assert TRIANGLE_DTYPE.itemsize == 40 and SPHERE_DTYPE.itemsize == 44 and NODE_DTYPE.itemsize == 32
assert RAY_DTYPE.itemsize == 60 and SHADOW_DTYPE.itemsize == 44 and BBOX_DTYPE.itemsize == 24

# kernel.cu:67-71 enum Refl_t
DIFF, SPEC, REFR, PHONG, LIGHT = 0, 1, 2, 3, 4


@dataclasses.dataclass
class Camera:
    """camera.h:3-9: the fields the render path reads."""

    position: tuple = (1.0, 30.0, 90.0)
    direction: tuple = (1.0, 0.0, 0.0)
    up: tuple = (0.0, 0.0, 1.0)
    focalDistance: float = 1.0
    lensRadius: float = 0.0


@dataclasses.dataclass
class SceneData:
    name: str
    triangles: np.ndarray  # TRIANGLE_DTYPE
    spheres: np.ndarray  # SPHERE_DTYPE[7]
    camera: Camera
    sun_position: tuple = (0.05, 0.3)  # variables.cpp:3
    triangle_materials: bool = False  # extension flag
    light_list: bool = False  # extension flag: LIGHT triangles emit and are sampled by NEE
    triangle_emission: tuple = (3.0, 3.0, 3.0)  # with light_list; the reference light's value (kernel.cu:680)
    triangle_colors: bool = False  # extension flag: colour / emission per triangle, palette index = Triangle byte 37
    palette_color: np.ndarray | None = None  # float32[256][3]
    palette_emission: np.ndarray | None = None  # float32[256][3]


def hash_u32(index: np.ndarray, seed: int) -> np.ndarray:
    """murmur3 fmix32 of (index * golden + seed); uint32 in, uint32 out."""
    x = (index.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def hash_unit(index: np.ndarray, seed: int) -> np.ndarray:
    """uniform [0,1) float64 from the hash"""
    return hash_u32(index, seed).astype(np.float64) / 4294967296.0


def make_triangles(v0: np.ndarray, v1: np.ndarray, v2: np.ndarray, material: np.ndarray | int = DIFF) -> np.ndarray:
    """Scene.cpp:39-45: Triangle{vert = v0, e1 = v1 - v0, e2 = v2 - v0} in float32."""
    v0 = np.asarray(v0, dtype=np.float32).reshape(-1, 3)
    v1 = np.asarray(v1, dtype=np.float32).reshape(-1, 3)
    v2 = np.asarray(v2, dtype=np.float32).reshape(-1, 3)
    t = np.zeros(v0.shape[0], dtype=TRIANGLE_DTYPE)
    t["vert"] = v0
    t["e1"] = v1 - v0
    t["e2"] = v2 - v0
    t["materialType"] = material
    return t


def _quad(p0, p1, p2, p3, facing) -> np.ndarray:
    """two triangles (p0,p1,p2), (p0,p2,p3); checks that e1 x e2 faces `facing`."""
    p = [np.asarray(q, dtype=np.float64) for q in (p0, p1, p2, p3)]
    n = np.cross(p[1] - p[0], p[2] - p[0])
    assert np.dot(n, np.asarray(facing, dtype=np.float64)) > 0, "quad winding faces away"
    return make_triangles([p[0], p[0]], [p[1], p[2]], [p[2], p[3]])


def _box(cx, cy, hx, hy, z0, z1, angle) -> np.ndarray:
    """rotated box with outward-facing sides, 12 triangles"""
    c, s = np.cos(angle), np.sin(angle)

    def P(lx, ly, z):
        return (cx + c * lx - s * ly, cy + s * lx + c * ly, z)

    def N(lx, ly, lz):
        return (c * lx - s * ly, s * lx + c * ly, lz)

    a, b, cc, d = (-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy)
    quads = [
        _quad(P(*a, z1), P(*b, z1), P(*cc, z1), P(*d, z1), N(0, 0, 1)),  # top
        _quad(P(*a, z0), P(*d, z0), P(*cc, z0), P(*b, z0), N(0, 0, -1)),  # bottom
        _quad(P(*a, z0), P(*b, z0), P(*b, z1), P(*a, z1), N(0, -1, 0)),  # -y side
        _quad(P(*b, z0), P(*cc, z0), P(*cc, z1), P(*b, z1), N(1, 0, 0)),  # +x side
        _quad(P(*cc, z0), P(*d, z0), P(*d, z1), P(*cc, z1), N(0, 1, 0)),  # +y side
        _quad(P(*d, z0), P(*a, z0), P(*a, z1), P(*d, z1), N(-1, 0, 0)),  # -x side
    ]
    return np.concatenate(quads)


def room_walls() -> np.ndarray:
    """5 inward-facing walls of the room [-50,50] x [-50,50] x [0,100], open toward -y: 10 triangles."""
    L, H = 50.0, 100.0
    return np.concatenate(
        [
            _quad((-L, -L, 0), (L, -L, 0), (L, L, 0), (-L, L, 0), (0, 0, 1)),  # floor
            _quad((-L, -L, H), (-L, L, H), (L, L, H), (L, -L, H), (0, 0, -1)),  # ceiling
            _quad((-L, L, 0), (L, L, 0), (L, L, H), (-L, L, H), (0, -1, 0)),  # back wall
            _quad((-L, -L, 0), (-L, L, 0), (-L, L, H), (-L, -L, H), (1, 0, 0)),  # left wall
            _quad((L, L, 0), (L, -L, 0), (L, -L, H), (L, L, H), (-1, 0, 0)),  # right wall
        ]
    )


def cornell_spheres(light_z: float = 86.0) -> np.ndarray:
    """7-entry sphere table (kernel.cu:14,123): light in slot 6 (kernel.cu:421), the
    ground sphere of kernel.cu:678 in slot 4, everything else parked out of reach."""
    s = np.zeros(7, dtype=SPHERE_DTYPE)
    for i in range(7):
        s[i] = (1.0, (0.0, 5000.0 + 100.0 * i, -5000.0), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), DIFF)
    s[4] = (1e4, (0.0, 0.0, -1e4 - 20.0), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), DIFF)
    s[6] = (9.0, (0.0, 0.0, light_z), (0.0, 1.0, 0.0), (3.0, 3.0, 3.0), LIGHT)
    return s


def reference_spheres() -> np.ndarray:
    """the reference's hard-wired table, kernel.cu:674-680"""
    s = np.zeros(7, dtype=SPHERE_DTYPE)
    s[0] = (16.5, (0, 40, 16.5), (1, 1, 1), (0, 0, 0), DIFF)
    s[1] = (16.5, (40, 0, 16.5), (0.5, 0.5, 0.06), (0, 0, 0), REFR)
    s[2] = (16.5, (-40, -50, 36.5), (0.6, 0.5, 0.4), (0, 0, 0), PHONG)
    s[3] = (16.5, (-40, -50, 16.5), (0.6, 0.5, 0.4), (0, 0, 0), SPEC)
    s[4] = (1e4, (0, 0, -1e4 - 20), (1, 1, 1), (0, 0, 0), DIFF)
    s[5] = (20, (0, -80, 20), (1.0, 0.0, 0.0), (0, 0, 0), DIFF)
    s[6] = (9, (0, -80, 120.0), (0.0, 1.0, 0.0), (3, 3, 3), LIGHT)
    return s


CORNELL_CAMERA = Camera(position=(0.0, -190.0, 50.0), direction=(0.0, 1.0, 0.0), up=(0.0, 0.0, 1.0), focalDistance=1.0, lensRadius=0.0)


def cornell_box() -> SceneData:
    """C1: 36 triangles = 5 walls (10) + short box (12) + tall box (12) + ceiling patch (2)."""
    tris = np.concatenate(
        [
            room_walls(),
            _box(18.0, -12.0, 15.0, 15.0, 0.0, 30.0, -0.3),
            _box(-16.0, 14.0, 15.0, 15.0, 0.0, 60.0, 0.3),
            _quad((-12, -12, 99.5), (-12, 12, 99.5), (12, 12, 99.5), (12, -12, 99.5), (0, 0, -1)),
        ]
    )
    assert tris.shape[0] == 36
    return SceneData("cornell36", tris, cornell_spheres(), CORNELL_CAMERA)


def cornell_area_light() -> SceneData:
    """Cornell box lit by emissive triangles (extension): the ceiling patch and a small panel on the
    left wall are materialType LIGHT, the short box is a mirror, the tall box Phong; spheres[6] stays a light too, so
    next-event estimation picks among three emitters of two kinds."""
    short = _box(18.0, -12.0, 15.0, 15.0, 0.0, 30.0, -0.3)
    short["materialType"] = SPEC
    tall = _box(-16.0, 14.0, 15.0, 15.0, 0.0, 60.0, 0.3)
    tall["materialType"] = PHONG
    patch = _quad((-12, -12, 99.5), (-12, 12, 99.5), (12, 12, 99.5), (12, -12, 99.5), (0, 0, -1))
    patch["materialType"] = LIGHT
    panel = make_triangles([(-49.5, -10.0, 40.0)], [(-49.5, 10.0, 40.0)], [(-49.5, 0.0, 60.0)], LIGHT)
    assert np.cross(panel["e1"][0], panel["e2"][0])[0] > 0  # faces +x, into the room
    tris = np.concatenate([room_walls(), short, tall, patch, panel])
    return SceneData("cornell_area_light", tris, cornell_spheres(light_z=70.0), CORNELL_CAMERA, triangle_materials=True, light_list=True, triangle_emission=(4.0, 3.5, 3.0))


def cornell_colored() -> SceneData:
    """The classic coloured Cornell box (extension, SURVEY.md 8f-3: the reference's commented-out `tempTriangle.color`,
    Scene.cpp:44): red left wall, green right wall, a blue glass short box, a gold Phong tall box, a grey mirror strip
    on the back wall, and two emissive triangle panels of DIFFERENT emission (warm ceiling patch, cold wall panel) next
    to spheres[6].  Colour / emission index = the first padding byte of the 40-byte Triangle record."""
    walls = room_walls()
    walls["pad_"][:, 0] = 0
    walls["pad_"][6:8, 0] = 1  # left wall
    walls["pad_"][8:10, 0] = 2  # right wall
    short = _box(18.0, -12.0, 15.0, 15.0, 0.0, 30.0, -0.3)
    short["materialType"] = REFR
    short["pad_"][:, 0] = 3
    tall = _box(-16.0, 14.0, 15.0, 15.0, 0.0, 60.0, 0.3)
    tall["materialType"] = PHONG
    tall["pad_"][:, 0] = 4
    strip = _quad((-30, 49.5, 55), (30, 49.5, 55), (30, 49.5, 85), (-30, 49.5, 85), (0, -1, 0))
    strip["materialType"] = SPEC
    strip["pad_"][:, 0] = 5
    patch = _quad((-12, -12, 99.5), (-12, 12, 99.5), (12, 12, 99.5), (12, -12, 99.5), (0, 0, -1))
    patch["materialType"] = LIGHT
    patch["pad_"][:, 0] = 6
    panel = make_triangles([(-49.5, -10.0, 40.0)], [(-49.5, 10.0, 40.0)], [(-49.5, 0.0, 60.0)], LIGHT)
    panel["pad_"][:, 0] = 7
    tris = np.concatenate([walls, short, tall, strip, patch, panel])
    col = np.ones((256, 3), dtype=np.float32)
    em = np.full((256, 3), 3.0, dtype=np.float32)
    col[0] = (0.73, 0.73, 0.73)
    col[1] = (0.65, 0.05, 0.05)
    col[2] = (0.12, 0.45, 0.15)
    col[3] = (0.02, 0.015, 0.004)  # REFR: Beer-Lambert absorption per unit length (kernel.cu:511-513)
    col[4] = (0.9, 0.7, 0.3)
    col[5] = (0.8, 0.8, 0.8)
    em[6] = (6.0, 5.0, 3.5)
    em[7] = (1.5, 2.5, 5.0)
    return SceneData("cornell_colored", tris, cornell_spheres(light_z=70.0), CORNELL_CAMERA, triangle_materials=True, light_list=True, triangle_colors=True,
                     palette_color=col, palette_emission=em)


def random_soup(n: int, seed: int = 12345, lo=(-48.0, -48.0, 2.0), hi=(48.0, 48.0, 98.0), edge: float = 1.5) -> np.ndarray:
    """n triangles: vert uniform in [lo,hi], e1,e2 uniform in [-edge,edge]^3."""
    i = np.arange(n, dtype=np.uint64)
    u = np.stack([hash_unit(i * np.uint64(9) + np.uint64(k), seed) for k in range(9)], axis=1)
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    v0 = lo + u[:, 0:3] * (hi - lo)
    e1 = (u[:, 3:6] * 2.0 - 1.0) * edge
    e2 = (u[:, 6:9] * 2.0 - 1.0) * edge
    t = np.zeros(n, dtype=TRIANGLE_DTYPE)
    t["vert"] = v0.astype(np.float32)
    t["e1"] = e1.astype(np.float32)
    t["e2"] = e2.astype(np.float32)
    return t


def cornell_soup(n: int = 10000, seed: int = 12345) -> SceneData:
    """C2: Cornell box + n seeded random diffuse triangles in the room interior."""
    base = cornell_box()
    tris = np.concatenate([base.triangles, random_soup(n, seed)])
    return SceneData(f"cornell36+soup{n}", tris, base.spheres, base.camera)


def heightfield(cells: int, seed: int = 12345, spec_fraction: float = 0.3, refr_fraction: float = 0.0) -> np.ndarray:
    """2*cells^2 up-facing triangles of a displaced height field over [-50,50]^2.

    materialType = SPEC for `spec_fraction`, REFR for `refr_fraction` of the triangle ids (by hash), DIFF otherwise."""
    g = cells
    xs = np.linspace(-50.0, 50.0, g + 1)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    vid = (np.arange((g + 1) * (g + 1), dtype=np.uint64)).reshape(g + 1, g + 1)
    Z = 22.0 + 9.0 * np.sin(X * 0.15) * np.cos(Y * 0.13) + 3.0 * np.sin(X * 0.7 + Y * 0.45) + 0.6 * (hash_unit(vid, seed) - 0.5)
    P = np.stack([X, Y, Z], axis=-1).astype(np.float32)
    p00 = P[:-1, :-1].reshape(-1, 3)
    p10 = P[:-1, 1:].reshape(-1, 3)  # +x
    p11 = P[1:, 1:].reshape(-1, 3)
    p01 = P[1:, :-1].reshape(-1, 3)  # +y
    v0 = np.empty((2 * g * g, 3), dtype=np.float32)
    v1 = np.empty_like(v0)
    v2 = np.empty_like(v0)
    v0[0::2], v1[0::2], v2[0::2] = p00, p10, p11
    v0[1::2], v1[1::2], v2[1::2] = p00, p11, p01
    tid = np.arange(2 * g * g, dtype=np.uint64)
    u = hash_unit(tid, seed ^ 0x5BD1E995)
    mat = np.full(2 * g * g, DIFF, dtype=np.uint8)
    mat[u < spec_fraction] = SPEC
    mat[(u >= spec_fraction) & (u < spec_fraction + refr_fraction)] = REFR
    return make_triangles(v0, v1, v2, mat)


def mesh_scene(cells: int = 706, seed: int = 12345, spec_fraction: float = 0.3, refr_fraction: float = 0.0) -> SceneData:
    """C3 (cells=706: 996,872 + 10 triangles), C5 (cells=2236, refr_fraction=0.05)."""
    tris = np.concatenate([room_walls(), heightfield(cells, seed, spec_fraction, refr_fraction)])
    return SceneData(f"room+heightfield{cells}", tris, cornell_spheres(), CORNELL_CAMERA, triangle_materials=True)


# The view from which the room's opening fills a 16:9 frame.  kernel.cu:698-699 scales camera_right by 1.5 * W / H and camera_up by
# 1.5, and kernel.cu:274-278 spans them over [-0.5, 0.5]: at distance d along the view direction the frame is 2.667 d wide and 1.5 d
# high.  From SURVEY.md 8d's Cornell camera (0, -190, 50) the 100 x 100 opening at y = -50 (d = 140) covers 0.268 x 0.476 = 12.7 % of a
# 1080p frame -- 61 % of a C3 render's extend rays never enter the tree.  From d = 37.5 the frame is exactly as wide as the opening
# (and 56 high, inside its 100): every camera ray enters the room (but for the leftmost pixel column, whose jitter reaches a pixel further).
FRAMED_CAMERA = Camera(position=(0.0, -87.5, 50.0), direction=(0.0, 1.0, 0.0), up=(0.0, 0.0, 1.0), focalDistance=1.0, lensRadius=0.0)


def mesh_scene_framed(cells: int = 706, seed: int = 12345) -> SceneData:
    """bench.py's secondary workload `c3_framed`: the C3 scene (same triangles, same tree) seen from FRAMED_CAMERA."""
    sc = mesh_scene(cells, seed)
    return dataclasses.replace(sc, name=sc.name + "_framed", camera=FRAMED_CAMERA)


def glass_dof_scene(cells: int = 2236, seed: int = 12345) -> SceneData:
    """C5: the C3 generator with 5 % REFR triangles, a thin lens (lensRadius 0.5) focused on the
    height field from the Cornell camera (kernel.cu:286-293 multiplies focalDistance by 3), and a high sun."""
    sc = mesh_scene(cells, seed, spec_fraction=0.3, refr_fraction=0.05)
    cam = Camera(position=CORNELL_CAMERA.position, direction=CORNELL_CAMERA.direction, up=CORNELL_CAMERA.up, focalDistance=60.0, lensRadius=0.5)
    return SceneData(f"glass+dof{cells}", sc.triangles, sc.spheres, cam, sun_position=(0.3, 0.2), triangle_materials=True)  # sun (0.25,-0.77,0.59): shines in through the open side


def tyrant_default(cells: int = 24, seed: int = 7) -> SceneData:
    """The reference's own sphere table (all five materials) over a small height-field
    placed near the spheres: exercises every branch of shade (kernel.cu:404-597)."""
    hf = heightfield(cells, seed, spec_fraction=0.0)
    hf["vert"] = hf["vert"] * np.float32(1.6) + np.array([0.0, 0.0, -50.0], dtype=np.float32)
    hf["e1"] = hf["e1"] * np.float32(1.6)
    hf["e2"] = hf["e2"] * np.float32(1.6)
    cam = Camera(position=(0.0, -250.0, 95.0), direction=(0.0, 0.962964, -0.26962993), up=(0.0, 0.0, 1.0))
    return SceneData("tyrant_default", hf, reference_spheres(), cam)


def triangle_bboxes(tris: np.ndarray) -> np.ndarray:
    """Scene.cpp:29-33: per-face BBox from the three vertices (float32 min/max)."""
    v0 = tris["vert"]
    v1 = tris["vert"] + tris["e1"]
    v2 = tris["vert"] + tris["e2"]
    b = np.zeros(tris.shape[0], dtype=BBOX_DTYPE)
    b["bounds"][:, 0, :] = np.minimum(np.minimum(v0, v1), v2)
    b["bounds"][:, 1, :] = np.maximum(np.maximum(v0, v1), v2)
    return b
