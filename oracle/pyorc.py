"""ctypes bindings of the CPU oracle (oracle/_build/liborc.so) and of the
reference-header harness (oracle/_ref/libref_traverse.so).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module; nothing under tyrant_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liborc.so")
REF_PATH = os.path.join(_HERE, "_ref", "libref_traverse.so")

c_f = C.c_float
c_u32 = C.c_uint32
c_u64 = C.c_uint64
c_i = C.c_int
P = C.c_void_p


class SunParams(C.Structure):
    _fields_ = [
        ("sunDirection", c_f * 3),
        ("sunAngularDiameterCos", c_f),
        ("sunE", c_f),
        ("rayleighAtX", c_f * 3),
        ("mieAtX", c_f * 3),
        ("totalLightAtX", c_f * 3),
        ("mixFactor", c_f),
        ("coneDir", c_f * 3),
        ("coneO1", c_f * 3),
        ("coneO2", c_f * 3),
        ("coneExtent", c_f),
    ]


class CameraC(C.Structure):
    _fields_ = [("position", c_f * 3), ("direction", c_f * 3), ("up", c_f * 3), ("focalDistance", c_f), ("lensRadius", c_f)]


class InputState(C.Structure):
    _fields_ = [("key_w", C.c_uint8), ("key_s", C.c_uint8), ("key_a", C.c_uint8), ("key_d", C.c_uint8), ("key_space", C.c_uint8), ("key_left_control", C.c_uint8), ("key_left_shift", C.c_uint8),
                ("key_left_alt", C.c_uint8), ("cursor_x", C.c_double), ("cursor_y", C.c_double), ("window_w", C.c_int32), ("window_h", C.c_int32)]


class CameraPose(C.Structure):
    _fields_ = [("position", c_f * 3), ("direction", c_f * 3), ("up", c_f * 3), ("horizontal_angle", C.c_double), ("vertical_angle", C.c_double)]


class Counters(C.Structure):
    _fields_ = [
        ("primary_ray_cnt", c_u32),
        ("start_position", c_u32),
        ("shadow_ray_cnt", c_u32),
        ("n_live", c_u32),
        ("frame", c_u32),
        ("pad_", c_u32),
        ("budget_remaining", c_u64),
        ("total_extend_rays", c_u64),
        ("total_shadow_rays", c_u64),
        ("total_primary_rays", c_u64),
        ("nodes_extend", c_u64),
        ("tris_extend", c_u64),
        ("nodes_connect", c_u64),
        ("tris_connect", c_u64),
        ("n_survive", c_u64),
        ("n_shadow_visible", c_u64),
        ("rays_in_tree_extend", c_u64),
        ("rays_in_tree_connect", c_u64),
    ]

    def asdict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build(ref: bool = True) -> None:
    """compile the oracle (and the reference harness when /root/reference exists)"""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    if ref and os.path.isdir("/root/reference/PathTracer"):
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(ref=False)
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(c_f)
        up = C.POINTER(c_u32)
        L.orc_random_int.restype = c_u32
        L.orc_random_int.argtypes = [up]
        L.orc_random_float.restype = c_f
        L.orc_random_float.argtypes = [up]
        L.orc_random_float2.restype = c_f
        L.orc_random_float2.argtypes = [up]
        L.orc_random_int_between_0_and_max.restype = c_i
        L.orc_random_int_between_0_and_max.argtypes = [up, c_i]
        L.orc_random_2d_stratified_sample.argtypes = [up, fp]
        L.orc_concentric_sample_disk.argtypes = [fp, fp]
        L.orc_orthonormal_basis_naive.argtypes = [fp, fp, fp]
        for name in ("orc_dm_sinf", "orc_dm_cosf", "orc_dm_expf"):
            getattr(L, name).restype = c_f
            getattr(L, name).argtypes = [c_f]
        L.orc_dm_powf.restype = c_f
        L.orc_dm_powf.argtypes = [c_f, c_f]
        L.orc_sun_setup.argtypes = [fp, C.POINTER(SunParams)]
        for name in ("orc_sun", "orc_sky", "orc_sunsky"):
            getattr(L, name).argtypes = [C.POINTER(SunParams), fp, fp]
        L.orc_cone_sample.argtypes = [C.POINTER(SunParams), up, fp]
        L.orc_bvh_build.restype = c_i
        L.orc_bvh_build.argtypes = [P, c_i, P, P, c_i]
        L.orc_triangle_bbox.argtypes = [P, P]
        L.orc_triangle_intersect.restype = c_f
        L.orc_triangle_intersect.argtypes = [P, fp, fp]
        L.orc_bbox_intersect.restype = c_i
        L.orc_bbox_intersect.argtypes = [P, fp, fp, C.POINTER(c_i), c_f]
        L.orc_bvh_intersect.restype = c_i
        L.orc_bvh_intersect.argtypes = [P, P, P, C.POINTER(c_u64)]
        L.orc_bvh_intersect_batch.argtypes = [P, P, P, c_i, P]
        L.orc_bvh_intersect_simple.restype = c_i
        L.orc_bvh_intersect_simple.argtypes = [P, P, P, c_f, C.POINTER(c_u64)]
        L.orc_sphere_intersect.restype = c_f
        L.orc_sphere_intersect.argtypes = [P, fp, fp]
        L.orc_create.restype = P
        L.orc_create.argtypes = [c_u32, c_u32, c_u32, c_u32, c_u32, c_u32]
        L.orc_destroy.argtypes = [P]
        L.orc_scene_upload.restype = c_i
        L.orc_scene_upload.argtypes = [P, P, c_i, P, c_i]
        L.orc_set_spheres.argtypes = [P, P]
        L.orc_set_triangle_emission.argtypes = [P, fp]
        L.orc_set_triangle_palette.argtypes = [P, P, P]
        L.orc_default_spheres.argtypes = [P]
        L.orc_set_camera.argtypes = [P, C.POINTER(CameraC)]
        L.orc_set_sun_position.argtypes = [P, c_f, c_f]
        L.orc_set_budget.argtypes = [P, c_u64]
        L.orc_launch_kernels.restype = c_i
        L.orc_launch_kernels.argtypes = [P]
        L.orc_render.restype = c_i
        L.orc_render.argtypes = [P, c_u32, c_i]
        L.orc_reset_accum.argtypes = [P]
        L.orc_reset_accum.argtypes = [P]
        L.orc_get_counters.argtypes = [P, C.POINTER(Counters)]
        L.orc_blit_buffer.restype = P
        L.orc_blit_buffer.argtypes = [P]
        L.orc_resolve.argtypes = [P, P]
        L.orc_ray_queue.restype = P
        L.orc_ray_queue.argtypes = [P, c_i]
        L.orc_shadow_queue.restype = P
        L.orc_shadow_queue.argtypes = [P]
        L.orc_sun_params.restype = C.POINTER(SunParams)
        L.orc_sun_params.argtypes = [P]
        for name in ("orc_stage_begin", "orc_stage_primary", "orc_stage_extend", "orc_stage_extend_debug", "orc_stage_shade", "orc_stage_connect", "orc_stage_end"):
            getattr(L, name).argtypes = [P]
        L.orc_import_work_queue.argtypes = [P, P, c_u32]
        L.orc_bbox_host_ops.argtypes = [P, c_i, P, P]
        L.orc_glm.restype = c_i
        L.orc_glm.argtypes = [c_i, P, P, P, c_i, P]
        L.orc_camera_handle_input.argtypes = [C.POINTER(CameraPose), C.POINTER(InputState), C.c_double]
        L.orc_camera_update.argtypes = [C.POINTER(CameraPose)]
        _lib = L
    return _lib


def ref() -> C.CDLL | None:
    """the reference-header harness, or None when it has not been built (it cannot be built on the GPU box)"""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_PATH):
            return None
        R = C.CDLL(REF_PATH)
        fp = C.POINTER(c_f)
        R.ref_layout.restype = c_i
        R.ref_layout.argtypes = [C.POINTER(c_i), c_i]
        R.ref_constants.restype = c_i
        R.ref_constants.argtypes = [C.POINTER(C.c_double), c_i]
        R.ref_bbox_intersect.restype = c_i
        R.ref_bbox_intersect.argtypes = [P, fp, fp, C.POINTER(c_i), c_f]
        R.ref_triangle_intersect.restype = c_f
        R.ref_triangle_intersect.argtypes = [P, fp, fp]
        R.ref_bbox_host_ops.argtypes = [fp, c_i, P, fp]
        R.ref_bvh_intersect.argtypes = [P, P, P, c_i, C.POINTER(c_i), C.POINTER(c_i)]
        R.ref_bvh_intersect_simple.argtypes = [P, P, P, c_i, C.POINTER(c_i)]
        if hasattr(R, "ref_glm"):
            R.ref_glm.restype = c_i
            R.ref_glm.argtypes = [c_i, P, P, P, c_i, P]
        if hasattr(R, "ref_bvh_build"):  # the reference's bvh.cpp / Bbox.cpp / sunsky.cu (oracle/ref_host_harness.cpp)
            R.ref_bvh_build.restype = c_i
            R.ref_bvh_build.argtypes = [P, c_i, P, P, c_i]
            R.ref_bbox_union.argtypes = [P, P, c_i, P]
            R.ref_sun_setup.argtypes = [fp, fp]
            R.ref_atmosphere.restype = c_i
            R.ref_atmosphere.argtypes = [c_i, P, c_i, P]
            R.ref_sun_helpers.argtypes = [P, c_i, P]
            R.ref_mie_at_x.argtypes = [fp]
            R.ref_cone_samples.argtypes = [C.POINTER(c_u32), c_i, P]
        _ref = R
    return _ref


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(P)


def f3(x):
    return (c_f * 3)(*[float(v) for v in x])


# ---------------------------------------------------------------------------
# convenience wrappers
# ---------------------------------------------------------------------------


def bvh_build(tris: np.ndarray, bboxes: np.ndarray, algo: int = 2):
    """returns (nodes[:nNodes], reordered triangles) -- bvh.cpp:3-25"""
    from tyrant_amd.scenes import NODE_DTYPE  # layouts only

    n = tris.shape[0]
    prims = np.ascontiguousarray(tris.copy())
    nodes = np.zeros(max(2 * n - 1, 1), dtype=NODE_DTYPE)
    bb = np.ascontiguousarray(bboxes)
    nn = lib().orc_bvh_build(_ptr(prims), n, _ptr(bb), _ptr(nodes), algo)
    if nn < 0:
        raise RuntimeError(f"orc_bvh_build failed: {nn}")
    return nodes[:nn].copy(), prims


def sun_setup(sun_position=(0.05, 0.3)) -> SunParams:
    S = SunParams()
    lib().orc_sun_setup((c_f * 2)(*sun_position), C.byref(S))
    return S


def _vec_fn(name, S, d):
    out = (c_f * 3)()
    getattr(lib(), name)(C.byref(S), f3(d), out)
    return np.array(out[:], dtype=np.float32)


def sun(S, d):
    return _vec_fn("orc_sun", S, d)


def sky(S, d):
    return _vec_fn("orc_sky", S, d)


def sunsky(S, d):
    return _vec_fn("orc_sunsky", S, d)


class Oracle:
    """one orc_ctx: the serial wavefront renderer"""

    def __init__(self, width, height, queue_size, rank=0, nranks=1, flags=0):
        self.L = lib()
        self.W, self.H, self.N = width, height, queue_size
        self.h = self.L.orc_create(width, height, queue_size, rank, nranks, flags)
        if not self.h:
            raise ValueError("orc_create rejected the configuration")

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, nodes: np.ndarray, prims: np.ndarray):
        nodes = np.ascontiguousarray(nodes)
        prims = np.ascontiguousarray(prims)
        rc = self.L.orc_scene_upload(self.h, _ptr(nodes), nodes.shape[0], _ptr(prims), prims.shape[0])
        if rc:
            raise RuntimeError("orc_scene_upload failed")

    def set_spheres(self, spheres: np.ndarray):
        s = np.ascontiguousarray(spheres)
        assert s.nbytes == 7 * 44
        self.L.orc_set_spheres(self.h, _ptr(s))

    def set_triangle_emission(self, rgb):
        self.L.orc_set_triangle_emission(self.h, (C.c_float * 3)(*[float(v) for v in rgb]))

    def set_camera(self, cam):
        c = CameraC(f3(cam.position), f3(cam.direction), f3(cam.up), cam.focalDistance, cam.lensRadius)
        self.L.orc_set_camera(self.h, C.byref(c))

    def set_sun_position(self, x, y):
        self.L.orc_set_sun_position(self.h, x, y)

    def set_budget(self, n):
        self.L.orc_set_budget(self.h, n)

    def load_scene(self, scene, nodes, prims):
        self.upload(nodes, prims)
        self.set_spheres(scene.spheres)
        self.set_camera(scene.camera)
        self.set_sun_position(*scene.sun_position)
        self.set_triangle_emission(getattr(scene, "triangle_emission", (3.0, 3.0, 3.0)))
        if getattr(scene, "palette_color", None) is not None:
            self.set_triangle_palette(scene.palette_color, scene.palette_emission)

    def set_triangle_palette(self, color, emission=None):
        col = np.ascontiguousarray(color, dtype=np.float32).reshape(256, 3)
        em = None if emission is None else np.ascontiguousarray(emission, dtype=np.float32).reshape(256, 3)
        self.L.orc_set_triangle_palette(self.h, _ptr(col), None if em is None else _ptr(em))

    def launch_kernels(self):
        return self.L.orc_launch_kernels(self.h)

    def render(self, spp, max_iterations=1 << 30):
        return self.L.orc_render(self.h, spp, max_iterations)

    def reset_accum(self):
        self.L.orc_reset_accum(self.h)

    def stage(self, name):
        getattr(self.L, "orc_stage_" + name)(self.h)

    def counters(self) -> dict:
        k = Counters()
        self.L.orc_get_counters(self.h, C.byref(k))
        return k.asdict()

    def blit_buffer(self) -> np.ndarray:
        p = self.L.orc_blit_buffer(self.h)
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(c_f)), shape=(self.H * self.W, 4))
        return a.copy()

    def resolve(self) -> np.ndarray:
        out = np.zeros((self.H * self.W, 4), dtype=np.float32)
        self.L.orc_resolve(self.h, _ptr(out))
        return out

    def ray_queue(self, which=0, count=None) -> np.ndarray:
        from tyrant_amd.scenes import RAY_DTYPE

        p = self.L.orc_ray_queue(self.h, which)
        n = self.N if count is None else count
        buf = (C.c_char * (60 * n)).from_address(p)
        return np.frombuffer(buf, dtype=RAY_DTYPE, count=n).copy()

    def shadow_queue(self, count=None) -> np.ndarray:
        from tyrant_amd.scenes import SHADOW_DTYPE

        p = self.L.orc_shadow_queue(self.h)
        n = self.N if count is None else count
        buf = (C.c_char * (44 * n)).from_address(p)
        return np.frombuffer(buf, dtype=SHADOW_DTYPE, count=n).copy()

    def import_work_queue(self, rays: np.ndarray, n_survivors: int):
        r = np.ascontiguousarray(rays)
        self.L.orc_import_work_queue(self.h, _ptr(r), n_survivors)

    def sun_params(self) -> SunParams:
        return self.L.orc_sun_params(self.h).contents
