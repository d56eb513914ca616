"""ctypes binding of tyrant_amd/lib/libtyrant_hip.so (the C ABI of include/tyr_c.h).

This is glue for the tests and bench.py; the product is the shared library.  There is
no fallback: if the library is missing or no HIP device is present, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import scenes

_HERE = os.path.dirname(os.path.abspath(__file__))
# TYRANT_HIP_LIBRARY: load another build of the same ABI (diagnostic builds made by tools/*.sh); never a CPU path
LIB_PATH = os.environ.get("TYRANT_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libtyrant_hip.so")

TYR_FLAG_TRIANGLE_MATERIALS = 1
TYR_FLAG_PROFILE = 2
TYR_FLAG_COUNT_VISITS = 4
TYR_FLAG_LIGHT_LIST = 8
TYR_FLAG_TRIANGLE_COLORS = 16
TYR_FLAG_DEBUG_BVH = 32
TYR_ERR_NO_DEVICE = -2
TYR_ERR_UNSUPPORTED = -7
TYR_DIST_GATHER, TYR_DIST_REDUCE = 0, 1
TYR_DIST_ID_BYTES = 128
KERNEL_NAMES = ("primary", "extend", "shade", "connect", "resolve")

c_f, c_u32, c_u64, c_i32, P = C.c_float, C.c_uint32, C.c_uint64, C.c_int32, C.c_void_p


class TyrError(RuntimeError):
    def __init__(self, status: int, what: str):
        self.status = status
        super().__init__(f"{what}: status {status} ({status_string(status)})")


class Config(C.Structure):
    _fields_ = [
        ("width", c_u32),
        ("height", c_u32),
        ("queue_size", c_u32),
        ("device", c_i32),
        ("rank", c_u32),
        ("nranks", c_u32),
        ("flags", c_u32),
        ("stream", P),
    ]


class CameraC(C.Structure):
    _fields_ = [("position", c_f * 3), ("direction", c_f * 3), ("up", c_f * 3), ("focalDistance", c_f), ("lensRadius", c_f)]


class Counters(C.Structure):
    _fields_ = [
        ("primary_ray_cnt", c_u32),
        ("start_position", c_u32),
        ("shadow_ray_cnt", c_u32),
        ("n_live", c_u32),
        ("frame", c_u32),
        ("device_error", c_u32),
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
        ("debug", c_u64 * 16),
    ]

    def asdict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "debug"}
        d["debug"] = [int(x) for x in self.debug]
        return d


class SceneInfo(C.Structure):
    _fields_ = [("n_prims", c_u32), ("n_pair_nodes", c_u32), ("n_quad_nodes", c_u32), ("n_staged_nodes", c_u32), ("n_lights", c_u32), ("max_quad_nodes", c_u32), ("max_prim_offset", c_u32),
                ("quad_max_stack", c_u32), ("device_bytes", c_u64), ("upload_layout_s", C.c_double), ("upload_copy_s", C.c_double), ("layout_on_device", c_u32), ("reserved_", c_u32)]


class LayoutStats(C.Structure):
    _fields_ = [("n_pair_nodes", c_u32), ("n_quad_nodes", c_u32), ("n_staged_nodes", c_u32), ("quad_max_stack", c_u32), ("root_ref", c_u32), ("quad_root_ref", c_u32),
                ("hash_pairs", c_u64), ("hash_quads", c_u64), ("hash_tris", c_u64), ("seconds", C.c_double)]


class Timings(C.Structure):
    _fields_ = [("ms", C.c_double * 5), ("launches", c_u64 * 5)]


# every symbol include/tyr_c.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "tyr_status_string": (C.c_char_p, [C.c_int]),
    "tyr_abi_version": (C.c_int, []),
    "tyr_create": (C.c_int, [C.POINTER(P), C.POINTER(Config)]),
    "tyr_destroy": (C.c_int, [P]),
    "tyr_scene_upload": (C.c_int, [P, P, c_i32, P, c_i32]),
    "tyr_set_spheres": (C.c_int, [P, P]),
    "tyr_set_triangle_emission": (C.c_int, [P, P]),
    "tyr_set_triangle_palette": (C.c_int, [P, P, P]),
    "tyr_set_camera": (C.c_int, [P, C.POINTER(CameraC)]),
    "tyr_set_sun_position": (C.c_int, [P, c_f, c_f]),
    "tyr_set_blit_buffer": (C.c_int, [P, P]),
    "tyr_get_blit_buffer": (P, [P]),
    "tyr_launch_kernels": (C.c_int, [P]),
    "tyr_set_budget": (C.c_int, [P, c_u64]),
    "tyr_set_frame": (C.c_int, [P, c_u32]),
    "tyr_get_counters": (C.c_int, [P, C.POINTER(Counters)]),
    "tyr_render": (C.c_int, [P, c_u32, c_u32, C.POINTER(c_u32)]),
    "tyr_resolve": (C.c_int, [P, P]),
    "tyr_reset_accum": (C.c_int, [P]),
    "tyr_read_accum": (C.c_int, [P, P]),
    "tyr_stage_begin": (C.c_int, [P]),
    "tyr_stage_primary": (C.c_int, [P]),
    "tyr_stage_extend": (C.c_int, [P]),
    "tyr_stage_shade": (C.c_int, [P]),
    "tyr_stage_connect": (C.c_int, [P]),
    "tyr_stage_end": (C.c_int, [P]),
    "tyr_sync": (C.c_int, [P]),
    "tyr_queue_export": (C.c_int, [P, C.c_int, P, c_u32]),
    "tyr_queue_import": (C.c_int, [P, P, c_u32]),
    "tyr_queue_rank_check": (C.c_int, [P, C.c_int, P, P]),
    "tyr_shadow_export": (C.c_int, [P, P, c_u32]),
    "tyr_shadow_import": (C.c_int, [P, P, c_u32]),
    "tyr_get_scene_info": (C.c_int, [P, C.POINTER(SceneInfo)]),
    "tyr_layout_probe": (C.c_int, [P, c_i32, P, c_i32, c_i32, C.POINTER(LayoutStats)]),
    "tyr_scene_hash": (C.c_int, [P, C.POINTER(LayoutStats)]),
    "tyr_vecmath_probe": (C.c_int, [c_i32, c_i32, P, P, P, c_u32, P]),
    "tyr_sunsky_probe": (C.c_int, [c_i32, C.c_float, C.c_float, c_i32, P, c_u32, P]),
    "tyr_sun_setup": (C.c_int, [C.c_float, C.c_float, P]),
    "tyr_camera_handle_input": (C.c_int, [P, P, C.c_double]),
    "tyr_get_timings": (C.c_int, [P, C.POINTER(Timings), C.c_int]),
    "tyr_set_tuning": (C.c_int, [P, C.c_int, C.c_int]),
    "tyr_bvh_build": (C.c_int, [P, c_i32, P, P, c_i32]),
    "tyr_bvh_build_device": (C.c_int, [c_i32, P, c_i32, P, P, P]),
    "tyr_scene_build_upload": (C.c_int, [P, P, c_i32, P, P, P, P]),
    "tyr_triangle_bboxes": (C.c_int, [P, c_i32, P]),
    "tyr_set_build_threads": (C.c_int, [c_i32]),
    "tyr_camera_update": (C.c_int, [C.c_double, C.c_double, C.POINTER(c_f)]),
    "tyr_load_ply": (C.c_int, [C.c_char_p, C.POINTER(P)]),
    "tyr_free": (None, [P]),
    "tyr_write_ppm": (C.c_int, [C.c_char_p, P, c_u32, c_u32]),
    "tyr_write_pfm": (C.c_int, [C.c_char_p, P, c_u32, c_u32]),
    "tyr_write_png": (C.c_int, [C.c_char_p, P, c_u32, c_u32]),
    "tyr_default_spheres": (C.c_int, [P]),
    "tyr_dist_unique_id": (C.c_int, [P]),
    "tyr_dist_create": (C.c_int, [C.POINTER(P), P, P, c_i32, c_i32]),
    "tyr_dist_destroy": (C.c_int, [P]),
    "tyr_dist_combine": (C.c_int, [P, c_i32, c_i32, P]),
    "tyr_dist_wait": (C.c_int, [P]),
    "tyr_dist_info": (C.c_int, [P, C.POINTER(c_i32), C.POINTER(c_i32)]),
    "tyr_dist_owned_rows": (C.c_int, [c_u32, c_u32, c_u32, C.POINTER(c_u32), C.POINTER(c_u32)]),
    "tyr_dist_row_owner": (C.c_int, [c_u32, c_u32, C.POINTER(c_u32), C.POINTER(c_u32)]),
    "tyr_dist_pack_rows": (C.c_int, [P, P, c_u32, c_u32, c_u32, c_u32, P]),
    "tyr_dist_scatter_rows": (C.c_int, [P, P, c_u32, c_u32, c_u32, P]),
}

_libs: dict = {}


def lib() -> C.CDLL:
    """Load order matters when PyTorch shares the process: its wheels carry their own HIP runtime in the global symbol
    scope.  Import torch BEFORE the first call of this function (then this library's HIP calls bind to that one runtime);
    the other order leaves two runtimes in the process and the one that initialises second reports no device."""
    path = LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `python __graft_entry__.py build` (there is no fallback path)")
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the ABI and the header disagree
            fn.restype = res
            fn.argtypes = args
        _libs[path] = L
    return _libs[path]


def status_string(status: int) -> str:
    return lib().tyr_status_string(status).decode()


def _check(status: int, what: str):
    if status != 0:
        raise TyrError(status, what)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(P)


# ---- host side of the hot path ---------------------------------------------------------------


def triangle_bboxes(tris: np.ndarray) -> np.ndarray:
    """Scene.cpp:22-33"""
    t = np.ascontiguousarray(tris)
    out = np.zeros(t.shape[0], dtype=scenes.BBOX_DTYPE)
    _check(lib().tyr_triangle_bboxes(_ptr(t), t.shape[0], _ptr(out)), "tyr_triangle_bboxes")
    return out


def bvh_build(tris: np.ndarray, bboxes: np.ndarray | None = None, algo: int = 2):
    """class BVH (bvh.cpp:3-25): returns (nodes[:nNodes], reordered triangles)"""
    prims = np.ascontiguousarray(tris.copy())
    n = prims.shape[0]
    bb = np.ascontiguousarray(bboxes) if bboxes is not None else triangle_bboxes(prims)
    nodes = np.zeros(max(2 * n - 1, 1), dtype=scenes.NODE_DTYPE)
    nn = lib().tyr_bvh_build(_ptr(prims), n, _ptr(bb), _ptr(nodes), algo)
    if nn < 0:
        raise TyrError(nn, "tyr_bvh_build")
    return nodes[:nn].copy(), prims


def bvh_build_device(tris: np.ndarray, bboxes: np.ndarray | None = None, device: int = 0):
    """(nodes, prims, (device_seconds, copy_seconds)): tyr_bvh_build_device -- the SAH build on the GPU, the reference's bytes"""
    prims = np.array(tris, dtype=scenes.TRIANGLE_DTYPE, copy=True)
    bb = triangle_bboxes(prims) if bboxes is None else np.ascontiguousarray(bboxes)  # (tyr_triangle_bboxes: the same boxes bvh_build() takes)
    n = prims.shape[0]
    nodes = np.zeros(max(2 * n - 1, 1), dtype=scenes.NODE_DTYPE)
    sec = (C.c_double * 2)(0.0, 0.0)
    rc = lib().tyr_bvh_build_device(device, _ptr(prims), n, _ptr(bb), _ptr(nodes), sec)
    if rc < 0:
        raise TyrError(rc, "tyr_bvh_build_device")
    return nodes[:rc].copy(), prims, (sec[0], sec[1])


def layout_probe(nodes: np.ndarray, prims: np.ndarray, want_pairs: bool = True) -> dict:
    """the host half of tyr_scene_upload without a device: sizes, FNV-1a hashes of the device arrays, seconds (tyr_layout_probe)"""
    nodes, prims = np.ascontiguousarray(nodes), np.ascontiguousarray(prims)
    st = LayoutStats()
    _check(lib().tyr_layout_probe(_ptr(nodes), nodes.shape[0], _ptr(prims), prims.shape[0], int(want_pairs), C.byref(st)), "tyr_layout_probe")
    return {k: getattr(st, k) for k, _ in st._fields_}


def load_ply(path: str) -> np.ndarray:
    """Scene::Load's import half for a PLY file: TRIANGLE_DTYPE array (Scene.cpp:3-47)"""
    out = P()
    n = lib().tyr_load_ply(os.fsencode(path), C.byref(out))
    if n < 0:
        raise TyrError(n, f"tyr_load_ply({path})")
    try:
        if n == 0:
            return np.zeros(0, dtype=scenes.TRIANGLE_DTYPE)
        buf = (C.c_char * (40 * n)).from_address(out.value)
        return np.frombuffer(buf, dtype=scenes.TRIANGLE_DTYPE, count=n).copy()
    finally:
        lib().tyr_free(out)


def write_image(path: str, rgba: np.ndarray, width: int, height: int):
    """PPM (tonemapped 8-bit) or PFM (float) by extension"""
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    low = path.lower()
    fn = lib().tyr_write_pfm if low.endswith(".pfm") else lib().tyr_write_png if low.endswith(".png") else lib().tyr_write_ppm
    _check(fn(os.fsencode(path), _ptr(a), width, height), "tyr_write_image")


def set_build_threads(threads: int):
    _check(lib().tyr_set_build_threads(threads), "tyr_set_build_threads")


def camera_update(horizontal_angle: float, vertical_angle: float) -> np.ndarray:
    """Camera::update (camera.cpp:46-52)"""
    out = (c_f * 3)()
    _check(lib().tyr_camera_update(horizontal_angle, vertical_angle, out), "tyr_camera_update")
    return np.array(out[:], dtype=np.float32)


def default_spheres() -> np.ndarray:
    s = np.zeros(7, dtype=scenes.SPHERE_DTYPE)
    _check(lib().tyr_default_spheres(_ptr(s)), "tyr_default_spheres")
    return s


# ---- the renderer ------------------------------------------------------------------------------


class Renderer:
    """one tyr_ctx"""

    def __init__(self, width, height, queue_size, device=0, rank=0, nranks=1, flags=0, stream=None, blit_buffer=None):
        self.L = lib()
        self.W, self.H, self.N = width, height, queue_size
        cfg = Config(width, height, queue_size, device, rank, nranks, flags, stream)
        h = P()
        _check(self.L.tyr_create(C.byref(h), C.byref(cfg)), "tyr_create")
        self.h = h
        self._dists = []
        _check(self.L.tyr_set_blit_buffer(self.h, blit_buffer), "tyr_set_blit_buffer")

    def close(self):
        for d in list(getattr(self, "_dists", [])):
            d.close()
        if getattr(self, "h", None):
            self.L.tyr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, nodes: np.ndarray, prims: np.ndarray):
        nodes = np.ascontiguousarray(nodes)
        prims = np.ascontiguousarray(prims)
        _check(self.L.tyr_scene_upload(self.h, _ptr(nodes), nodes.shape[0], _ptr(prims), prims.shape[0]), "tyr_scene_upload")

    def set_spheres(self, spheres: np.ndarray | None):
        if spheres is None:
            _check(self.L.tyr_set_spheres(self.h, None), "tyr_set_spheres")
            return
        s = np.ascontiguousarray(spheres)
        _check(self.L.tyr_set_spheres(self.h, _ptr(s)), "tyr_set_spheres")

    def set_camera(self, cam):
        f3 = lambda x: (c_f * 3)(*[float(v) for v in x])  # noqa: E731
        c = CameraC(f3(cam.position), f3(cam.direction), f3(cam.up), cam.focalDistance, cam.lensRadius)
        _check(self.L.tyr_set_camera(self.h, C.byref(c)), "tyr_set_camera")

    def set_sun_position(self, x, y):
        _check(self.L.tyr_set_sun_position(self.h, x, y), "tyr_set_sun_position")

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
        _check(self.L.tyr_set_triangle_palette(self.h, _ptr(col), None if em is None else _ptr(em)), "tyr_set_triangle_palette")

    def set_triangle_emission(self, rgb):
        _check(self.L.tyr_set_triangle_emission(self.h, (C.c_float * 3)(*[float(v) for v in rgb])), "tyr_set_triangle_emission")

    def set_budget(self, n):
        _check(self.L.tyr_set_budget(self.h, n), "tyr_set_budget")

    def set_frame(self, frame=1):
        """restart the frame counter every seed is built from (kernel.cu:667): the next render repeats the one that began at `frame`"""
        _check(self.L.tyr_set_frame(self.h, frame), "tyr_set_frame")

    def launch_kernels(self):
        _check(self.L.tyr_launch_kernels(self.h), "tyr_launch_kernels")

    def render(self, spp, max_iterations=0xFFFFFFFF) -> int:
        it = c_u32(0)
        _check(self.L.tyr_render(self.h, spp, max_iterations, C.byref(it)), "tyr_render")
        return it.value

    def stage(self, name):
        _check(getattr(self.L, "tyr_stage_" + name)(self.h), "tyr_stage_" + name)

    def counters(self) -> dict:
        k = Counters()
        _check(self.L.tyr_get_counters(self.h, C.byref(k)), "tyr_get_counters")
        return k.asdict()

    def timings(self, reset=False) -> dict:
        t = Timings()
        _check(self.L.tyr_get_timings(self.h, C.byref(t), int(reset)), "tyr_get_timings")
        return {n: {"ms": t.ms[i], "launches": int(t.launches[i])} for i, n in enumerate(KERNEL_NAMES)}

    TUNING_KEYS = {"refill_min_idle": 1, "waves_per_simd": 2, "min_traversing": 4, "ticket_chunk": 5, "static_share": 8, "staged_nodes": 9, "profile_mask": 11, "merge_trace": 12, "static_interleave": 13, "run_ahead": 14, "wide_drain": 15, "fold_spheres": 19, "retire_sky": 20, "resolve_shadows": 21, "wide_block_min_items": 22, "fold_prologue": 23, "layout_on_device": 24, "scan_in_trace": 25, "kernel_snapshot": 26}

    def set_tuning(self, **knobs):
        for name, v in knobs.items():
            if v is not None:
                _check(self.L.tyr_set_tuning(self.h, self.TUNING_KEYS[name], int(v)), "tyr_set_tuning")

    def reset_accum(self):
        _check(self.L.tyr_reset_accum(self.h), "tyr_reset_accum")

    def blit_buffer(self) -> np.ndarray:
        out = np.zeros((self.H * self.W, 4), dtype=np.float32)
        _check(self.L.tyr_read_accum(self.h, _ptr(out)), "tyr_read_accum")
        return out

    def resolve_into(self, device_ptr):
        _check(self.L.tyr_resolve(self.h, device_ptr), "tyr_resolve")

    def queue_rank_check(self, which=1):
        """(records, mismatches): the device's rank tables against the order ray_queue() presents (tyr_queue_rank_check)"""
        n, bad = c_u32(0), c_u32(0)
        _check(self.L.tyr_queue_rank_check(self.h, which, C.byref(n), C.byref(bad)), "tyr_queue_rank_check")
        return n.value, bad.value

    def ray_queue(self, which=0, count=None) -> np.ndarray:
        n = self.N if count is None else count
        out = np.zeros(n, dtype=scenes.RAY_DTYPE)
        _check(self.L.tyr_queue_export(self.h, which, _ptr(out), n), "tyr_queue_export")
        return out

    def shadow_queue(self, count=None) -> np.ndarray:
        n = self.N if count is None else count
        out = np.zeros(n, dtype=scenes.SHADOW_DTYPE)
        _check(self.L.tyr_shadow_export(self.h, _ptr(out), n), "tyr_shadow_export")
        return out

    def import_shadow_queue(self, rays: np.ndarray):
        r = np.ascontiguousarray(rays)
        _check(self.L.tyr_shadow_import(self.h, _ptr(r), r.shape[0]), "tyr_shadow_import")

    def scene_hash(self) -> dict:
        """sizes and FNV-1a hashes of the scene arrays this ctx holds in HBM, read back (tyr_scene_hash): layout_probe()'s figures"""
        st = LayoutStats()
        _check(self.L.tyr_scene_hash(self.h, C.byref(st)), "tyr_scene_hash")
        return {k: getattr(st, k) for k, _ in st._fields_}

    def build_upload(self, tris: np.ndarray, bboxes: np.ndarray | None = None, want_nodes: bool = True):
        """(nodes or None, prims, (build_s, layout_s, copy_s)): tyr_scene_build_upload -- the tree built and laid out on this ctx's device"""
        prims = np.array(tris, dtype=scenes.TRIANGLE_DTYPE, copy=True)
        bb = triangle_bboxes(prims) if bboxes is None else np.ascontiguousarray(bboxes)
        n = prims.shape[0]
        nodes = np.zeros(max(2 * n - 1, 1), dtype=scenes.NODE_DTYPE) if want_nodes else None
        sec = (C.c_double * 3)(0.0, 0.0, 0.0)
        nn = c_i32(0)
        _check(self.L.tyr_scene_build_upload(self.h, _ptr(prims), n, _ptr(bb), None if nodes is None else _ptr(nodes), C.byref(nn), sec), "tyr_scene_build_upload")
        return (None if nodes is None else nodes[: nn.value].copy()), prims, (sec[0], sec[1], sec[2])

    def scene_info(self) -> dict:
        s = SceneInfo()
        _check(self.L.tyr_get_scene_info(self.h, C.byref(s)), "tyr_get_scene_info")
        return {k: (float(getattr(s, k)) if k.endswith("_s") else int(getattr(s, k))) for k, _ in s._fields_}

    def import_work_queue(self, rays: np.ndarray, n_survivors: int):
        r = np.ascontiguousarray(rays)
        _check(self.L.tyr_queue_import(self.h, _ptr(r), n_survivors), "tyr_queue_import")


def vecmath_probe(op: int, a: np.ndarray, b: np.ndarray, c: np.ndarray, device: int = 0) -> np.ndarray:
    """hip/vecmath.hpp function `op` on the device over float3 arrays (op codes: oracle/ref_harness.cpp ref_glm)"""
    a, b, c = (np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3) for x in (a, b, c))
    out = np.zeros_like(a)
    _check(lib().tyr_vecmath_probe(device, op, _ptr(a), _ptr(b), _ptr(c), a.shape[0], _ptr(out)), "tyr_vecmath_probe")
    return out


# ---- multi-GPU combine (RCCL behind the C ABI) --------------------------------------------------


class InputState(C.Structure):
    """tyr_input_state: what Camera::handle_input reads from the GLFW window (camera.cpp:3-44)"""

    _fields_ = [("key_w", C.c_uint8), ("key_s", C.c_uint8), ("key_a", C.c_uint8), ("key_d", C.c_uint8), ("key_space", C.c_uint8), ("key_left_control", C.c_uint8), ("key_left_shift", C.c_uint8),
                ("key_left_alt", C.c_uint8), ("cursor_x", C.c_double), ("cursor_y", C.c_double), ("window_w", c_i32), ("window_h", c_i32)]


class CameraPose(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("direction", C.c_float * 3), ("up", C.c_float * 3), ("horizontal_angle", C.c_double), ("vertical_angle", C.c_double)]


def camera_handle_input(pose: "CameraPose", state: "InputState", delta: float):
    _check(lib().tyr_camera_handle_input(C.byref(pose), C.byref(state), float(delta)), "tyr_camera_handle_input")


SUN_PARAM_FIELDS = (("sunDirection", 3), ("sunAngularDiameterCos", 1), ("sunE", 1), ("rayleighAtX", 3), ("mieAtX", 3), ("totalLightAtX", 3), ("mixFactor", 1), ("coneDir", 3), ("coneO1", 3), ("coneO2", 3), ("coneExtent", 1))


def sun_setup(sun_x: float, sun_y: float) -> dict:
    """the library's per-sun-change constants (host code; no GPU needed)"""
    out = np.zeros(25, dtype=np.float32)
    _check(lib().tyr_sun_setup(sun_x, sun_y, _ptr(out)), "tyr_sun_setup")
    res, k = {}, 0
    for name, n in SUN_PARAM_FIELDS:
        res[name] = out[k : k + n].copy() if n > 1 else out[k]
        k += n
    return res


def sunsky_probe(which: int, sun_position, dirs: np.ndarray, device: int = 0) -> np.ndarray:
    """sun / sky / sunsky (which 0 / 1 / 2) of the device code over float3 directions"""
    d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
    out = np.zeros_like(d)
    _check(lib().tyr_sunsky_probe(device, float(sun_position[0]), float(sun_position[1]), which, _ptr(d), d.shape[0], _ptr(out)), "tyr_sunsky_probe")
    return out


def cone_probe(sun_position, seed: int, n: int, device: int = 0):
    """n sun-cone samples of the device code along one xorshift stream; returns (samples[n][3], seed after)"""
    inp = np.array([seed], dtype=np.uint32).view(np.float32)
    out = np.zeros(3 * n + 1, dtype=np.float32)
    _check(lib().tyr_sunsky_probe(device, float(sun_position[0]), float(sun_position[1]), 3, _ptr(inp), n, _ptr(out)), "tyr_sunsky_probe")
    return out[: 3 * n].reshape(n, 3).copy(), int(out[3 * n : 3 * n + 1].view(np.uint32)[0])


def dist_unique_id() -> bytes:
    """ncclGetUniqueId through the library: 128 opaque bytes that rank 0 hands to the other ranks"""
    buf = (C.c_char * TYR_DIST_ID_BYTES)()
    _check(lib().tyr_dist_unique_id(buf), "tyr_dist_unique_id")
    return bytes(buf)


def dist_owned_rows(height: int, rank: int, nranks: int):
    first, n = c_u32(0), c_u32(0)
    _check(lib().tyr_dist_owned_rows(height, rank, nranks, C.byref(first), C.byref(n)), "tyr_dist_owned_rows")
    return first.value, n.value


def dist_row_owner(y: int, nranks: int):
    r, yl = c_u32(0), c_u32(0)
    _check(lib().tyr_dist_row_owner(y, nranks, C.byref(r), C.byref(yl)), "tyr_dist_row_owner")
    return r.value, yl.value


class Dist:
    """one tyr_dist: the RCCL communicator of a Renderer (same rank / nranks as its pixel shard)"""

    def __init__(self, renderer: "Renderer", unique_id: bytes, rank: int, nranks: int):
        assert len(unique_id) == TYR_DIST_ID_BYTES
        self.L = renderer.L
        self.r = renderer  # keeps the ctx alive
        h = P()
        _check(self.L.tyr_dist_create(C.byref(h), renderer.h, unique_id, rank, nranks), "tyr_dist_create")
        self.h = h
        renderer._dists.append(self)  # Renderer.close() closes its communicators first (a combine reads the ctx)

    def combine(self, frame_out_device_ptr, mode=TYR_DIST_GATHER, root=0):
        _check(self.L.tyr_dist_combine(self.h, mode, root, frame_out_device_ptr), "tyr_dist_combine")

    def info(self) -> dict:
        n, r = c_i32(-1), c_i32(-1)
        _check(self.L.tyr_dist_info(self.h, C.byref(n), C.byref(r)), "tyr_dist_info")
        return {"comm_ranks": int(n.value), "rank": int(r.value)}

    def wait(self):
        _check(self.L.tyr_dist_wait(self.h), "tyr_dist_wait")

    def close(self):
        if getattr(self, "h", None):
            self.L.tyr_dist_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
