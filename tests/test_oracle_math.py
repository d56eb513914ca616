"""Accuracy of the deterministic transcendental layer (oracle/orc_internal.h; the HIP kernels
implement the same contract in hip/detmath.hpp) against numpy's float64 libm.  CPU only."""
import ctypes as C

import numpy as np


def _ulp_err(got: np.ndarray, exact: np.ndarray) -> np.ndarray:
    """error of float32 `got` in units of the float32 spacing at `exact`"""
    ex32 = exact.astype(np.float32)
    spacing = np.spacing(np.abs(ex32)).astype(np.float64)
    spacing = np.where(spacing == 0, np.finfo(np.float32).tiny, spacing)
    return np.abs(got.astype(np.float64) - exact) / spacing


def _map1(fn, x):
    return np.array([fn(C.c_float(float(v))) for v in x], dtype=np.float32)


def test_sin_cos(orc):
    L = orc.lib()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-8, 8, 20000), rng.uniform(-0.01, 0.01, 2000), np.linspace(0, 2 * np.pi, 4097)]).astype(np.float32)
    xd = x.astype(np.float64)
    assert _ulp_err(_map1(L.orc_dm_sinf, x), np.sin(xd)).max() <= 0.5001 + 1e-3 * 0  # essentially correctly rounded
    assert _ulp_err(_map1(L.orc_dm_cosf, x), np.cos(xd)).max() <= 0.5001
    # exact symmetries of the kernel
    assert L.orc_dm_sinf(0.0) == 0.0 and L.orc_dm_cosf(0.0) == 1.0
    assert np.isnan(L.orc_dm_sinf(float("inf"))) and np.isnan(L.orc_dm_cosf(float("nan")))


def test_exp(orc):
    L = orc.lib()
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-100, 10, 20000), rng.uniform(-1e-3, 1e-3, 2000), [-87.5, -95.0, -103.0, 88.0, 0.0]]).astype(np.float32)
    got = _map1(L.orc_dm_expf, x)
    exact = np.exp(x.astype(np.float64))
    normal = exact > 1.2e-38
    assert _ulp_err(got[normal], exact[normal]).max() <= 0.5001
    # denormal results: absolute error below one denormal step
    assert np.all(np.abs(got[~normal].astype(np.float64) - exact[~normal]) <= 1.5e-45)
    assert L.orc_dm_expf(-1000.0) == 0.0 and L.orc_dm_expf(1000.0) == float("inf") and L.orc_dm_expf(0.0) == 1.0
    assert L.orc_dm_expf(float("-inf")) == 0.0 and np.isnan(L.orc_dm_expf(float("nan")))


def test_pow(orc):
    L = orc.lib()
    rng = np.random.default_rng(2)
    # the three uses on the path: pow(1 - r2, 1/41) and pow(c, 40) (kernel.cu:527, 554), pow(x, 1/2.2) (kernel.cu:661)
    for xs, y in (
        (rng.uniform(0, 1, 20000), np.float32(1.0) / np.float32(41.0)),
        (rng.uniform(1e-3, 1, 20000), np.float32(40.0)),
        (rng.uniform(0, 1, 20000), np.float32(1.0) / np.float32(2.2)),
        (rng.uniform(0, 50, 5000), np.float32(2.5)),
    ):
        x = xs.astype(np.float32)
        got = np.array([L.orc_dm_powf(C.c_float(float(v)), C.c_float(float(y))) for v in x], dtype=np.float32)
        exact = np.power(x.astype(np.float64), np.float64(y))
        ok = exact > 1.2e-38
        assert _ulp_err(got[ok], exact[ok]).max() <= 0.5001
    assert L.orc_dm_powf(0.0, 0.5) == 0.0 and L.orc_dm_powf(1.0, 123.0) == 1.0
    assert np.isnan(L.orc_dm_powf(float("nan"), 0.4545)) and np.isnan(L.orc_dm_powf(-1.0, 0.4545))
    assert L.orc_dm_powf(float("inf"), 0.4545) == float("inf")


def test_sampling_helpers(orc):
    L = orc.lib()
    fp = C.POINTER(C.c_float)
    # ConcentricSampleDisk (kernel.cu:190-208): inside the unit disk, centre maps to the centre
    rng = np.random.default_rng(3)
    for _ in range(2000):
        u = rng.uniform(0, 1, 2).astype(np.float32)
        out = np.zeros(2, dtype=np.float32)
        L.orc_concentric_sample_disk(u.ctypes.data_as(fp), out.ctypes.data_as(fp))
        assert out[0] ** 2 + out[1] ** 2 <= 1.0 + 1e-6
    c = np.array([0.5, 0.5], dtype=np.float32)
    out = np.ones(2, dtype=np.float32)
    L.orc_concentric_sample_disk(c.ctypes.data_as(fp), out.ctypes.data_as(fp))
    assert out[0] == 0 and out[1] == 0
    # computeOrthonormalBasisNaive (kernel.cu:181-189): orthonormal, right-handed w = u x v up to sign convention
    for _ in range(500):
        w = rng.normal(size=3)
        w = (w / np.linalg.norm(w)).astype(np.float32)
        u = np.zeros(3, dtype=np.float32)
        v = np.zeros(3, dtype=np.float32)
        L.orc_orthonormal_basis_naive(w.ctypes.data_as(fp), u.ctypes.data_as(fp), v.ctypes.data_as(fp))
        assert abs(np.dot(u, w)) < 1e-5 and abs(np.dot(v, w)) < 1e-5 and abs(np.dot(u, v)) < 1e-5
        assert abs(np.linalg.norm(u) - 1) < 1e-5 and abs(np.linalg.norm(v) - 1) < 1e-5
    # Random2DStratifiedSample (kernel.cu:44-65): inside the pixel
    s = C.c_uint32(777)
    for _ in range(2000):
        o = np.zeros(2, dtype=np.float32)
        L.orc_random_2d_stratified_sample(C.byref(s), o.ctypes.data_as(fp))
        assert 0 <= o[0] <= 1.0 and 0 <= o[1] <= 1.0
