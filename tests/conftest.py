"""Shared fixtures.

`-m "not gpu"`: oracle vs golden vectors / the reference-header harness, host logic,
C-ABI symbol checks -- no GPU, no /root/reference at run time (the harness .so is prebuilt).
`-m gpu`: parity of the HIP path (through the C ABI) against the oracle on an MI355X.
"""
from __future__ import annotations

import functools
import json
import os
import sys

import numpy as np
import pytest

# PyTorch's wheels bundle their own HIP runtime (libamdhip64.so + ROCr, loaded into the global symbol scope); the product
# library links the ROCm installation's.  A process must end up with ONE of them: when PyTorch is imported first, the
# library's HIP calls bind to PyTorch's copy (what bench.py has always done); loaded the other way round, the process
# holds two runtimes and the second one to initialise finds no device.  The GPU tests that use torch for device memory
# therefore import it here, before any fixture loads the library (tyrant_amd/binding.py lib()).
try:
    import torch  # noqa: F401
except ImportError:  # the library alone: one runtime anyway
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run by the driver with -m gpu)")


# Collection order (round 4).  The driver runs `pytest -m gpu -x -q`: whatever is collected first decides whether anything
# else runs at all.  Kernel parity therefore comes first, host / scene / camera tests next, and tests that only check the
# FORMAT of bench.py's line (child processes, timings of micro-jobs) last: a bench-format assertion can never again
# stand in front of the parity tests (round 3: one rounded timing, `0.0 > 0`, left 101 parity tests unreached).
_ORDER = ("test_gpu_parity", "test_gpu_configs", "test_ref_pins", "test_bvh_build_device", "test_glm_pinning", "test_debug_bvh", "test_dist_native", "test_scene_io",
          "test_camera_input", "test_oracle_golden", "test_oracle_math", "test_oracle_wavefront", "test_host_and_abi", "test_dist_gloo")
_LAST = ("test_bench_contract",)


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if mod in _LAST:
            return len(_ORDER) + 1 + _LAST.index(mod)
        return _ORDER.index(mod) if mod in _ORDER else len(_ORDER)

    items.sort(key=rank)  # stable: the order inside a module stays the file's


@pytest.fixture(scope="session")
def orc():
    from oracle import pyorc

    pyorc.lib()  # builds oracle/_build/liborc.so on first use
    return pyorc


@pytest.fixture(scope="session")
def ref(orc):
    r = orc.ref()
    if r is None:
        pytest.skip("oracle/_ref/libref_traverse.so not built (needs /root/reference; run `make -C oracle ref`)")
    return r


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


@functools.lru_cache(maxsize=None)
def built_scene(name: str):
    """(scene, nodes, prims) with the BVH built by the ORACLE's builder (tests only)."""
    from oracle import pyorc
    from tyrant_amd import scenes

    makers = {
        "cornell36": scenes.cornell_box,
        "cornell_soup2k": lambda: scenes.cornell_soup(2000),
        "cornell_soup10k": lambda: scenes.cornell_soup(10000),
        "mesh32": lambda: scenes.mesh_scene(32),
        "mesh128": lambda: scenes.mesh_scene(128),
        "mesh706": lambda: scenes.mesh_scene(706),
        "tyrant_default": scenes.tyrant_default,
        "glass_dof48": lambda: scenes.glass_dof_scene(48),
        "cornell_area_light": scenes.cornell_area_light,
        "cornell_colored": scenes.cornell_colored,
    }
    sc = makers[name]()
    nodes, prims = pyorc.bvh_build(sc.triangles, scenes.triangle_bboxes(sc.triangles))
    return sc, nodes, prims


def bits(a: np.ndarray) -> np.ndarray:
    """float32 array -> uint32 bit patterns (for exact comparisons that treat NaN == NaN)"""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="session")
def hip():
    """the product library through its C ABI; GPU tests fail loudly when it is missing"""
    from tyrant_amd import binding

    binding.lib()
    return binding
