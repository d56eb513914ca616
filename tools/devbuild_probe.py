#!/usr/bin/env python3
"""tools/devbuild_probe.py [cells] -- tyr_bvh_build_device on the C3 mesh (or a cells x cells one), three times, beside the host
builder; for `rocprofv3 --kernel-trace --stats` (which kernels the 15 ms are)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tyrant_amd import binding, scenes  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 706
tris = scenes.mesh_scene(cells).triangles
binding.bvh_build_device(tris[:1000])
t0 = time.perf_counter()
hn, hp = binding.bvh_build(tris)
th = time.perf_counter() - t0
for _ in range(3):
    t0 = time.perf_counter()
    dn, dp, sec = binding.bvh_build_device(tris)
    tw = time.perf_counter() - t0
    print(f"{len(tris)} triangles -> {len(dn)} nodes: device {sec[0] * 1e3:.2f} ms + copies {sec[1] * 1e3:.2f} ms (call: {tw * 1e3:.1f} ms incl. allocation); host builder {th * 1e3:.1f} ms; same bytes: {dn.tobytes() == hn.tobytes() and dp.tobytes() == hp.tobytes()}", flush=True)
