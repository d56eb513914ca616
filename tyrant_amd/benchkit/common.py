"""constants and job shapes shared by bench.py and its child modes"""
from __future__ import annotations

import os
import shutil
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BENCH = os.path.join(ROOT, "bench.py")  # the child modes re-enter through its command line (--pmc-child, --dist-preflight)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
GATHER_CEILING_GBS = 7400.0  # ibid., "Indexed rows: gather into LDS": uniformly random rows out of the Infinity Cache, 7.4-7.9 TB/s (151 MB table) ... 8.6 TB/s (38 MB)
NUM_XCD, NUM_SIMD, NUM_CU = 8, 1024, 256  # MI355X: 8 XCDs x 32 CUs x 4 SIMDs
REF_N = 2097152  # variables.h:44
TRACE_KERNEL = os.environ.get("TYR_BENCH_TRACE_KERNEL", "k_trace_flat<12")   # the traversal kernel: extend(i + 1) + connect(i) in one launch (tyr_render), or one kind of ray alone.  A name PREFIX: rocprofv3 lists its two block shapes, k_trace_flat<12, 768u> (launches of 3 Mi rays and more: six waves per SIMD) and k_trace_flat<12, 256u>; both are "the kernel" of the roofline
SHADE_KERNEL = "k_shade<"            # the second kernel of a render by time
EXTEND_KERNEL = TRACE_KERNEL
SHADE_BYTES_PER_RAY = 52 + 24 + 16   # state + e1, e2 + pixel RMW; + 44 per survivor + 48 per shadow ray (added from the counters)


def dominant_kernel(tune_args) -> str:
    return TRACE_KERNEL
PMC_PASSES = (
    ("FETCH_SIZE",),
    ("WRITE_SIZE",),
    ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"),
)

def build_workload(name: str, binding, scenes):
    if name == "c2":
        sc = scenes.cornell_soup(10000)
        label = "C2: Cornell box (36 tris) + 10,000 seeded random diffuse triangles"
    elif name == "c3":
        sc = scenes.mesh_scene(706)
        label = "C3: room + 706x706 height-field mesh (996,882 tris), 70% DIFF / 30% SPEC"
    elif name == "c3_framed":
        sc = scenes.mesh_scene_framed(706)
        label = "C3 framed: the C3 scene from (0, -87.5, 50), where the room's opening fills the 16:9 frame (kernel.cu:698-699's 1.5 x W/H by 1.5 extents): every camera ray enters the room"
    elif name == "c5":
        sc = scenes.glass_dof_scene(2236)
        label = "C5: room + 2236x2236 height-field mesh (9,999,402 tris), 65% DIFF / 30% SPEC / 5% REFR, thin lens 0.5, sun (0.3,0.2); quoted at --width 3840 --height 2160 --spp 16"
    elif name == "c1":
        sc = scenes.cornell_box()
        label = "C1: Cornell box (36 tris)"
    else:
        raise SystemExit(f"unknown workload {name}")
    t0 = time.perf_counter()
    nodes, prims = binding.bvh_build(sc.triangles)  # host SAH build (bvh.cpp:3-225), outside the timed region
    return sc, nodes, prims, label, time.perf_counter() - t0

def job_shape(args, world: int):
    """(spp_total, queue slots per rank)"""
    if args.spp > 0:
        spp_total = args.spp * (world if args.scaling == "weak" else 1)
    elif world == 1:
        spp_total = 8
    else:
        spp_total = 64 if args.scaling == "strong" else 8 * world
    local_pixels = args.width * (args.height // world)
    N = args.queue if args.queue > 0 else min(spp_total * local_pixels, 1 << 25)
    return spp_total, N

def find_rocprof():
    p = shutil.which("rocprofv3")
    if p is None and os.path.exists("/opt/rocm/bin/rocprofv3"):
        p = "/opt/rocm/bin/rocprofv3"
    return p
