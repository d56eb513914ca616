#!/usr/bin/env python3
"""bench.py -- Mrays/s of the wavefront path-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With N > 1 and no launcher in the environment the parent starts the N ranks itself (torch.distributed.run as a
child process, before anything here has touched a GPU) and relays rank 0's JSON line; under a launcher
(RANK / WORLD_SIZE set) every rank runs main() directly, one rank per GPU.

One step = one complete render of the workload through the C ABI (tyr_reset_accum + tyr_render: top-up -> extend ->
shade -> connect until every path has finished) plus, for N > 1, the combine of the ranks' rows on rank 0 over RCCL.
The scene lives in HBM before the timed region starts; nothing crosses PCIe inside it except 200-byte counter reads.

Workloads (BASELINE.json configs):
    c3  room + 706x706 height-field mesh (996,882 triangles), 30 % SPEC, 1920x1080, 8 spp   (default: the scene
        north_star's target names; N = 1)
    c2  Cornell box + 10,000 seeded random diffuse triangles, 1920x1080, 8 spp
    c5  room + 2236x2236 mesh (9,999,402 triangles), 5 % REFR, thin lens, sun; quoted at --width 3840 --height 2160 --spp 16
    N > 1 (default --scaling strong) is config C4: the c3 scene at 64 spp IN TOTAL, the frame's rows dealt
    y % N == rank, so the job is fixed and every GPU renders 64 spp over 1/N of the pixels (N = 8: exactly the one-GPU
    c3 job per GPU).  rank 0 afterwards renders the same 64-spp job alone: config.strong_scaling holds the one-GPU
    time of the SAME job, the speed-up and the efficiency.  --scaling weak renders 8*N spp (fixed work per GPU).
Queue size: the reference's ray_queue_buffer_size (2,097,152, variables.h:44) was chosen for a small GPU and forces 20
thin wavefront iterations per 8-spp frame.  It is a runtime parameter here and the headline run sizes it for the GPU --
spp x local pixels slots (16.6 M: 10.4 GB of 288 GB with the queues' eight segments per class sized for the worst case; capped at 32 Mi), i.e. every primary ray of the render in flight at
once.  The same workload at the reference's queue size is measured too (config.reference_queue_size).
config.steady_state: the same kernels with the queue kept full by top-ups for as long as the measurement lasts (the
reference's viewer never stops: main.cpp:164-170) -- no thin iterations at the end of a render.  Reported beside the
metric, never as it.

Mrays/s = (extend rays + shadow rays traced by all ranks) / wall time.  config.in_tree_Mrays/s
is the same with only the rays that pass the root box (three primary rays in four miss the tree on c3 and cost one box
test each).

roofline: the dominant kernel (the production extend kernel).  Three fractions are reported and `bound` names the
tightest: the HBM fraction from the memory-side counters (FETCH_SIZE x 2 + WRITE_SIZE per launch / launch time / 8 TB/s),
and the vector / scalar instruction-issue fractions (SQ_ACTIVE_INST_VALU / _SCA against the SIMD / scalar-unit cycles).
The counters are measured live: before this process touches the GPU it runs three `rocprofv3 --pmc` passes over a
one-render child of itself (--pmc off skips them and falls back to the committed profiles/pmc_<workload>.json).  The
ALGORITHMIC figure of SURVEY.md 8d (24 + 8 + 32 B per node and 36 B per triangle the REFERENCE's binary tree visits,
counted by the library's counting build on pair nodes, not by the timed quad-node kernel) over the hipEvent time of
the timed launches is reported beside it as roofline.algorithmic -- the tree is largely cache resident, so that
figure prices bytes the fabric never carried and may exceed the HBM peak; it is not `frac`.

cpu_baseline (N = 1), 1 core of this host: `value` is ONE ray set (the queue of the third iteration: bounce rays in front,
fresh primary rays behind) through the reference's own CachedBVH::intersect (oracle/_ref, bvh.h:118-161; kind "reference")
-- or the oracle's restatement of it where oracle/_ref is absent (kind "port"); both are listed, their distances are
bit-identical.  Beside it: the oracle's whole wavefront loop on the first iterations of the workload, and the SAH build
of the scene by the port and by the reference's own bvh.cpp.
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
NUM_XCD, NUM_SIMD, NUM_CU = 8, 1024, 256  # MI355X: 8 XCDs x 32 CUs x 4 SIMDs
REF_N = 2097152  # variables.h:44
TRACE_KERNEL = "k_trace_flat<12"   # the traversal kernel: extend(i + 1) + connect(i) in one launch (tyr_render), or one kind of ray alone.  A name PREFIX: rocprofv3 lists its two block shapes, k_trace_flat<12, 768u> (launches of 3 Mi rays and more: six waves per SIMD) and k_trace_flat<12, 256u>; both are "the kernel" of the roofline
SHADE_KERNEL = "k_shade<"            # the second kernel of a render by time
EXTEND_KERNEL = TRACE_KERNEL
SHADE_BYTES_PER_RAY = 52 + 24 + 16   # state + e1, e2 + pixel RMW; + 44 per survivor + 48 per shadow ray (added from the counters)


ORACLE_COUNTER_FIELDS = ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible")


def oracle_counters_check(args, world, W, H, spp, N, n_tris, m):
    """config.oracle_counters_match: the counter deltas of the TIMED renders against the oracle's counters for this very job,
    committed as tests/golden/bench_c3_counters.json (made by tests/golden/make_bench_counters.py: orc_render, the serial C
    restatement of kernel.cu:664-748).  Every timed step restarts the frame counter, so K steps must have cast exactly K times
    the oracle's rays -- extend, shadow, survivors, visible shadow rays, iterations.  None when the job is not the committed one
    (another workload, resolution, spp, queue size or rank count).  The file is data: nothing under oracle/ is loaded here."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", f"bench_{args.workload}_counters.json")) as f:
            gold = json.load(f)
    except (OSError, ValueError):
        return {"oracle_counters_match": None, "oracle_counters_note": f"no committed oracle counters for workload {args.workload}"}
    j = gold["job"]
    if world != 1 or (j["width"], j["height"], j["spp"], j["queue_size"], j["triangles"]) != (W, H, spp, N, n_tris):
        return {"oracle_counters_match": None, "oracle_counters_note": "this job is not the one the committed oracle counters were made for (tests/golden/make_bench_counters.py: c3, 1920x1080, 8 spp, queue 16,588,800, one rank)"}
    want = {f: gold["per_render"][f] * args.steps for f in ORACLE_COUNTER_FIELDS}
    got = m["counter_deltas"]
    ok = all(int(got[f]) == int(want[f]) for f in ORACLE_COUNTER_FIELDS) and m["iters"] == gold["per_render"]["iterations"] * args.steps
    out = {"oracle_counters_match": bool(ok),
           "oracle_counters": {"source": "tests/golden/bench_c3_counters.json (orc_render on this job; tests/test_gpu_configs.py::test_benchmarked_render_path_matches_oracle_at_full_size[bench_shape_16M_8spp] holds the live oracle, the file and the GPU to each other, pixels included)",
                               "per_render": gold["per_render"], "timed_renders": args.steps}}
    if not ok:
        out["oracle_counters"]["timed_deltas"] = {f: int(got[f]) for f in ORACLE_COUNTER_FIELDS}
        out["oracle_counters"]["timed_iterations"] = m["iters"]
    return out


def dominant_kernel(tune_args) -> str:
    return TRACE_KERNEL
PMC_PASSES = (
    ("FETCH_SIZE",),
    ("WRITE_SIZE",),
    ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"),
)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel: 0 = the configuration's (8 at one GPU; N > 1: 64 in total when --scaling strong, 8 per GPU when weak)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1: strong = BASELINE config C4, 64 spp in total whatever N; weak = 8*N spp")
    ap.add_argument("--queue", type=int, default=0, help="ray_queue_buffer_size (variables.h:44); 0 = sized for the GPU: spp x local pixels, at most 32 Mi slots")
    ap.add_argument("--no-reference-queue", action="store_true", help="skip the second measurement at the reference's queue size (2,097,152)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-steady-state", action="store_true", help="skip config.steady_state (the queue kept full by top-ups, as the reference's viewer runs: no thin iterations)")
    ap.add_argument("--no-one-gpu-job", action="store_true", help="N > 1, strong: skip rank 0's solo render of the same job (strong_scaling block)")
    ap.add_argument("--cpu-iterations", type=int, default=2)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearse the multi-rank path with every rank on ONE GPU (the combine then goes through host memory and torch)")
    ap.add_argument("--combine", default="gather", choices=["gather", "reduce"], help="N > 1: rank 0 gets the rows each rank owns (1/N of the frame per rank) or the sum of the full buffers")
    ap.add_argument("--dist-preflight", action="store_true", help=argparse.SUPPRESS)  # internal: one rank of the native exchange's pre-flight check
    ap.add_argument("--preflight-store", default="", help=argparse.SUPPRESS)  # internal: the file the pre-flight children rendezvous through
    ap.add_argument("--combine-impl", default="native", choices=["native", "torch"], help="native = tyr_dist_* (RCCL behind the C ABI, double-buffered); torch = torch.distributed collectives (always used with --backend gloo)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"], help="auto: N = 1 and rocprofv3 present -> three --pmc child passes feed the roofline block")
    ap.add_argument("--save-pmc", default="", help="write the live PMC counters to this JSON file (copied to profiles/pmc_<workload>.json, the fallback when rocprofv3 cannot run beside the bench)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--tune", action="append", default=[], help="launch-shape knob of tyr_set_tuning, e.g. --tune refill_min_idle=8 (never changes results)")
    return ap.parse_args(argv)


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args) -> int:
    """--gpus N > 1 without a launcher: start N ranks as a CHILD (this process has not touched a GPU and never will:
    no exec of a process that holds the device), relay their output and exit code"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def build_workload(name: str, binding, scenes):
    if name == "c2":
        sc = scenes.cornell_soup(10000)
        label = "C2: Cornell box (36 tris) + 10,000 seeded random diffuse triangles"
    elif name == "c3":
        sc = scenes.mesh_scene(706)
        label = "C3: room + 706x706 height-field mesh (996,882 tris), 70% DIFF / 30% SPEC"
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


# ---------------------------------------------------------------------------------------------------------------
# PMC child passes (N = 1): `rocprofv3 --pmc <counters> -- python3 bench.py --pmc-child ...`, the program itself after
# `--`, no tracing domain combined with --pmc, one pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one).
# ---------------------------------------------------------------------------------------------------------------
def pmc_child(args) -> int:
    """one cold + one counted render of the workload, no torch, no timing: what the profiler looks at"""
    from tyrant_amd import binding, scenes

    spp_total, N = job_shape(args, 1)
    sc, nodes, prims, _, _ = build_workload(args.workload, binding, scenes)
    flags = binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0
    r = binding.Renderer(args.width, args.height, N, flags=flags)
    r.load_scene(sc, nodes, prims)
    tune = {k: int(v) for k, v in (kv.split("=") for kv in args.tune)}
    if tune:
        r.set_tuning(**tune)
    iters = 0
    for _ in range(2):
        r.reset_accum()
        iters = r.render(spp_total)
    k = r.counters()
    assert k["device_error"] == 0
    r.close()
    print(json.dumps({"pmc_child_iterations": iters}), flush=True)
    if os.environ.get("TYR_BENCH_PRINT_DEBUG"):  # the -DTYR_QUAD_STATS build's loop counters (quad_block)
        print(json.dumps({"child_debug": [int(v) for v in k["debug"]], "renders": 2}), flush=True)
    return 0


def find_rocprof():
    p = shutil.which("rocprofv3")
    if p is None and os.path.exists("/opt/rocm/bin/rocprofv3"):
        p = "/opt/rocm/bin/rocprofv3"
    return p


def run_pmc_passes(args, timeout_s: float = 150.0):
    """-> {"counters": {name: average per launch of the production extend kernel in the LAST render}, "launches": n} or None"""
    rocprof = find_rocprof()
    if rocprof is None:
        return None
    out_root = tempfile.mkdtemp(prefix="tyr_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload, "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp),
             "--queue", str(args.queue)] + [x for kv in args.tune for x in ("--tune", kv)]
    counters, launches = {}, None
    shade_counters, shade_launches = {}, None
    kernel = dominant_kernel(args.tune)
    try:
        for i, group in enumerate(PMC_PASSES):
            d = os.path.join(out_root, f"g{i}")
            cmd = [rocprof, "--pmc", *group, "--output-format", "csv", "-d", d, "-o", "pmc", "--"] + child
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=out_root, env=dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp")))
            if p.returncode != 0:
                print(f"[bench] rocprofv3 --pmc {' '.join(group)} failed (rc {p.returncode}): {(p.stderr or p.stdout)[-300:]}", file=sys.stderr)
                return None
            iters = None
            for line in p.stdout.splitlines():
                if line.startswith('{"pmc_child_iterations"'):
                    iters = json.loads(line)["pmc_child_iterations"]
            allrows = []
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    allrows += list(csv.DictReader(f))
            rows = [r for r in allrows if kernel in r["Kernel_Name"]]
            if not rows or not iters:
                print(f"[bench] rocprofv3 pass {group}: no rows for {kernel}", file=sys.stderr)
                return None
            for name in group:
                mine = sorted((r for r in rows if r["Counter_Name"] == name), key=lambda r: int(r["Dispatch_Id"]))
                if len(mine) < 2 or len(mine) % 2:
                    return None
                last = mine[len(mine) // 2:]  # the child renders twice: the second (warm) render's launches of this kernel
                counters[name] = sum(float(r["Counter_Value"]) for r in last) / len(last)
                launches = len(last)
                # the same for the shade kernel (summed over the render's launches: its per-render figure)
                sh = sorted((r for r in allrows if SHADE_KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name), key=lambda r: int(r["Dispatch_Id"]))
                if sh and len(sh) % 2 == 0:
                    shade_counters[name] = sum(float(r["Counter_Value"]) for r in sh[len(sh) // 2:])
                    shade_launches = len(sh) // 2
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        print(f"[bench] PMC passes abandoned: {e!r}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(out_root, ignore_errors=True)
    return {"counters": counters, "launches_averaged": launches, "kernel": kernel, "shade_counters_per_render": shade_counters, "shade_launches_per_render": shade_launches, "source": "live: rocprofv3 --pmc child passes of this command (" + " | ".join(" ".join(g) for g in PMC_PASSES) + ")"}


def committed_pmc(workload: str, N: int):
    try:
        with open(os.path.join(ROOT, "profiles", f"pmc_{workload}.json")) as f:
            j = json.load(f)
        if j.get("queue_size") == N:
            return {"counters": j["counters"], "launches_averaged": j.get("launches_averaged"), "kernel": j.get("kernel"), "shade_counters_per_render": j.get("shade_counters_per_render", {}), "shade_launches_per_render": j.get("shade_launches_per_render"), "source": f"committed: profiles/pmc_{workload}.json ({j.get('source', '')})"}
    except (OSError, KeyError, ValueError):
        pass
    return None


def shade_block(pmc, shade_ms_per_render, rays_per_render, survivors_per_render, shadows_per_render):
    """the second kernel of a render: k_shade against its byte roofline (52 + 24 + 16 B per ray, 44 per
    survivor, 48 per shadow ray) and, from the counters, its vector-issue fraction -- it is bound by arithmetic"""
    alg = SHADE_BYTES_PER_RAY * rays_per_render + 44.0 * survivors_per_render + 48.0 * shadows_per_render
    t = shade_ms_per_render * 1e-3
    out = {"kernel": "k_shade<false>", "ms_per_render": round(shade_ms_per_render, 4), "rays_per_render": int(rays_per_render),
           "algorithmic": {"bytes_per_render": round(alg), "GBps": round(alg / t / 1e9, 2) if t > 0 else None, "frac_of_hbm_peak": round(alg / t / 1e9 / HBM_PEAK_GBS, 4) if t > 0 else None,
                           "bytes_per_ray": round(alg / max(rays_per_render, 1), 1)}}
    c = (pmc or {}).get("shade_counters_per_render") or {}
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_ACTIVE_INST_VALU"):
        cyc = c["GRBM_GUI_ACTIVE"] / NUM_XCD
        valu = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (NUM_SIMD * cyc)
        lanes = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        out.update({"bound": "valu-issue", "frac": round(valu, 4), "frac_kind": "vector-ALU issue cycles / SIMD cycles while k_shade runs (not an HBM fraction)",
                    "salu_issue_frac": round(4.0 * c["SQ_ACTIVE_INST_SCA"] / (NUM_SIMD * cyc), 4), "lanes_active_per_valu_inst": round(lanes, 4)})
        if c.get("FETCH_SIZE") is not None and c.get("WRITE_SIZE") is not None and t > 0:
            hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            out["traffic"] = round(hbm / t / 1e9, 2)
            out["hbm_counter_frac"] = round(hbm / t / 1e9 / HBM_PEAK_GBS, 4)
    return out


def steady_state(binding, sc, nodes, prims, W, H, N, flags, device):
    """The path as the reference's viewer drives it (main.cpp:164-170): launch_kernels without end, every iteration's
    queue topped up with fresh camera rays (kernel.cu:247-297), so that no iteration is thin.  An 8-spp render -- the
    metric -- ends in four or five iterations of survivors only, each paying a traversal launch's drain for a tenth of the
    rays; this is the same kernels' throughput without that tail.  NOT the metric: reported beside it."""
    import time

    import torch

    r = binding.Renderer(W, H, N, device=device, flags=flags)
    r.load_scene(sc, nodes, prims)
    spp_never = 1 << 12  # a budget the measurement never uses up (tyr_render sets it again at every call)
    r.render(spp_never, 8)  # the mix of fresh rays and survivors of every depth has settled after max-bounces iterations
    torch.cuda.synchronize()
    k0 = r.counters()
    t0 = time.perf_counter()
    iters = r.render(spp_never, 12)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k1 = r.counters()
    ok = k1["device_error"] == 0
    r.close()
    rays = (k1["total_extend_rays"] - k0["total_extend_rays"]) + (k1["total_shadow_rays"] - k0["total_shadow_rays"])
    if not ok or dt <= 0 or iters == 0:
        return None
    return {"Mrays/s": round(rays / dt / 1e6, 1), "ms_per_iteration": round(dt / iters * 1e3, 3), "iterations": iters, "queue_size": N,
            "note": "tyr_render with a budget that never runs out: every iteration's queue is full (survivors + top-up), the reference viewer's mode; an 8-spp render ends in thin iterations instead"}


def drain_block(args):
    """How much of a traversal launch is its drain: an instrumented build of the library (-DTYR_LAUNCH_ANATOMY: three
    s_memrealtime stamps per wave) renders the workload once in a CHILD process; per launch, `feed` = first wave's start ->
    first wave to find the queue used up, `drain` = from there to the last wave's exit."""
    lib = os.path.join(ROOT, "tyrant_amd", "lib", "libtyrant_hip_anatomy.so")
    if not os.path.exists(lib):
        return None
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload, "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp), "--queue", str(args.queue)]
    try:
        p = subprocess.run(child, capture_output=True, text=True, timeout=120, env=dict(os.environ, TYRANT_HIP_LIBRARY=lib, TYR_ANATOMY="1"))
    except (subprocess.TimeoutExpired, OSError):
        return None
    rows = []
    for line in p.stderr.splitlines():
        if line.startswith("[anatomy]") and "feed" in line:
            try:
                rows.append((float(line.split("feed")[1].split("us")[0]), float(line.split("drain")[1].split("us")[0])))
            except (IndexError, ValueError):
                pass
    if p.returncode != 0 or len(rows) < 2:
        return None
    rows = rows[len(rows) // 2:]  # the second (warm) render
    feed, drain = sum(r[0] for r in rows), sum(r[1] for r in rows)
    return {"drain_frac": round(drain / (feed + drain), 4), "feed_us_per_launch": [round(r[0], 1) for r in rows], "drain_us_per_launch": [round(r[1], 1) for r in rows],
            "source": "one render of the same workload by libtyrant_hip_anatomy.so (-DTYR_LAUNCH_ANATOMY) in a child process; the render's last launch (shadow rays only) is not stamped"}


def quad_block(args):
    """What the timed kernel's OWN layout needs, counted by an instrumented build of it (-DTYR_QUAD_STATS) in a child process:
    quad steps (one 128-byte quad node each, 112 bytes of it read) and triangle tests (48-byte records) per render."""
    lib = os.path.join(ROOT, "tyrant_amd", "lib", "libtyrant_hip_stats.so")
    if not os.path.exists(lib):
        return None
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload, "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp), "--queue", str(args.queue)]
    try:
        p = subprocess.run(child, capture_output=True, text=True, timeout=120, env=dict(os.environ, TYRANT_HIP_LIBRARY=lib, TYR_BENCH_PRINT_DEBUG="1"))
    except (subprocess.TimeoutExpired, OSError):
        return None
    if p.returncode != 0:
        return None
    for line in p.stdout.splitlines():
        if line.startswith('{"child_debug"'):
            j = json.loads(line)
            d, renders = j["child_debug"], max(j.get("renders", 1), 1)
            # tyr_counters.debug of the TYR_QUAD_STATS build: [1] lanes x trips of the quad-test loop, [5] lanes x trips of the triangle loop (traverse_flat.hip TYR_DBG)
            return {"quad_steps_per_render": d[1] / renders, "triangle_tests_per_render": d[5] / renders,
                    "source": "one cold + one warm render of the same workload by libtyrant_hip_stats.so (-DTYR_QUAD_STATS) in a child process, averaged"}
    return None


def roofline_block(pmc, ext_ms, ext_launches, ext_rays, visits, kernel_ms_per_render, kernel=EXTEND_KERNEL, con_ms=0.0, shadow_rays=0.0, quad=None, renders=1):
    """`bound` = the tightest of the measured resource fractions of the dominant kernel; the algorithmic-bytes figure of
    SURVEY.md 8d is a separate entry.  Merged launches (kernel = TRACE_KERNEL): the kernel traces this iteration's extend
    rays and the previous iteration's shadow rays, and the launch that ends a render with the last iteration's shadow rays is
    the same kernel: ext_ms / ext_launches are ALL its launches (main() adds the one timed as the connect stage; con_ms stays
    for callers that time a connect launch apart); the algorithmic figure covers the whole traversal stage (all extend + all
    shadow rays over ext_ms + con_ms)."""
    avg_launch_s = ext_ms / max(ext_launches, 1) * 1e-3
    bytes_per_ext = 24 + 8 + 32 * visits["nodes_per_ext"] + 36 * visits["tris_per_ext"]
    merged = kernel == TRACE_KERNEL
    if merged:
        bytes_per_con = 44 + 32 * visits["nodes_per_con"] + 36 * visits["tris_per_con"] + 12 * visits.get("visible_frac", 0.0)
        alg_total = bytes_per_ext * ext_rays + bytes_per_con * shadow_rays
        alg_gbs = alg_total / ((ext_ms + con_ms) * 1e-3) / 1e9 if ext_ms + con_ms > 0 else 0.0
        alg_bytes_per_launch = alg_total / max(ext_launches, 1)
    else:
        alg_bytes_per_launch = bytes_per_ext * ext_rays / max(ext_launches, 1)
        alg_gbs = alg_bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    out = {
        "kernel": f"{kernel}, 768u | 256u> " + ("(768-thread blocks, six waves per SIMD, for launches of at least TYR_TUNE_WIDE_BLOCK_MIN_ITEMS rays, 256-thread blocks at five otherwise; extend of an iteration + connect of the one before in one persistent launch: quad nodes, closest- and any-hit rays side by side)" if merged else "(the production extend kernel: quad nodes, persistent grid)"),
        "avg_launch_ms": round(avg_launch_s * 1e3, 4),
        "launches": ext_launches,
        "launch_time_source": "hipEvent pairs on the ctx stream around the stage (sphere pre-passes + the traversal kernel) inside the timed region",
    }
    fr = {}
    if pmc:
        c = pmc["counters"]
        hbm_bytes = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0  # KB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM)
        traffic = hbm_bytes / avg_launch_s / 1e9
        cyc = c["GRBM_GUI_ACTIVE"] / NUM_XCD  # the counter sums the XCDs' clocks
        fr["hbm"] = traffic / HBM_PEAK_GBS
        fr["valu-issue"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (NUM_SIMD * cyc)  # quad-cycles a SIMD spends issuing vector ALU work
        fr["salu-issue"] = 4.0 * c["SQ_ACTIVE_INST_SCA"] / (NUM_SIMD * cyc)   # = busy cycles of the CU's one scalar unit (shared by 4 SIMDs)
        lanes = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        bound = max(fr, key=fr.get)
        out.update({
            "bound": bound,
            "achieved": round(traffic, 2) if bound == "hbm" else round(100.0 * fr[bound], 2),
            "peak": HBM_PEAK_GBS if bound == "hbm" else 100.0,
            "unit": "GB/s" if bound == "hbm" else "% of issue cycles (SQ_ACTIVE_INST_* x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8))",
            "frac": round(fr[bound], 4),
            "frac_kind": ("HBM bytes by the memory-side counters / 8 TB/s" if bound == "hbm" else ("vector" if bound == "valu-issue" else "scalar") + "-ALU issue cycles / available cycles while the kernel runs: the tightest MEASURED resource fraction -- NOT an HBM fraction (that is hbm_counter_frac; the nominal byte count of SURVEY.md 8d is algorithmic.frac_of_hbm_peak)"),
            "traffic": round(traffic, 2),
            "hbm_counter_frac": round(fr["hbm"], 4),
            "valu_issue_frac": round(fr["valu-issue"], 4),
            "salu_issue_frac": round(fr["salu-issue"], 4),
            "lanes_active_per_valu_inst": round(lanes, 4),
            "useful_lane_issue_frac": round(fr["valu-issue"] * lanes, 4),
            "wave_wait_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAVE_CYCLES") else None,
            "traffic_detail": {"hbm_bytes_per_launch": round(hbm_bytes), "FETCH_SIZE_KB": round(c["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(c["WRITE_SIZE"], 1), "correction": "FETCH_SIZE x 2 (gfx950), WRITE_SIZE as is"},
            "pmc_source": pmc["source"],
            "pmc_launches_averaged": pmc["launches_averaged"],
        })
    else:
        # no counters at all: only the nominal figure exists; it is an HBM fraction only while it stays below 1
        out.update({"bound": "hbm", "achieved": round(min(alg_gbs, HBM_PEAK_GBS), 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(min(alg_gbs / HBM_PEAK_GBS, 1.0), 4), "traffic": None,
                    "note": "no PMC data (rocprofv3 absent and no committed profile): algorithmic bytes, clamped at the peak"})
    out["algorithmic"] = {
        "GBps": round(alg_gbs, 2),
        "frac_of_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4),
        "bytes_per_extend_ray": round(bytes_per_ext, 1),
        "bytes_per_launch": round(alg_bytes_per_launch),
        "covers": "every extend and every shadow ray of the timed renders over the time of all trace launches (the one that ends a render with the last shadow rays included)" if merged else "the extend launches",
        "nodes_per_ray": round(visits["nodes_per_ext"], 2),
        "tris_per_ray": round(visits["tris_per_ext"], 3),
        "connect_nodes_per_ray": round(visits["nodes_per_con"], 2),
        "connect_tris_per_ray": round(visits["tris_per_con"], 3),
        "note": "SURVEY.md 8d: 24 + 8 + 32 B x nodes + 36 B x triangles the REFERENCE's binary tree visits per extend ray (44 + 32 x nodes + 36 x triangles + 12 x p_visible per shadow ray), counted by the counting build (k_extend_count / k_connect_count, pair nodes) in an untimed render -- not by the timed quad-node kernel; nominal, exceeds the HBM peak when the tree is cache resident",
    }
    # (a) the same nominal count charged to the traversal kernel only for the rays it is handed: extend rays of class 1
    # (they fail the root box in the kernel that MAKES them, hip/kernels.hpp "Queues") cost it nothing -- their one
    # box test (32 B of the nominal count) and their 32-byte record belong to k_primary / k_shade
    t_all = (ext_ms + con_ms) * 1e-3
    in_ext = visits.get("in_tree_ext_frac")
    if in_ext is not None and merged and t_all > 0:
        rays_in = in_ext * ext_rays
        nodes_in = max(visits["nodes_per_ext"] * ext_rays - (ext_rays - rays_in), 0.0)  # the counting build counts ONE node for a ray that misses the root box
        alg_in = (24 + 8) * rays_in + 32 * nodes_in + 36 * visits["tris_per_ext"] * ext_rays + bytes_per_con * shadow_rays
        out["algorithmic"]["class0_only"] = {"GBps": round(alg_in / t_all / 1e9, 2), "frac_of_hbm_peak": round(alg_in / t_all / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_launch": round(alg_in / max(ext_launches, 1)),
                                             "extend_rays_charged": round(rays_in), "note": "SURVEY.md 8d's count for the rays that reach k_trace_flat: extend rays that pass the root box + every shadow ray"}
    # (b) what the kernel's own layout needs: 128 B per quad step, 48 B per triangle test, 32 B per ray handed to it
    if quad and t_all > 0:
        handed = (in_ext if in_ext is not None else 1.0) * ext_rays + shadow_rays
        qb = (128.0 * quad["quad_steps_per_render"] + 48.0 * quad["triangle_tests_per_render"]) * renders + 32.0 * handed
        out["algorithmic"]["quad"] = {"GBps": round(qb / t_all / 1e9, 2), "frac_of_hbm_peak": round(qb / t_all / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_launch": round(qb / max(ext_launches, 1)),
                                      "quad_steps_per_render": round(quad["quad_steps_per_render"]), "triangle_tests_per_render": round(quad["triangle_tests_per_render"]),
                                      "note": "bytes the 128-byte quad nodes and 48-byte triangle records of the timed kernel amount to (every step and test counted, cache hits included): " + quad["source"]}
    # (c) north_star's ">= 50 % of the HBM roofline", answered both ways
    out["hbm_target_met"] = {"target": 0.5, "nominal": bool(out["algorithmic"]["frac_of_hbm_peak"] >= 0.5), "nominal_class0_only": (bool(out["algorithmic"]["class0_only"]["frac_of_hbm_peak"] >= 0.5) if "class0_only" in out["algorithmic"] else None),
                             "counters": (bool(out["hbm_counter_frac"] >= 0.5) if out.get("hbm_counter_frac") is not None else None),
                             "note": "nominal = SURVEY.md 8d's per-ray bytes of the REFERENCE's binary tree over the traversal time (can exceed 1: not traffic); counters = bytes that crossed the fabric (FETCH_SIZE x 2 + WRITE_SIZE) / 8 TB/s -- the tree lives in the 256 MB Infinity Cache and the kernel is bound by instruction issue and by its launches' drains, not by HBM"}
    out["kernel_ms_per_render"] = kernel_ms_per_render
    return out


PREFLIGHT_TIMEOUT_S = 150.0
PREFLIGHT_TORCH_NCCL_ONLY = 2  # exit code of the pre-flight child: the native exchange failed, torch's nccl backend works


def dist_preflight(args) -> int:
    """One rank of the pre-flight check of the native exchange (tyr_dist_*: RCCL behind the C ABI), run as a CHILD of the
    bench rank of the same number before that rank has touched its GPU: a small frame, rows dealt y % world == rank, a
    2-spp render per rank, GATHER and REDUCE onto rank 0, every pixel must hold exactly 2 finished paths.  The exchange
    has only met one GPU per box before the driver's multi-GPU run; a hang, a crash or a wrong frame here costs this child,
    not the measurement: the bench ranks then use torch.distributed for the combine.  Exit code 0 = verified on every rank."""
    import datetime

    import torch
    import torch.distributed as dist

    from tyrant_amd import binding, scenes

    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("TYR_BENCH_PREFLIGHT_ONE_DEVICE"):
        local_rank = 0  # rehearsal on a one-GPU box: RCCL refuses two ranks on one device, which is the failure path under test
    local_rank %= max(torch.cuda.device_count(), 1)  # (fewer GPUs than ranks: the same failure path, not an invalid-device crash)
    dist.init_process_group("gloo", init_method=f"file://{args.preflight_store}", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    ok = 1
    try:
        torch.cuda.set_device(local_rank)
        W, H, spp = 64, 8 * world, 2
        sc = scenes.cornell_box()
        nodes, prims = binding.bvh_build(sc.triangles)
        r = binding.Renderer(W, H, 4096, device=local_rank, rank=rank, nranks=world)
        r.load_scene(sc, nodes, prims)
        ids = [binding.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = binding.Dist(r, ids[0], rank, world)
        frame = torch.zeros(H * W * 4, dtype=torch.float32, device=f"cuda:{local_rank}") if rank == 0 else None
        torch.cuda.synchronize()
        for mode in (binding.TYR_DIST_GATHER, binding.TYR_DIST_REDUCE):
            r.reset_accum()
            r.render(spp)
            comm.combine(frame.data_ptr() if frame is not None else None, mode=mode, root=0)
            comm.wait()
            torch.cuda.synchronize()
            if rank == 0:
                a = frame.view(H * W, 4)[:, 3]
                if not (float(a.min()) == float(a.max()) == float(spp)):
                    print(f"[bench preflight] mode {mode}: combined frame holds {float(a.min())}..{float(a.max())} paths per pixel, expected {spp}", file=sys.stderr)
                    ok = 0
                frame.zero_()
        comm.close()
        r.close()
    except Exception as e:  # noqa: BLE001
        print(f"[bench preflight] rank {rank}: {e!r}", file=sys.stderr)
        ok = 0
    # ... and torch's own RCCL backend (what the combine falls back to when the native exchange does not verify): one
    # all-reduce on this rank's device.  If that fails too, the bench ranks combine over gloo, host-staged.
    nccl_ok = 1
    if not ok or os.environ.get("TYR_BENCH_PREFLIGHT_PROBE_TORCH_NCCL"):
        try:
            g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=45))
            t = torch.ones(4, dtype=torch.float32, device=f"cuda:{local_rank}")
            dist.all_reduce(t, group=g)
            torch.cuda.synchronize()
            nccl_ok = int(float(t[0].item()) == float(world))
        except Exception as e:  # noqa: BLE001
            print(f"[bench preflight] rank {rank}: torch's nccl backend: {e!r}", file=sys.stderr)
            nccl_ok = 0
    flag = torch.tensor([ok, nccl_ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    return 0 if int(flag[0].item()) == 1 else (PREFLIGHT_TORCH_NCCL_ONLY if int(flag[1].item()) == 1 else 1)


def run_dist_preflight() -> int:
    """spawn this rank's pre-flight child (this process has not initialised the GPU yet) and wait for it, bounded:
    0 = the native exchange verified on every rank; PREFLIGHT_TORCH_NCCL_ONLY = it did not, torch's nccl backend does;
    1 = neither (or the child crashed / ran out of time)"""
    # the children make their own rendezvous through a FILE (no second port to find free and to agree on): one name per
    # launch -- the launcher's pid is the parent of every rank, its master port tells concurrent launches apart -- and
    # without the launcher's TORCHELASTIC_* variables, which would tell them that an agent already hosts a store
    store = os.path.join(tempfile.gettempdir(), f"tyr_preflight_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}")
    cmd = [sys.executable, os.path.abspath(__file__), "--dist-preflight", "--preflight-store", store]
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    try:
        p = subprocess.run(cmd, timeout=PREFLIGHT_TIMEOUT_S, env=env, stdout=subprocess.DEVNULL)
        return p.returncode if p.returncode in (0, PREFLIGHT_TORCH_NCCL_ONLY) else 1
    except subprocess.TimeoutExpired:
        print(f"[bench] rank {os.environ.get('RANK', '?')}: the native exchange's pre-flight did not finish in {PREFLIGHT_TIMEOUT_S:.0f} s", file=sys.stderr)
        return 1


def main():
    args = parse_args()
    if args.pmc_child:
        sys.exit(pmc_child(args))
    if args.dist_preflight:
        sys.exit(dist_preflight(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    spp_total, N = job_shape(args, world)

    # the counter passes run in child processes BEFORE this process initialises the GPU
    pmc = None
    if world == 1 and args.pmc == "auto":
        pmc = run_pmc_passes(args)
    if pmc is not None and args.save_pmc:
        with open(args.save_pmc, "w") as f:
            json.dump({"workload": args.workload, "resolution": f"{args.width}x{args.height}", "spp": spp_total, "queue_size": N, "kernel": pmc["kernel"], "counters": pmc["counters"],
                       "launches_averaged": pmc["launches_averaged"], "shade_counters_per_render": pmc.get("shade_counters_per_render", {}), "shade_launches_per_render": pmc.get("shade_launches_per_render"), "source": pmc["source"], "units": "per launch of the kernel, averaged over the launches of one warm render; FETCH_SIZE / WRITE_SIZE in KB"}, f, indent=1)
    if pmc is None and world == 1:
        pmc = committed_pmc(args.workload, N)

    # the native exchange is checked by a child of every rank BEFORE this process initialises its GPU (see dist_preflight)
    preflight_ok = True
    backend = args.backend  # what torch.distributed runs on: the asked-for backend, or gloo when RCCL does not work on this box at all
    backend_fallback = None
    want_native = world > 1 and args.combine_impl == "native" and (args.backend == "nccl" or bool(os.environ.get("TYR_BENCH_PREFLIGHT_ONE_DEVICE")))
    if want_native:
        pre = run_dist_preflight()
        preflight_ok = pre == 0
        if pre == 1 and args.backend == "nccl":
            # neither the library's RCCL exchange nor torch's nccl backend came through the children's check (an RCCL that cannot
            # initialise, two ranks on one device, a hang): the ranks combine over gloo, host-staged -- slower, and a line
            # instead of a crash.  Every rank's child reports the children's common verdict, so all ranks branch alike.
            backend = "gloo"
            backend_fallback = "pre-flight: neither the native exchange (tyr_dist_*) nor torch's nccl backend works on this box; the combine runs over gloo, host-staged"
            print("[bench] " + backend_fallback, file=sys.stderr)

    import numpy as np  # noqa: F401
    import torch

    from tyrant_amd import binding, scenes
    from tyrant_amd import dist as tdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.backend == "gloo":
        local_rank = 0  # rehearsal: all ranks share device 0
    local_rank %= max(torch.cuda.device_count(), 1)  # (a box with fewer GPUs than ranks: the gloo fallback of an nccl run shares devices)
    torch.cuda.set_device(local_rank)
    dist = tdist.init_process_group(backend) if world > 1 else None
    dev = f"cuda:{local_rank}"
    cdev = "cpu" if backend == "gloo" else dev  # where torch's collectives live

    W, H = args.width, args.height
    sc, nodes, prims, label, t_build = build_workload(args.workload, binding, scenes)
    flags = binding.TYR_FLAG_PROFILE | (binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0)
    shard = tdist.shard_spec(rank, world, H)
    tune = {k: int(v) for k, v in (kv.split("=") for kv in args.tune)}
    native = world > 1 and args.combine_impl == "native" and backend == "nccl"
    if native:
        # every rank's child must have verified, or nobody uses the native path (the ranks have to branch alike)
        flag = torch.tensor([1 if preflight_ok else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        native = bool(int(flag.item()))
        if not native and rank == 0:
            print("[bench] the native exchange did not pass its pre-flight: using torch.distributed for the combine", file=sys.stderr)
    mode = binding.TYR_DIST_GATHER if args.combine == "gather" else binding.TYR_DIST_REDUCE
    use_torch_gather = False
    if world > 1 and not native and args.combine == "gather":
        use_torch_gather = tdist.agree_gather_works(cdev)

    def measure(N, steps, warmup, spp, shard, ranks):
        """one renderer at queue size N: untimed counting render, warm-up, `steps` timed renders (ranks > 1: + the combine)"""
        # the caller owns blit_buffer (main.cpp:129-130): a torch tensor here (device memory is torch's job in this script)
        accum = torch.zeros(H * W * 4, dtype=torch.float32, device=dev)
        frame = torch.zeros(H * W * 4, dtype=torch.float32, device=dev) if (ranks > 1 and rank == 0) else None
        torch.cuda.synchronize()  # the library launches on its own stream
        r = binding.Renderer(W, H, N, device=local_rank, flags=flags, blit_buffer=accum.data_ptr(), **shard)
        r.load_scene(sc, nodes, prims)
        info = r.scene_info()
        upload = {"layout": round(info["upload_layout_s"], 6), "copy": round(info["upload_copy_s"], 6), "device_bytes": info["device_bytes"]}
        if tune:
            r.set_tuning(**tune)
        comm = None
        native_ok = [True]
        if ranks > 1 and native:
            # ncclCommInitRank is collective: first agree that every rank can open librccl at all
            ok, uid = 1, bytes(binding.TYR_DIST_ID_BYTES)
            try:
                uid = binding.dist_unique_id()
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: tyr_dist_unique_id failed ({e!r})", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            native_ok[0] = bool(int(flag.item()))
            if native_ok[0]:
                # the 128-byte ncclUniqueId travels over the process group that exists anyway; the communicator itself is the library's
                idt = torch.tensor(list(uid), dtype=torch.uint8, device=cdev)
                dist.broadcast(idt, src=0)
                comm = binding.Dist(r, bytes(idt.cpu().tolist()), rank, world)

        def step():
            # every step is the SAME job: the frame counter all seeds are built from (kernel.cu:667) restarts, so each timed render
            # casts exactly the rays of the render the oracle's counters were committed for (oracle_counters_match below)
            r.set_frame(1)
            r.reset_accum()
            it = r.render(spp)
            if ranks > 1:
                if comm is not None and native_ok[0]:
                    # pack on the render stream, ship on the communicator's: the next reset_accum / render overlaps it
                    comm.combine(frame.data_ptr() if frame is not None else None, mode=mode, root=0)
                else:
                    # (after a failed native check: the sum-reduce, which needs no probing)
                    combine = (lambda t: tdist.gather_rows(t, H, W, rank, world, dst=0)) if use_torch_gather else (lambda t: tdist.reduce_accum(t, dst=0))
                    if backend == "gloo":
                        host = accum.cpu()
                        combine(host)
                        if rank == 0:
                            frame.copy_(host)
                    else:
                        combine(accum)
                        torch.cuda.current_stream().synchronize()  # torch's stream: the next tyr_reset_accum must not zero what RCCL still reads
                        if rank == 0:
                            frame.copy_(accum)
            return it

        def fence():
            if comm is not None and native_ok[0]:
                comm.wait()
            if ranks > 1:
                dist.barrier()
            torch.cuda.synchronize()

        # untimed: nodes / triangles per ray and rays entering the tree, from the counting build (one render, this rank's shard)
        rc = binding.Renderer(W, H, N, device=local_rank, flags=flags | binding.TYR_FLAG_COUNT_VISITS, **shard)
        rc.load_scene(sc, nodes, prims)
        rc.render(spp)
        kc = rc.counters()
        ne, ns = max(kc["total_extend_rays"], 1), max(kc["total_shadow_rays"], 1)
        visits = {
            "nodes_per_ext": kc["nodes_extend"] / ne,
            "tris_per_ext": kc["tris_extend"] / ne,
            "nodes_per_con": kc["nodes_connect"] / ns,
            "tris_per_con": kc["tris_connect"] / ns,
            "visible_frac": kc["n_shadow_visible"] / ns,
            "in_tree_frac": (kc["rays_in_tree_extend"] + kc["rays_in_tree_connect"]) / (ne + ns),
            "in_tree_ext_frac": kc["rays_in_tree_extend"] / ne,
        }
        rc.close()

        if comm is not None:
            # The native exchange has only ever met one GPU per box before the driver's multi-GPU run: check one combined
            # frame (every pixel must hold exactly spp finished paths) and let ALL ranks fall back to torch.distributed
            # together if it does not -- a wrong or failing exchange must cost the path, not the measurement.
            ok = 1
            try:
                step()
                fence()
                if rank == 0:
                    a = frame.view(H * W, 4)[:, 3]
                    ok = int(float(a.min()) == float(a.max()) == float(spp))
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: native combine failed ({e!r})", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            native_ok[0] = bool(int(flag.item()))
            if not native_ok[0] and rank == 0:
                print("[bench] native combine did not verify: using torch.distributed for the exchange", file=sys.stderr)
        if warmup > 0:
            step()  # cold: the first launches load the code objects and size the persistent grids (~75 ms), not a timing of anything
        r.timings(reset=True)
        for _ in range(warmup):
            step()
        fence()
        # Every hipEvent pair between two kernels is ~10 us of idle GPU (1.7-3.4 % of a render with all stages
        # bracketed).  The warm-up renders time every stage (-> kernel_ms, per render); the timed region keeps only
        # the pair the roofline needs, around the extend stage.
        tm_all = r.timings() if warmup > 0 else None
        if tm_all is not None:
            r.set_tuning(profile_mask=(1 << 1) | (1 << 3))  # TYR_K_EXTEND (the trace launches) + TYR_K_CONNECT (merged renders: one launch per render); shade's time comes from the warm-up renders
        k0 = r.counters()
        r.timings(reset=True)
        fence()
        t0 = time.perf_counter()
        iters = 0
        for _ in range(steps):
            iters += step()
        fence()
        dt = time.perf_counter() - t0
        k1 = r.counters()
        tm = r.timings()
        assert k1["device_error"] == 0, k1
        ext = k1["total_extend_rays"] - k0["total_extend_rays"]
        shd = k1["total_shadow_rays"] - k0["total_shadow_rays"]
        survivors = k1["n_survive"] - k0["n_survive"]
        deltas = {f: k1[f] - k0[f] for f in ORACLE_COUNTER_FIELDS}
        stats = torch.tensor([float(ext), float(shd), dt], dtype=torch.float64, device=cdev)
        if ranks > 1:
            tmax = stats[2:3].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(stats[0:2], op=dist.ReduceOp.SUM)
            stats[2] = tmax[0]
        ext_all, shd_all, dt_all = (float(x) for x in stats.tolist())
        if rank == 0:
            # sanity of the combined frame: every pixel has exactly spp completed paths
            a = (frame if frame is not None else accum).view(H * W, 4)[:, 3]
            assert float(a.min()) == float(a.max()) == float(spp), (float(a.min()), float(a.max()), spp)
        native_used = comm is not None and native_ok[0]
        comm_info = None
        if comm is not None:
            try:
                comm_info = comm.info()  # ncclCommCount of the communicator the exchange ran on
            except Exception:  # noqa: BLE001
                comm_info = None
            comm.close()
            comm = True if native_used else None  # (only its truth value is reported below)
        r.close()
        per_render = {k: round(v["ms"] / warmup, 3) for k, v in tm_all.items()} if tm_all is not None else {k: round(v["ms"] / steps, 3) for k, v in tm.items()}
        return {"native_combine": bool(comm is not None and native_ok[0]), "comm_info": comm_info, "ext": ext, "shd": shd, "survivors": survivors, "ext_all": ext_all, "shd_all": shd_all, "dt_all": dt_all, "iters": iters, "tm": tm, "kernel_ms_per_render": per_render, "counter_deltas": deltas, "upload": upload, **visits}

    m = measure(N, args.steps, args.warmup, spp_total, shard, world)
    mref = None
    if not args.no_reference_queue and N != REF_N:
        mref = measure(REF_N, max(1, min(args.steps, 2)), 1, spp_total, shard, world)
    # strong scaling: the SAME job (spp_total over the whole frame) on ONE GPU, rank 0 alone; the others wait at the barrier
    solo = None
    if world > 1 and args.scaling == "strong" and not args.no_one_gpu_job:
        if rank == 0:
            solo = measure(min(spp_total * W * H, 1 << 25), max(1, min(args.steps, 2)), 1, spp_total, tdist.shard_spec(0, 1, H), 1)
        dist.barrier()

    steady = None
    if world == 1 and not args.no_steady_state:
        steady = steady_state(binding, sc, nodes, prims, W, H, N, flags, local_rank)

    if rank == 0:
        mrays = (m["ext_all"] + m["shd_all"]) / m["dt_all"] / 1e6
        tm = m["tm"]
        out = {
            "metric": "Mrays/s at 1080p 8spp" if (world == 1 and (W, H, spp_total) == (1920, 1080, 8)) else f"Mrays/s at {W}x{H} {spp_total}spp",
            "value": round(mrays, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(m["dt_all"] / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": label + (f"; as BASELINE config C4: {spp_total} spp in total shared by {world} GPUs" if world > 1 and args.scaling == "strong" else ""),
                "resolution": f"{W}x{H}",
                "spp_total": spp_total,
                "queue_size": N,
                "triangles": int(prims.shape[0]),
                "bvh_nodes": int(nodes.shape[0]),
                "sharding": (f"rows y % {world} == rank; " + (("tyr_dist_combine (RCCL behind the C ABI): " + ("ncclSend/ncclRecv of each rank's packed rows, double-buffered" if args.combine == "gather" else "ncclReduce(sum) of the full buffers"))
                                                            if m["native_combine"] else ("torch.distributed gather of each rank's rows" if use_torch_gather else "torch.distributed reduce(sum) of the accumulation buffer"))) if world > 1 else "none",
                "backend": (backend if backend_fallback is None else f"{backend} (asked for {args.backend})") if world > 1 else None,
                **({"combine": {"native_combine": bool(m["native_combine"]), "form": args.combine if m["native_combine"] else ("gather" if use_torch_gather else "reduce"),
                                "rccl_comm_ranks": (m.get("comm_info") or {}).get("comm_ranks"),
                                "fallback_reason": None if m["native_combine"] else (backend_fallback if backend_fallback else ("--combine-impl torch / gloo backend" if not want_native else ("pre-flight of the native exchange failed" if not preflight_ok else "the native exchange did not verify on the real job"))),
                                "measured_on_hardware_before": "no: the N > 1 RCCL exchange had never run when this was written (one GPU per test box)"}} if world > 1 else {}),
                "wavefront_iterations_per_step": m["iters"] / args.steps,
                "extend_Mrays/s": round(m["ext_all"] / m["dt_all"] / 1e6, 3),
                "shadow_Mrays/s": round(m["shd_all"] / m["dt_all"] / 1e6, 3),
                "in_tree_Mrays/s": round(mrays * m["in_tree_frac"], 3),
                "in_tree_fraction": {"all_rays": round(m["in_tree_frac"], 4), "extend_rays": round(m["in_tree_ext_frac"], 4), "note": "rays whose test of the root box passes (counting build, rank 0's shard); the others cost one box test"},
                "host_bvh_build_s": round(t_build, 6),
                "host_scene_upload_s": {**m["upload"], "note": "tyr_scene_upload of the timed renderer (Scene.cpp:53-67's upload half), outside the timed region: `layout` = the host's re-layout of the reference's node array as quad nodes + 48-byte triangles on the builder's threads, `copy` = device allocation + the copies to HBM"},
                **oracle_counters_check(args, world, W, H, spp_total, N, int(prims.shape[0]), m),
                "render_path": "tyr_render defaults: merged traversal launches (extend(i + 1) + connect(i)), sphere halves folded into shade, rays whose fate is known where they are made (camera rays / survivors that hit nothing, shadow rays that cannot reach a triangle) finished in place -- they count as rays, they never enter a queue; launch-per-iteration (TYR_TUNE_STREAM_TAIL = 0)",
                **({"tuning": tune} if tune else {}),
                **({"steady_state": steady} if steady else {}),
                **(
                    {
                        "reference_queue_size": {
                            "queue_size": REF_N,
                            "Mrays/s": round((mref["ext_all"] + mref["shd_all"]) / mref["dt_all"] / 1e6, 3),
                            "wavefront_iterations_per_step": mref["iters"] / max(1, min(args.steps, 2)),
                            "extend_avg_launch_ms": round(mref["tm"]["extend"]["ms"] / max(mref["tm"]["extend"]["launches"], 1), 4),
                        }
                    }
                    if mref
                    else {}
                ),
            },
            # merged renders: EVERY traversal launch is k_trace_flat -- the ones timed as the extend stage and the one that ends a
            # render with the last iteration's shadow rays (timed as the connect stage): one kernel, one average
            "roofline": (roofline_block(pmc, tm["extend"]["ms"] + tm["connect"]["ms"], tm["extend"]["launches"] + tm["connect"]["launches"], m["ext"], m, m["kernel_ms_per_render"], kernel=TRACE_KERNEL, con_ms=0.0, shadow_rays=m["shd"],
                                        quad=(quad_block(args) if world == 1 and args.pmc != "off" else None), renders=args.steps)
                         if dominant_kernel(args.tune) == TRACE_KERNEL else
                         roofline_block(pmc, tm["extend"]["ms"], tm["extend"]["launches"], m["ext"], m, m["kernel_ms_per_render"], kernel=EXTEND_KERNEL, con_ms=tm["connect"]["ms"], shadow_rays=m["shd"])),
        }
        per_render = 1.0 / args.steps
        k_sh = tm["shade"]
        out["roofline"]["shade"] = shade_block(pmc, k_sh["ms"] * per_render if k_sh["launches"] else m["kernel_ms_per_render"].get("shade", 0.0), m["ext"] * per_render, m["survivors"] * per_render, m["shd"] * per_render)
        if world == 1 and args.pmc != "off":
            d = drain_block(args)
            if d:
                out["roofline"].update(d)
        if solo is not None:
            nsteps_solo = max(1, min(args.steps, 2))
            t1, tn = solo["dt_all"] / nsteps_solo, m["dt_all"] / args.steps
            out["config"]["strong_scaling"] = {
                "job": f"{spp_total} spp over the whole {W}x{H} frame",
                "one_gpu_ms": round(t1 * 1e3, 3),
                "one_gpu_Mrays/s": round((solo["ext_all"] + solo["shd_all"]) / solo["dt_all"] / 1e6, 3),
                f"{world}_gpu_ms": round(tn * 1e3, 3),
                "speedup_vs_1gpu": round(t1 / tn, 3),
                "efficiency_vs_1gpu": round(t1 / tn / world, 4),
                "note": "rank 0 renders the same job alone after the timed region (same process, same build)",
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, W, H, N, args.cpu_iterations, sc.triangle_materials, spp_total)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sc, W, H, N, iterations, tri_materials, spp):
    """the CPU baseline leg (oracle/cpu_baseline.py: the oracle and, where built, the reference's own traversal and builder on
    one host core) -- the only place this script touches oracle/, after the timed region"""
    from oracle.cpu_baseline import cpu_baseline as run

    return run(sc, W, H, N, iterations, tri_materials, spp)


if __name__ == "__main__":
    main()
