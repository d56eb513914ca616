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
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# what is not the measurement itself lives in tyrant_amd/benchkit/ (child-process modes, counters -> roofline arithmetic, the
# pre-flight of the native exchange); re-exported here: tests/test_bench_contract.py and the tools address them as bench.<name>
from tyrant_amd.benchkit.children import committed_pmc, drain_block, pmc_child, quad_block, run_pmc_passes  # noqa: E402,F401
from tyrant_amd.benchkit.common import (EXTEND_KERNEL, HBM_PEAK_GBS, NUM_CU, NUM_SIMD, NUM_XCD, PMC_PASSES, REF_N, SHADE_BYTES_PER_RAY, SHADE_KERNEL, TRACE_KERNEL,  # noqa: E402,F401
                                        build_workload, dominant_kernel, find_rocprof, job_shape)
from tyrant_amd.benchkit.preflight import PREFLIGHT_TIMEOUT_S, PREFLIGHT_TORCH_NCCL_ONLY, dist_preflight, run_dist_preflight  # noqa: E402,F401
from tyrant_amd.benchkit.roofline import ORACLE_COUNTER_FIELDS, nominal_step_frac, oracle_counters_check, roofline_block, shade_block  # noqa: E402,F401


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=["c1", "c2", "c3", "c3_framed", "c5"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel: 0 = the configuration's (8 at one GPU; N > 1: 64 in total when --scaling strong, 8 per GPU when weak)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1: strong = BASELINE config C4, 64 spp in total whatever N; weak = 8*N spp")
    ap.add_argument("--queue", type=int, default=0, help="ray_queue_buffer_size (variables.h:44); 0 = sized for the GPU: spp x local pixels, at most 32 Mi slots")
    ap.add_argument("--no-reference-queue", action="store_true", help="skip the second measurement at the reference's queue size (2,097,152)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-framed", action="store_true", help="workload c3 at one GPU: skip the secondary workload c3_framed (config.framed: the same scene from where the room's opening fills the frame)")
    ap.add_argument("--no-spread", action="store_true", help="one GPU: skip config.spread (ten blocks of twenty more renders behind the timed region: ms per render p50 / p95 / max)")
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

    def pct(v, p):
        v = sorted(v)
        return v[min(len(v) - 1, int(round(p * (len(v) - 1))))] if v else None

    def measure(N, steps, warmup, spp, shard, ranks, sc=sc, spread_blocks=0):
        """one renderer at queue size N: untimed counting render, warm-up, `steps` timed renders (ranks > 1: + the combine);
        spread_blocks: that many blocks of twenty more renders BEHIND the timed region, each render timed by itself"""
        # the caller owns blit_buffer (main.cpp:129-130): a torch tensor here (device memory is torch's job in this script)
        accum = torch.zeros(H * W * 4, dtype=torch.float32, device=dev)
        frame = torch.zeros(H * W * 4, dtype=torch.float32, device=dev) if (ranks > 1 and rank == 0) else None
        torch.cuda.synchronize()  # the library launches on its own stream
        r = binding.Renderer(W, H, N, device=local_rank, flags=flags, blit_buffer=accum.data_ptr(), **shard)
        r.load_scene(sc, nodes, prims)
        info = r.scene_info()
        upload = {"layout": round(info["upload_layout_s"], 6), "copy": round(info["upload_copy_s"], 6), "layout_on": "device" if info["layout_on_device"] else "host", "device_bytes": info["device_bytes"]}
        if info["layout_on_device"]:  # ... and the host pass on the same arrays, for the line
            upload["host_layout"] = round(binding.layout_probe(nodes, prims, want_pairs=False)["seconds"], 6)
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
        step_ms = []  # (tyr_render returns when the render's counters are on the host: a step's end needs no extra wait to be timed)
        ts = t0
        for _ in range(steps):
            iters += step()
            tn = time.perf_counter()
            step_ms.append((tn - ts) * 1e3)
            ts = tn
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
        # what the last timed render left in the accumulation buffer (every step resets it and renders the same job): the sums the committed
        # oracle file holds for this job are compared with it below (oracle_counters_check: oracle_radiance_match)
        radiance = [float(x) for x in accum.view(H * W, 4)[:, :3].to(torch.float64).sum(0).tolist()] if ranks == 1 else None
        spread = None
        if spread_blocks > 0 and ranks == 1:
            # behind the timed region: the job is 5 ms and boxes differ by 1-3 %; K steps say little about a render's spread
            per, blocks = [], []
            for _ in range(spread_blocks):
                tb = time.perf_counter()
                for _ in range(20):
                    t1 = time.perf_counter()
                    step()
                    per.append((time.perf_counter() - t1) * 1e3)
                blocks.append((time.perf_counter() - tb) * 1e3 / 20.0)
            spread = {"renders": len(per), "ms_min": round(min(per), 3), "ms_p50": round(pct(per, 0.5), 3), "ms_p95": round(pct(per, 0.95), 3), "ms_max": round(max(per), 3),
                      "block_means_ms": [round(b, 3) for b in blocks], "note": f"{spread_blocks} blocks of 20 renders of the same job behind the timed region, each render timed by itself (host clock around tyr_render, which returns with the render's counters on the host)"}
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
        return {"native_combine": bool(comm is not None and native_ok[0]), "comm_info": comm_info, "ext": ext, "shd": shd, "survivors": survivors, "ext_all": ext_all, "shd_all": shd_all, "dt_all": dt_all, "iters": iters, "tm": tm, "kernel_ms_per_render": per_render, "counter_deltas": deltas, "upload": upload, "step_ms": step_ms, "spread": spread, "radiance_sum_rgb": radiance, **visits}

    m = measure(N, args.steps, args.warmup, spp_total, shard, world, spread_blocks=0 if (world > 1 or args.no_spread) else 10)
    # the secondary workload (NOT the metric): the same scene and job from where the room's opening fills the frame
    framed = None
    if world == 1 and args.workload == "c3" and not args.no_framed and (W, H, spp_total) == (1920, 1080, 8):
        fsteps = max(1, min(args.steps, 10))
        mf = measure(N, fsteps, 1, spp_total, shard, 1, sc=scenes.mesh_scene_framed(706))
        fr_rate = (mf["ext_all"] + mf["shd_all"]) / mf["dt_all"] / 1e6
        framed = {"workload": "c3_framed: the C3 scene from (0, -87.5, 50) -- kernel.cu:698-699's 1.5 x W/H by 1.5 extents make the frame exactly as wide as the room's opening there: every camera ray enters the room (from SURVEY.md 8d's (0, -190, 50) the opening covers 12.7 % of the frame)",
                  "Mrays/s": round(fr_rate, 3), "ms_per_step": round(mf["dt_all"] / fsteps * 1e3, 3), "steps": fsteps, "in_tree_Mrays/s": round(fr_rate * mf["in_tree_frac"], 3),
                  "in_tree_fraction": {"all_rays": round(mf["in_tree_frac"], 4), "extend_rays": round(mf["in_tree_ext_frac"], 4)},
                  "wavefront_iterations_per_step": mf["iters"] / fsteps, "nodes_per_extend_ray": round(mf["nodes_per_ext"], 2), "tris_per_extend_ray": round(mf["tris_per_ext"], 3),
                  "kernel_ms_per_render": mf["kernel_ms_per_render"],
                  **oracle_counters_check(args, 1, W, H, spp_total, N, int(prims.shape[0]), mf, workload="c3_framed", steps=fsteps)}
    mref = None
    if not args.no_reference_queue and N != REF_N:
        mref = measure(REF_N, max(1, min(args.steps, 2)), 1, spp_total, shard, world)
    # strong scaling: the SAME job (spp_total over the whole frame) on ONE GPU, rank 0 alone; the others wait at the barrier
    solo = None
    if world > 1 and args.scaling == "strong" and not args.no_one_gpu_job:
        if rank == 0:
            solo = measure(min(spp_total * W * H, 1 << 25), max(1, min(args.steps, 2)), 1, spp_total, tdist.shard_spec(0, 1, H), 1)
        dist.barrier()

    # the same tree built on the device (tyr_bvh_build_device: the reference's SAH build as kernels, hip/bvh_build_dev.hip), beside
    # the host builder's time above: outside the timed region, bytes compared
    dev_build = None
    if world == 1 and rank == 0:
        try:
            binding.bvh_build_device(sc.triangles[:1024], device=local_rank)  # (code objects)
            dn, dp, sec = binding.bvh_build_device(sc.triangles, device=local_rank)
            dev_build = {"device": round(sec[0], 6), "copies": round(sec[1], 6), "same_bytes_as_host_build": bool(dn.tobytes() == nodes.tobytes() and dp.tobytes() == prims.tobytes()),
                         "note": "tyr_bvh_build_device: host arrays in and out; `device` = first kernel to last (hipEvents), `copies` = triangles + boxes in, nodes + reordered triangles out"}
            # both halves of Scene::Load in one call, nothing but triangles crossing the bus (tyr_scene_build_upload): a ctx of its own
            r2 = binding.Renderer(64, 48, 64 * 48, device=local_rank, flags=flags)
            try:
                r2.build_upload(sc.triangles[:1024], want_nodes=False)
                _, p2, s3 = r2.build_upload(sc.triangles, want_nodes=False)
                h2 = r2.scene_hash()
                r2.upload(nodes, prims)
                h1 = r2.scene_hash()
                dev_build["build_upload"] = {"build": round(s3[0], 6), "layout": round(s3[1], 6), "copies": round(s3[2], 6),
                                             "same_scene_in_hbm_as_build_then_upload": bool(p2.tobytes() == prims.tobytes() and all(h1[k] == h2[k] for k in h1 if k != "seconds")),
                                             "note": "tyr_scene_build_upload: the tree built AND laid out on the device, the nodes never leave HBM; `copies` = triangles + boxes in, reordered triangles out"}
            finally:
                r2.close()
        except Exception as e:  # noqa: BLE001
            dev_build = {**(dev_build or {}), "error": repr(e)}

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
            "ms_per_step_spread": {"min": round(min(m["step_ms"]), 3), "p50": round(pct(m["step_ms"], 0.5), 3), "p95": round(pct(m["step_ms"], 0.95), 3), "max": round(max(m["step_ms"]), 3), "of": "the timed steps (host clock per tyr_render)"},
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": label + (f"; as BASELINE config C4: {spp_total} spp in total shared by {world} GPUs" if world > 1 and args.scaling == "strong" else ""),
                # what `value` is made of, first: the rays that actually enter the tree (the others cost one box test where they are made)
                "in_tree_Mrays/s": round(mrays * m["in_tree_frac"], 3),
                "in_tree_fraction": {"all_rays": round(m["in_tree_frac"], 4), "extend_rays": round(m["in_tree_ext_frac"], 4), "note": "rays whose test of the root box passes (counting build, rank 0's shard); the others cost one box test.  SURVEY.md 8d's camera sees the room's opening in 12.7 % of the frame: config.framed is the same job from where it fills the frame"},
                **({"framed": framed} if framed else {}),
                **({"spread": m["spread"]} if m.get("spread") else {}),
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
                "host_bvh_build_s": round(t_build, 6),
                **({"device_bvh_build_s": dev_build} if dev_build else {}),
                "host_scene_upload_s": {**m["upload"], "note": "tyr_scene_upload of the timed renderer (Scene.cpp:53-67's upload half), outside the timed region.  layout_on device (TYR_TUNE_LAYOUT_ON_DEVICE, the default): `copy` = the reference's node and triangle arrays to HBM as they are, `layout` = hip/bvh_layout_dev.hip making quad nodes + 48-byte triangles there (the same bytes), `host_layout` = the host pass on the builder's threads, for comparison; layout_on host: `layout` = that host pass, `copy` = allocation + the finished records to HBM"},
                **oracle_counters_check(args, world, W, H, spp_total, N, int(prims.shape[0]), m),
                "render_path": "tyr_render defaults: merged traversal launches (extend(i + 1) + connect(i)), sphere halves folded into shade, rays whose fate is known where they are made (camera rays / survivors that hit nothing, shadow rays that cannot reach a triangle) finished in place -- they count as rays, they never enter a queue",
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
        nf = nominal_step_frac(out["roofline"], args.steps, m["counter_deltas"]["total_primary_rays"], m["dt_all"])
        if nf:
            out["roofline"].update(nf)
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
