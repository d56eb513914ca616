#!/usr/bin/env python3
"""bench.py -- Mrays/s of the wavefront path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torchrun, one rank per GPU)

One step = one complete render of the workload through the C ABI (tyr_reset_accum +
tyr_render: top-up -> extend -> shade -> connect until every path has finished) plus, for
N > 1, the RCCL sum-reduction of the accumulation buffer onto rank 0.  The scene lives in HBM
before the timed region starts; nothing crosses PCIe inside it except 120-byte counter reads.

Workloads (BASELINE.json configs):
    c2  Cornell box + 10,000 seeded random diffuse triangles, 1920x1080, 8 spp     (default: configs[1])
    c3  room + 706x706 height-field mesh (996,882 triangles), 30 % SPEC, 1920x1080, 8 spp
Queue size: the reference's ray_queue_buffer_size (2,097,152, variables.h:44) was chosen for a small
GPU and forces 20 thin wavefront iterations per 8-spp frame.  It is a runtime parameter here, and the
headline run sizes it for the GPU -- spp x pixels slots (16.6 M, 2.5 GB of 288 GB), i.e. every primary of
the render in flight at once and 6 fat iterations.  The same workload at the reference's queue size is
measured too and reported under config.reference_queue_size.

At N GPUs the frame is pixel-sharded (rows y % N == rank) and rendered at 8*N spp, so each
GPU traces what one GPU traces at N = 1: weak scaling (N = 8 is BASELINE config C4's 64 spp).

Mrays/s = (extend rays + shadow rays traced by all ranks) / wall time.
roofline: the extend kernel; achieved = algorithmic bytes / hipEvent time of its launches
inside the timed region, algorithmic bytes per ray = 24 + 8 + 32 * nodes + 36 * triangles
with nodes / triangles per ray counted by the library's counting build of the same kernel
in an untimed pass.  cpu_baseline: the oracle's serial loop on this host, 1 core, on the
first wavefront iterations of the same workload (N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=8, help="samples per pixel per GPU (total spp = spp * gpus)")
    ap.add_argument("--queue", type=int, default=0, help="ray_queue_buffer_size (variables.h:44); 0 = sized for the GPU: spp x local pixels, i.e. every primary ray of the render in flight at once (16.6 M slots = 2.5 GB of the 288 GB)")
    ap.add_argument("--no-reference-queue", action="store_true", help="skip the second measurement at the reference's queue size (2,097,152)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iterations", type=int, default=3)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearse the multi-rank path with every rank on one GPU (the reduce then goes through host memory)")
    ap.add_argument("--combine", default="gather", choices=["gather", "reduce"], help="N > 1: how rank 0 gets the frame -- gather of the rows each rank owns (1/N of the frame per rank; falls back to the reduce if a probe of dist.gather fails on any rank) or sum-reduce of the full buffers")
    ap.add_argument("--tune", action="append", default=[], help="launch-shape knob of tyr_set_tuning, e.g. --tune shade_tiles=2 (never changes results)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from tyrant_amd import binding, scenes
    from tyrant_amd import dist as tdist

    rank, local_rank, world = tdist.env_rank_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 needs `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`")
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.backend == "gloo":
        local_rank = 0  # rehearsal: all ranks share device 0
    torch.cuda.set_device(local_rank)
    dist = tdist.init_process_group(args.backend) if world > 1 else None

    W, H = args.width, args.height
    spp_total = args.spp * world
    N = args.queue if args.queue > 0 else min(args.spp * W * H, 1 << 25)
    sc, nodes, prims, label, t_build = build_workload(args.workload, binding, scenes)
    flags = binding.TYR_FLAG_PROFILE | (binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0)
    shard = tdist.shard_spec(rank, world, H)
    # how rank 0 gets the frame: the ranks' own rows by point-to-point gather when that works everywhere, else the sum
    use_gather = False
    if world > 1 and args.combine == "gather":
        use_gather = tdist.agree_gather_works("cpu" if args.backend == "gloo" else f"cuda:{local_rank}")

    def measure(N, steps, warmup):
        """one renderer at queue size N: untimed counting render, warm-up, `steps` timed renders"""
        # the caller owns blit_buffer (main.cpp:129-130); here it is a torch tensor so RCCL can reduce it in place
        accum = torch.zeros(H * W * 4, dtype=torch.float32, device=f"cuda:{local_rank}")
        torch.cuda.synchronize()  # the library launches on its own stream
        r = binding.Renderer(W, H, N, device=local_rank, flags=flags, blit_buffer=accum.data_ptr(), **shard)
        r.load_scene(sc, nodes, prims)
        if tune:
            r.set_tuning(**tune)

        def step():
            r.reset_accum()
            it = r.render(spp_total)
            if world > 1:
                combine = (lambda t: tdist.gather_rows(t, H, W, rank, world, dst=0)) if use_gather else (lambda t: tdist.reduce_accum(t, dst=0))
                if args.backend == "gloo":
                    host = accum.cpu()
                    combine(host)
                    if rank == 0:
                        accum.copy_(host)
                else:
                    combine(accum)
                    # the collective is enqueued on torch's stream, the library renders on its own: the next step's
                    # tyr_reset_accum must not zero the buffer while RCCL still reads it
                    torch.cuda.current_stream().synchronize()
            return it

        def fence():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        # untimed: nodes / triangles per ray from the counting build of the same kernels (one render, this rank's shard)
        rc = binding.Renderer(W, H, N, device=local_rank, flags=flags | binding.TYR_FLAG_COUNT_VISITS, **shard)
        rc.load_scene(sc, nodes, prims)
        rc.render(spp_total)
        kc = rc.counters()
        visits = {
            "nodes_per_ext": kc["nodes_extend"] / max(kc["total_extend_rays"], 1),
            "tris_per_ext": kc["tris_extend"] / max(kc["total_extend_rays"], 1),
            "nodes_per_con": kc["nodes_connect"] / max(kc["total_shadow_rays"], 1),
            "tris_per_con": kc["tris_connect"] / max(kc["total_shadow_rays"], 1),
        }
        rc.close()

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
            r.set_tuning(profile_mask=1 << 1)  # TYR_K_EXTEND
        k0 = r.counters()
        r.timings(reset=True)
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
        stats = torch.tensor([float(ext), float(shd), dt], dtype=torch.float64, device="cpu" if args.backend == "gloo" else f"cuda:{local_rank}")
        if world > 1:
            tmax = stats[2:3].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(stats[0:2], op=dist.ReduceOp.SUM)
            stats[2] = tmax[0]
        ext_all, shd_all, dt_all = (float(x) for x in stats.tolist())
        if rank == 0:
            # sanity of the reduced frame: every pixel has exactly spp_total completed paths
            a = accum.view(H * W, 4)[:, 3]
            assert float(a.min()) == float(a.max()) == float(spp_total), (float(a.min()), float(a.max()), spp_total)
        r.close()
        per_render = {k: round(v["ms"] / warmup, 3) for k, v in tm_all.items()} if tm_all is not None else {k: round(v["ms"] / steps, 3) for k, v in tm.items()}
        return {"ext": ext, "ext_all": ext_all, "shd_all": shd_all, "dt_all": dt_all, "iters": iters, "tm": tm, "kernel_ms_per_render": per_render, **visits}

    tune = {k: int(v) for k, v in (kv.split("=") for kv in args.tune)}
    m = measure(N, args.steps, args.warmup)
    ext, ext_all, shd_all, dt_all, iters, tm = m["ext"], m["ext_all"], m["shd_all"], m["dt_all"], m["iters"], m["tm"]
    nodes_per_ext, tris_per_ext, nodes_per_con, tris_per_con = m["nodes_per_ext"], m["tris_per_ext"], m["nodes_per_con"], m["tris_per_con"]
    REF_N = 2097152  # variables.h:44
    mref = None
    if not args.no_reference_queue and N != REF_N:
        mref = measure(REF_N, max(1, min(args.steps, 2)), 1)

    if rank == 0:
        mrays = (ext_all + shd_all) / dt_all / 1e6
        # roofline of the dominant kernel (extend), this rank's launches inside the timed region
        bytes_per_ext = 24 + 8 + 32 * nodes_per_ext + 36 * tris_per_ext
        ext_ms, ext_launches = tm["extend"]["ms"], max(tm["extend"]["launches"], 1)
        bytes_per_launch = bytes_per_ext * ext / ext_launches
        achieved = (bytes_per_launch / (ext_ms / ext_launches * 1e-3)) / 1e9 if ext_ms > 0 else 0.0
        # HBM bytes per extend launch from rocprofv3 PMC passes of this same command (tools/pmc_traffic.sh); bench.py
        # cannot run under a profiler itself, so the committed measurement is attached when it matches the run
        traffic, traffic_detail = None, None
        try:
            with open(os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")) as f:
                tj = json.load(f)
            if tj["queue_size"] == N and world == 1:
                # same unit as `achieved`: bytes per launch / average launch duration
                traffic = round(tj["hbm_bytes_per_launch_corrected"] / (ext_ms / ext_launches * 1e-3) / 1e9, 2)
                traffic_detail = {"hbm_bytes_per_launch": round(tj["hbm_bytes_per_launch_corrected"]), "algorithmic_bytes_per_launch": round(bytes_per_launch), "source": tj["source"]}
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "Mrays/s at 1080p 8spp",
            "value": round(mrays, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt_all / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": label,
                "resolution": f"{W}x{H}",
                "spp_total": spp_total,
                "spp_per_gpu": args.spp,
                "queue_size": N,
                "triangles": int(prims.shape[0]),
                "bvh_nodes": int(nodes.shape[0]),
                "sharding": (f"rows y % {world} == rank, " + ("RCCL gather of each rank's rows onto rank 0" if use_gather else "RCCL reduce of the accumulation buffer")) if world > 1 else "none",
                "wavefront_iterations_per_step": iters / args.steps,
                "extend_Mrays/s": round(ext_all / dt_all / 1e6, 3),
                "shadow_Mrays/s": round(shd_all / dt_all / 1e6, 3),
                "host_bvh_build_s": round(t_build, 3),
                **({"tuning": tune} if tune else {}),
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
            "roofline": {
                "kernel": "extend stage = k_extend_spheres (sphere pre-pass) + k_extend_flat, one hipEvent pair around both",
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_detail": traffic_detail,
                "algorithmic_bytes_per_ray": round(bytes_per_ext, 1),
                "nodes_per_ray": round(nodes_per_ext, 2),
                "tris_per_ray": round(tris_per_ext, 3),
                "avg_launch_ms": round(ext_ms / ext_launches, 4),
                "launches": ext_launches,
                "connect_nodes_per_ray": round(nodes_per_con, 2),
                "connect_tris_per_ray": round(tris_per_con, 3),
                "kernel_ms_per_render": m["kernel_ms_per_render"],  # all stages: from the warm-up render(s); inside the timed region only extend is bracketed
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, W, H, N, args.cpu_iterations if N <= 4 * REF_N else 2, sc.triangle_materials, spp_total)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sc, W, H, N, iterations, tri_materials, spp):
    """the oracle (a serial CPU port of the reference's loop) on the first `iterations` wavefront
    iterations of the same workload: same scene, resolution, queue size, seeds; 1 core"""
    from oracle import pyorc
    from tyrant_amd import scenes

    t0 = time.perf_counter()
    nodes, prims = pyorc.bvh_build(sc.triangles, scenes.triangle_bboxes(sc.triangles))
    t_build = time.perf_counter() - t0
    o = pyorc.Oracle(W, H, N, flags=1 if tri_materials else 0)
    o.load_scene(sc, nodes, prims)
    o.set_budget(spp * W * H)  # the same primary-ray budget as the timed render: no top-up once it is spent
    t0 = time.perf_counter()
    for _ in range(iterations):
        o.launch_kernels()
    dt = time.perf_counter() - t0
    k = o.counters()
    rays = k["total_extend_rays"] + k["total_shadow_rays"]
    out = {
        "value": round(rays / dt / 1e6, 4),
        "unit": "Mrays/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {iterations} wavefront iterations of the same workload ({k['total_extend_rays']} extend + {k['total_shadow_rays']} shadow rays) in {dt:.1f} s",
        "bvh_build_s": round(t_build, 3),
        "host_cpus": os.cpu_count(),
    }
    # The reference's own traversal, where it can be had: oracle/_ref/libref_traverse.so is CachedBVH::intersect
    # (bvh.h:118-161) compiled from the reference's header in the authoring container (the builder and the kernels
    # cannot be built there).  Timed on the rays the next iteration would trace, through the same BVH, 1 core.
    R = pyorc.ref()
    if R is not None:
        import ctypes

        import numpy as np

        o.stage("begin"), o.stage("primary")
        n = min(o.counters()["n_live"], 1 << 21)
        q = np.ascontiguousarray(o.ray_queue(0, n))
        q["distance"] = 1e20  # VERY_FAR, variables.h:13: extend starts every ray there
        hit = np.zeros(n, dtype=np.int32)
        nd, pr = np.ascontiguousarray(nodes), np.ascontiguousarray(prims)
        t0 = time.perf_counter()
        R.ref_bvh_intersect(nd.ctypes.data_as(ctypes.c_void_p), pr.ctypes.data_as(ctypes.c_void_p), q.ctypes.data_as(ctypes.c_void_p), n, hit.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), None)
        dtr = time.perf_counter() - t0
        out["reference_trace"] = {
            "value": round(n / dtr / 1e6, 4),
            "unit": "Mrays/s",
            "cores": 1,
            "kind": "reference",
            "sample": f"CachedBVH::intersect of the reference's bvh.h over the first {n} rays of iteration {iterations + 1}'s queue (BVH only, no spheres, no shading) in {dtr:.1f} s; {int(hit.sum())} hits",
        }
    return out


if __name__ == "__main__":
    main()
