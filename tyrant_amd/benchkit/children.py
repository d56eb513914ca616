"""bench.py's child-process modes for N = 1, all run BEFORE the parent touches the GPU (or on an instrumented build of the library, never the
timed one): `rocprofv3 --pmc <counters> -- python3 bench.py --pmc-child ...` -- the program itself after `--`, no tracing domain combined with
--pmc, one pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one) --, the launch anatomy (-DTYR_LAUNCH_ANATOMY) and the quad-step
counts (-DTYR_QUAD_STATS)."""
from __future__ import annotations

import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

from .common import BENCH, PMC_PASSES, ROOT, SHADE_KERNEL, build_workload, dominant_kernel, find_rocprof, job_shape


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

def run_pmc_passes(args, timeout_s: float = 150.0):
    """-> {"counters": {name: average per launch of the production extend kernel in the LAST render}, "launches": n} or None"""
    rocprof = find_rocprof()
    if rocprof is None:
        return None
    out_root = tempfile.mkdtemp(prefix="tyr_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    child = [sys.executable, BENCH, "--pmc-child", "--workload", args.workload, "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp),
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

def drain_block(args):
    """How much of a traversal launch is its drain: an instrumented build of the library (-DTYR_LAUNCH_ANATOMY: three
    s_memrealtime stamps per wave) renders the workload once in a CHILD process; per launch, `feed` = first wave's start ->
    first wave to find the queue used up, `drain` = from there to the last wave's exit."""
    lib = os.path.join(ROOT, "tyrant_amd", "lib", "libtyrant_hip_anatomy.so")
    if not os.path.exists(lib):
        return None
    child = [sys.executable, BENCH, "--pmc-child", "--workload", args.workload, "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp), "--queue", str(args.queue)]
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
    child = [sys.executable, BENCH, "--pmc-child", "--workload", args.workload, "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp), "--queue", str(args.queue)]
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
