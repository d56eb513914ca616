"""counters -> the roofline / shade / oracle-counters blocks of bench.py's JSON line (pure arithmetic: tests/test_bench_contract.py)"""
from __future__ import annotations

import json
import os

from .common import EXTEND_KERNEL, GATHER_CEILING_GBS, HBM_PEAK_GBS, NUM_SIMD, NUM_XCD, ROOT, SHADE_BYTES_PER_RAY, TRACE_KERNEL

ORACLE_COUNTER_FIELDS = ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible")


def oracle_counters_check(args, world, W, H, spp, N, n_tris, m, workload=None, steps=None):
    """config.oracle_counters_match: the counter deltas of the TIMED renders against the oracle's counters for this very job,
    committed as tests/golden/bench_<workload>_counters.json (made by tests/golden/make_bench_counters.py: orc_render, the serial C
    restatement of kernel.cu:664-748; round 6: also for c2, c5 at 4K / 16 spp and the framed view of c3).  Every timed step restarts the frame counter, so K steps must have cast exactly K times
    the oracle's rays -- extend, shadow, survivors, visible shadow rays, iterations.  None when the job is not the committed one
    (another workload, resolution, spp, queue size or rank count).  The file is data: nothing under oracle/ is loaded here.
    `workload` / `steps`: the secondary workload's (c3_framed: tests/golden/bench_c3_framed_counters.json) instead of the command line's."""
    workload = workload or args.workload
    steps = args.steps if steps is None else steps
    try:
        with open(os.path.join(ROOT, "tests", "golden", f"bench_{workload}_counters.json")) as f:
            gold = json.load(f)
    except (OSError, ValueError):
        return {"oracle_counters_match": None, "oracle_counters_note": f"no committed oracle counters for workload {workload}"}
    j = gold["job"]
    if world != 1 or (j["width"], j["height"], j["spp"], j["queue_size"], j["triangles"]) != (W, H, spp, N, n_tris):
        return {"oracle_counters_match": None, "oracle_counters_note": f"this job is not the one the committed oracle counters were made for (tests/golden/bench_{workload}_counters.json: {j['width']}x{j['height']}, {j['spp']} spp, queue {j['queue_size']}, one rank)"}
    want = {f: gold["per_render"][f] * steps for f in ORACLE_COUNTER_FIELDS}
    got = m["counter_deltas"]
    ok = all(int(got[f]) == int(want[f]) for f in ORACLE_COUNTER_FIELDS) and m["iters"] == gold["per_render"]["iterations"] * steps
    out = {"oracle_counters_match": bool(ok),
           "oracle_counters": {"source": f"tests/golden/bench_{workload}_counters.json (orc_render on this job; tests/test_gpu_configs.py::test_benchmarked_render_path_matches_oracle_at_full_size[{'framed_16M_8spp' if workload == 'c3_framed' else 'bench_shape_16M_8spp'}] holds the live oracle, the file and the GPU to each other, pixels included)",
                               "per_render": gold["per_render"], "timed_renders": steps}}
    if not ok:
        out["oracle_counters"]["timed_deltas"] = {f: int(got[f]) for f in ORACLE_COUNTER_FIELDS}
        out["oracle_counters"]["timed_iterations"] = m["iters"]
    # ... and the picture: the sums over the accumulation buffer the last timed render left, against the oracle's for this job (the same additions
    # per pixel in another order: 1e-5, the tolerance of the full-size GPU tests)
    want_rgb, got_rgb = gold.get("radiance_sum_rgb"), m.get("radiance_sum_rgb")
    if want_rgb and got_rgb:
        rel = max(abs(g - w) / max(abs(w), 1e-30) for g, w in zip(got_rgb, want_rgb))
        out["oracle_radiance_match"] = bool(rel <= 1e-5)
        out["oracle_radiance"] = {"rel_err": float(f"{rel:.3e}"), "tolerance": 1e-5, "of": "sum over the accumulation buffer's r, g, b after the last timed render (float64 sums of the fp32 pixels) against radiance_sum_rgb of the same file"}
    return out

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
            "fabric_vs_gather_ceiling": round(traffic / GATHER_CEILING_GBS, 4),
            "fabric_vs_gather_ceiling_note": f"counter bytes per second / {GATHER_CEILING_GBS:.0f} GB/s -- the rate at which MI355X_MICROARCH.md 'Indexed rows' measured uniformly random rows gathered out of a table that lives in the Infinity Cache (7.4-7.9 TB/s for 151 MB, 8.6 TB/s for 38 MB; the lower end is used): the roofline that applies to a cache-resident scene (C3: 83 MB), which the 8 TB/s HBM figure is not",
            "valu_issue_frac": round(fr["valu-issue"], 4),
            "salu_issue_frac": round(fr["salu-issue"], 4),
            "lanes_active_per_valu_inst": round(lanes, 4),
            "useful_lane_issue_frac": round(fr["valu-issue"] * lanes, 4),
            "wave_wait_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAVE_CYCLES") else None,
            "traffic_detail": {"hbm_bytes_per_launch": round(hbm_bytes), "FETCH_SIZE_KB": round(c["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(c["WRITE_SIZE"], 1), "correction": "FETCH_SIZE x 2 (gfx950), WRITE_SIZE as is"},
            "pmc_source": pmc["source"],
            "pmc_launches_averaged": pmc["launches_averaged"],
        })
        if bound == "hbm":
            out["bound_note"] = ("the largest of the measured resource fractions, not a wall the launch stands at: profiles/r06_whatif_ray_order_c5.txt -- the same launch with its rays laid out "
                                 "by the triangle they will hit moves 24 % fewer bytes across the fabric and is 8.7 % shorter; with a region of the tree per XCD, 38 % fewer bytes and LONGER.  The fabric's "
                                 "rate is what six latency-bound waves per SIMD ask of it (wave_wait_frac, valu_issue_frac)")
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


def nominal_step_frac(roofline, steps, primary_rays, dt_s):
    """SURVEY.md 8d's nominal bytes of the WHOLE step (every traversal launch + every shade launch + 44 B per primary ray) over the
    timed region's wall time, as a fraction of the HBM peak (C3: 0.96).  Nominal like the traversal kernel's own figure: it counts every
    visit as if it went to HBM and exceeds 1 where the scene is small enough to be served by L2 (C2's 10 k triangles: 1.45)."""
    try:
        trace = roofline["algorithmic"]["bytes_per_launch"] * roofline["launches"]
        shade = roofline["shade"]["algorithmic"]["bytes_per_render"] * steps
    except (KeyError, TypeError):
        return None
    total = trace + shade + 44.0 * primary_rays
    return {"nominal_step_frac": round(total / dt_s / 1e9 / HBM_PEAK_GBS, 4) if dt_s > 0 else None,
            "nominal_step_bytes": {"traversal": round(trace), "shade": round(shade), "primary": round(44.0 * primary_rays), "timed_s": round(dt_s, 6),
                                   "note": "SURVEY.md 8d per-unit bytes x the units of the timed renders / wall time of the timed region / 8 TB/s; nominal: most of these bytes are served by the Infinity Cache and LDS (hbm_counter_frac is what crossed the fabric)"}}
