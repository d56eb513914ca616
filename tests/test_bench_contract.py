"""bench.py's contract, checked on CODE: the pure parts (job shapes, the roofline block, the launcher's command line)
on the CPU, and the emitted JSON line by actually running bench.py on the GPU (a tiny c1 job) -- a regression in
bench.py fails here, not only in a committed log."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402  (no torch / GPU at import time)


def test_job_shapes_follow_baseline_configs():
    a = bench.parse_args([])
    assert a.workload == "c3" and a.gpus == 1  # the default line is the configuration north_star's target names
    assert bench.job_shape(a, 1) == (8, 1920 * 1080 * 8)  # C3: 1080p, 8 spp, every primary ray in flight
    # C4: 64 spp in total whatever N (strong scaling, the default for N > 1); queue = spp x local pixels, capped at 32 Mi
    assert bench.job_shape(a, 2) == (64, 1 << 25)
    assert bench.job_shape(a, 4) == (64, 1920 * 270 * 64)
    assert bench.job_shape(a, 8) == (64, 1920 * 1080 * 8)
    w = bench.parse_args(["--scaling", "weak"])
    assert bench.job_shape(w, 4) == (32, 1920 * 270 * 32)
    q = bench.parse_args(["--queue", "2097152", "--spp", "3"])
    assert bench.job_shape(q, 1) == (3, 2097152) and bench.job_shape(q, 2) == (3, 2097152)


def _fake_pmc(fetch_kb, write_kb, valu, sca, cycles):
    c = {"FETCH_SIZE": fetch_kb, "WRITE_SIZE": write_kb, "SQ_ACTIVE_INST_VALU": valu, "SQ_ACTIVE_INST_SCA": sca, "SQ_THREAD_CYCLES_VALU": valu * 64 * 0.5, "SQ_INSTS_VALU": valu,
         "SQ_INSTS_SALU": sca, "SQ_WAVE_CYCLES": 1e9, "SQ_WAIT_ANY": 4e8, "GRBM_GUI_ACTIVE": cycles * bench.NUM_XCD}
    return {"counters": c, "launches_averaged": 6, "source": "unit test"}


def test_roofline_block_names_the_tightest_measured_resource_and_never_exceeds_one():
    visits = {"nodes_per_ext": 42.8, "tris_per_ext": 1.37, "nodes_per_con": 70.0, "tris_per_con": 2.0}
    # a cache-resident tree (round 1's C2): 0.23 GB of fabric traffic per 0.7 ms launch, issue pipes 60-66 % busy
    cyc = 1.6e6
    pmc = _fake_pmc(fetch_kb=100e3, write_kb=30e3, valu=0.60 * bench.NUM_SIMD * cyc / 4, sca=0.66 * bench.NUM_SIMD * cyc / 4, cycles=cyc)
    r = bench.roofline_block(pmc, ext_ms=4.2, ext_launches=6, ext_rays=32.5e6, visits=visits, kernel_ms_per_render={})
    assert r["bound"] == "salu-issue" and abs(r["frac"] - 0.66) < 1e-3
    assert 0 < r["hbm_counter_frac"] < 0.1 and abs(r["valu_issue_frac"] - 0.60) < 1e-3 and abs(r["lanes_active_per_valu_inst"] - 0.5) < 1e-6
    assert r["algorithmic"]["frac_of_hbm_peak"] > 1.0  # the nominal figure may exceed the peak -- which is why it is not `frac`
    assert 0 < r["frac"] <= 1 and r["frac"] == round(r["achieved"] / r["peak"], 4)
    assert abs(r["traffic"] - (2 * 100e3 + 30e3) * 1024 / 0.7e-3 / 1e9) < 1.0
    # an HBM-bound variant of the same launch: the memory side becomes the bound and the unit follows
    pmc = _fake_pmc(fetch_kb=2.0e6, write_kb=0.2e6, valu=0.3 * bench.NUM_SIMD * cyc / 4, sca=0.2 * bench.NUM_SIMD * cyc / 4, cycles=cyc)
    r = bench.roofline_block(pmc, ext_ms=4.2, ext_launches=6, ext_rays=32.5e6, visits=visits, kernel_ms_per_render={})
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] <= 1
    # no counters at all: the nominal figure, clamped, and said so
    r = bench.roofline_block(None, ext_ms=4.2, ext_launches=6, ext_rays=32.5e6, visits=visits, kernel_ms_per_render={})
    assert r["bound"] == "hbm" and r["frac"] <= 1.0 and r["traffic"] is None and "note" in r


def test_roofline_block_of_the_merged_trace_kernel():
    """merged launches: the algorithmic figure covers every extend AND every shadow ray of the timed renders over the
    trace launches plus the last iteration's connect launch; the bound still comes from the counters alone"""
    visits = {"nodes_per_ext": 12.3, "tris_per_ext": 0.9, "nodes_per_con": 14.9, "tris_per_con": 0.7, "visible_frac": 0.6}
    pmc = {"counters": {"FETCH_SIZE": 1.2e6, "WRITE_SIZE": 1.0e5, "GRBM_GUI_ACTIVE": 8 * 1.8e6, "SQ_ACTIVE_INST_VALU": 2.9e8, "SQ_ACTIVE_INST_SCA": 5.0e7,
                        "SQ_THREAD_CYCLES_VALU": 64 * 2.9e8 * 0.39, "SQ_INSTS_VALU": 2.8e8, "SQ_INSTS_SALU": 2.0e8, "SQ_WAVE_CYCLES": 9e9, "SQ_WAIT_ANY": 6e9},
           "launches_averaged": 6, "kernel": bench.TRACE_KERNEL, "source": "unit test"}
    ext_rays, shadow_rays = 31.0e6, 11.5e6
    r = bench.roofline_block(pmc, ext_ms=4.4, ext_launches=6, ext_rays=ext_rays, visits=visits, kernel_ms_per_render={}, kernel=bench.TRACE_KERNEL, con_ms=0.26, shadow_rays=shadow_rays)
    assert bench.TRACE_KERNEL in r["kernel"] and "connect" in r["kernel"]
    assert r["bound"] == "valu-issue" and 0 < r["frac"] <= 1 and r["frac"] == r["valu_issue_frac"]
    assert abs(r["lanes_active_per_valu_inst"] - 0.39) < 1e-3
    per_ext = 24 + 8 + 32 * 12.3 + 36 * 0.9
    per_con = 44 + 32 * 14.9 + 36 * 0.7 + 12 * 0.6
    want = (per_ext * ext_rays + per_con * shadow_rays) / ((4.4 + 0.26) * 1e-3) / 1e9
    assert abs(r["algorithmic"]["GBps"] - want) < 0.02 * want
    assert r["algorithmic"]["bytes_per_launch"] == round((per_ext * ext_rays + per_con * shadow_rays) / 6)
    assert "shadow" in r["algorithmic"]["covers"]
    # round 4: the nominal count charged for the rays the kernel is handed only, the quad layout's own bytes, and north_star's target answered both ways
    v2 = dict(visits, in_tree_ext_frac=0.4)
    quad = {"quad_steps_per_render": 2.0e8, "triangle_tests_per_render": 3.0e7, "source": "unit test"}
    r3 = bench.roofline_block(pmc, ext_ms=4.4, ext_launches=6, ext_rays=ext_rays, visits=v2, kernel_ms_per_render={}, kernel=bench.TRACE_KERNEL, con_ms=0.26, shadow_rays=shadow_rays, quad=quad, renders=1)
    c0 = r3["algorithmic"]["class0_only"]
    rays_in = 0.4 * ext_rays
    want0 = 32 * rays_in + 32 * (12.3 * ext_rays - (ext_rays - rays_in)) + 36 * 0.9 * ext_rays + per_con * shadow_rays
    assert abs(c0["bytes_per_launch"] - want0 / 6) < 1e-6 * want0 and c0["frac_of_hbm_peak"] < r3["algorithmic"]["frac_of_hbm_peak"]
    q = r3["algorithmic"]["quad"]
    assert abs(q["bytes_per_launch"] - (128 * 2.0e8 + 48 * 3.0e7 + 32 * (rays_in + shadow_rays)) / 6) < 1e3
    t = r3["hbm_target_met"]
    assert t["nominal"] == (r3["algorithmic"]["frac_of_hbm_peak"] >= 0.5) and t["counters"] == (r3["hbm_counter_frac"] >= 0.5) and t["nominal_class0_only"] == (c0["frac_of_hbm_peak"] >= 0.5)
    # the separate-launch form of the same numbers prices the extend launches only
    r2 = bench.roofline_block(pmc, ext_ms=4.4, ext_launches=6, ext_rays=ext_rays, visits=visits, kernel_ms_per_render={})
    assert bench.EXTEND_KERNEL in r2["kernel"] and r2["algorithmic"]["GBps"] < r["algorithmic"]["GBps"] * 1.2
    assert abs(r2["algorithmic"]["GBps"] - per_ext * ext_rays / 4.4e-3 / 1e9) < 0.02 * r2["algorithmic"]["GBps"]


def test_committed_pmc_profiles_parse():
    """profiles/pmc_<workload>.json is what bench.py falls back to when rocprofv3 cannot run beside it"""
    for wl, n in (("c3", 1920 * 1080 * 8),):
        path = os.path.join(ROOT, "profiles", f"pmc_{wl}.json")
        if not os.path.exists(path):
            pytest.skip(f"{path} not committed yet")
        pmc = bench.committed_pmc(wl, n)
        assert pmc is not None and all(k in pmc["counters"] for g in bench.PMC_PASSES for k in g)


def test_oracle_counters_check_compares_the_timed_deltas_with_the_committed_oracle_counters():
    """config.oracle_counters_match: K timed renders of the committed job must have cast exactly K times the oracle's rays
    (tests/golden/bench_c3_counters.json, made by tests/golden/make_bench_counters.py); another job -> None, with the reason"""
    with open(os.path.join(ROOT, "tests", "golden", "bench_c3_counters.json")) as f:
        gold = json.load(f)
    j, per = gold["job"], gold["per_render"]
    assert (j["workload"], j["width"], j["height"], j["spp"], j["queue_size"], j["nranks"]) == ("c3", 1920, 1080, 8, 1920 * 1080 * 8, 1)
    assert per["total_primary_rays"] == j["spp"] * j["width"] * j["height"] and per["n_survive"] == per["total_extend_rays"] - per["total_primary_rays"]
    a = bench.parse_args(["--steps", "20"])
    m = {"counter_deltas": {f: per[f] * 20 for f in bench.ORACLE_COUNTER_FIELDS}, "iters": per["iterations"] * 20}
    ok = bench.oracle_counters_check(a, 1, 1920, 1080, 8, 1920 * 1080 * 8, j["triangles"], m)
    assert ok["oracle_counters_match"] is True and ok["oracle_counters"]["timed_renders"] == 20
    m["counter_deltas"]["total_shadow_rays"] += 1  # one shadow ray too many in twenty renders
    bad = bench.oracle_counters_check(a, 1, 1920, 1080, 8, 1920 * 1080 * 8, j["triangles"], m)
    assert bad["oracle_counters_match"] is False and bad["oracle_counters"]["timed_deltas"]["total_shadow_rays"] == per["total_shadow_rays"] * 20 + 1
    # the picture too: the accumulation buffer's sums after the last timed render against the oracle's (same file), 1e-5
    assert "oracle_radiance_match" not in ok
    m["counter_deltas"]["total_shadow_rays"] -= 1
    m["radiance_sum_rgb"] = [x * (1.0 + 2e-7) for x in gold["radiance_sum_rgb"]]
    ok = bench.oracle_counters_check(a, 1, 1920, 1080, 8, 1920 * 1080 * 8, j["triangles"], m)
    assert ok["oracle_counters_match"] is True and ok["oracle_radiance_match"] is True and ok["oracle_radiance"]["rel_err"] < 1e-6
    m["radiance_sum_rgb"][1] *= 1.0 + 1e-4
    assert bench.oracle_counters_check(a, 1, 1920, 1080, 8, 1920 * 1080 * 8, j["triangles"], m)["oracle_radiance_match"] is False
    for other in (dict(world=2), dict(N=2097152), dict(spp=4), dict(W=1280)):
        kw = dict(world=1, W=1920, H=1080, spp=8, N=1920 * 1080 * 8)
        kw.update(other)
        r = bench.oracle_counters_check(a, kw["world"], kw["W"], kw["H"], kw["spp"], kw["N"], j["triangles"], m)
        assert r["oracle_counters_match"] is None and "oracle_counters_note" in r
    assert bench.oracle_counters_check(bench.parse_args(["--workload", "c1"]), 1, 1920, 1080, 8, 1920 * 1080 * 8, 36, m)["oracle_counters_match"] is None  # (no committed counters for c1)
    for wl in ("c2", "c5"):  # round 6: the other two bench workloads have their oracle counters too
        with open(os.path.join(ROOT, "tests", "golden", f"bench_{wl}_counters.json")) as f:
            g2 = json.load(f)
        j2, p2 = g2["job"], g2["per_render"]
        assert j2["workload"] == wl and p2["total_primary_rays"] == j2["spp"] * j2["width"] * j2["height"] and p2["n_survive"] == p2["total_extend_rays"] - p2["total_primary_rays"]
        m2 = {"counter_deltas": {f: p2[f] * 2 for f in bench.ORACLE_COUNTER_FIELDS}, "iters": p2["iterations"] * 2}
        a2 = bench.parse_args(["--workload", wl, "--steps", "2"])
        assert bench.oracle_counters_check(a2, 1, j2["width"], j2["height"], j2["spp"], j2["queue_size"], j2["triangles"], m2)["oracle_counters_match"] is True
    a5 = bench.parse_args(["--workload", "c5", "--width", "3840", "--height", "2160", "--spp", "16"])
    assert bench.job_shape(a5, 1) == (16, 1 << 25)  # the queue size the committed C5 counters were made for
    # the secondary workload (config.framed) is checked against its own file, with its own number of timed renders
    with open(os.path.join(ROOT, "tests", "golden", "bench_c3_framed_counters.json")) as f:
        fr = json.load(f)
    assert fr["job"]["workload"] == "c3_framed" and fr["job"]["triangles"] == j["triangles"] and fr["bvh_nodes_sha256"] == gold["bvh_nodes_sha256"]  # the same scene and tree, another camera
    assert fr["per_render"]["total_extend_rays"] > 2 * per["total_extend_rays"]  # ... from which the rays actually enter the room
    mf = {"counter_deltas": {f: fr["per_render"][f] * 10 for f in bench.ORACLE_COUNTER_FIELDS}, "iters": fr["per_render"]["iterations"] * 10}
    ok = bench.oracle_counters_check(a, 1, 1920, 1080, 8, 1920 * 1080 * 8, j["triangles"], mf, workload="c3_framed", steps=10)
    assert ok["oracle_counters_match"] is True and ok["oracle_counters"]["timed_renders"] == 10 and "framed_16M_8spp" in ok["oracle_counters"]["source"]
    assert bench.oracle_counters_check(a, 1, 1920, 1080, 8, 1920 * 1080 * 8, j["triangles"], m, workload="c3_framed", steps=10)["oracle_counters_match"] is False  # (C3's own counters are not the framed job's)


def test_nominal_step_fraction_and_gather_ceiling():
    """roofline.nominal_step_frac = SURVEY.md 8d's bytes of the whole step / wall time / 8 TB/s; roofline.fabric_vs_gather_ceiling = the
    counters' bytes per second / the guide's Infinity-Cache random-row rate (round 5's review recomputed both by hand: 0.96 and ~0.5)"""
    roof = {"algorithmic": {"bytes_per_launch": 5.064e9}, "launches": 140, "shade": {"algorithmic": {"bytes_per_render": 4.205e9}}}
    nf = bench.nominal_step_frac(roof, 20, 20 * 16588800, 0.10538)
    assert abs(nf["nominal_step_frac"] - 0.96) < 0.01 and nf["nominal_step_bytes"]["primary"] == round(44.0 * 20 * 16588800)
    assert bench.nominal_step_frac({}, 20, 1, 1.0) is None
    pmc = {"counters": {"FETCH_SIZE": 900000.0, "WRITE_SIZE": 202000.0, "GRBM_GUI_ACTIVE": 8 * 1.2e6, "SQ_ACTIVE_INST_VALU": 2.0e8, "SQ_ACTIVE_INST_SCA": 1.4e8, "SQ_THREAD_CYCLES_VALU": 64 * 2.0e8 * 0.68,
                        "SQ_WAVE_CYCLES": 1.0e9, "SQ_WAIT_ANY": 5.0e8}, "source": "test", "launches_averaged": 7}
    visits = {"nodes_per_ext": 21.56, "tris_per_ext": 1.66, "nodes_per_con": 28.06, "tris_per_con": 1.957, "visible_frac": 0.678}
    r = bench.roofline_block(pmc, ext_ms=3.7, ext_launches=7, ext_rays=32.9e6, visits=visits, kernel_ms_per_render={}, kernel=bench.TRACE_KERNEL, shadow_rays=9.5e6)
    assert abs(r["fabric_vs_gather_ceiling"] - r["traffic"] / 7400.0) < 1e-3 and r["fabric_vs_gather_ceiling"] > r["hbm_counter_frac"]
    assert r["bound"] == "valu-issue" and "bound_note" not in r
    # a launch whose largest measured fraction is the fabric's (C5) says what that fraction is and is not
    hot = dict(pmc, counters=dict(pmc["counters"], FETCH_SIZE=8.7e6, WRITE_SIZE=3.2e5))
    r5 = bench.roofline_block(hot, ext_ms=3.7 * 5, ext_launches=7, ext_rays=32.9e6, visits=visits, kernel_ms_per_render={}, kernel=bench.TRACE_KERNEL, shadow_rays=9.5e6)
    assert r5["bound"] == "hbm" and r5["unit"] == "GB/s" and "r06_whatif_ray_order_c5" in r5["bound_note"]


def _bench_line(stdout: str) -> dict:
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_emits_the_contract_line():
    """python bench.py on a tiny c1 job: ONE JSON line with the contract's keys, a roofline whose frac is in (0, 1] and a
    CPU baseline; --pmc auto exercises the rocprofv3 child passes when the profiler is installed"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c1", "--width", "320", "--height", "180", "--spp", "2", "--steps", "2", "--warmup", "1", "--no-reference-queue", "--cpu-iterations", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = _bench_line(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["in_tree_Mrays/s"] <= d["value"]
    assert list(d["config"])[:3] == ["workload", "in_tree_Mrays/s", "in_tree_fraction"]  # what `value` is made of comes first
    sp = d["ms_per_step_spread"]
    assert sp["min"] <= sp["p50"] <= sp["p95"] <= sp["max"] and d["config"]["spread"]["renders"] == 200 and "framed" not in d["config"]  # (the framed view belongs to the default C3 job)
    assert d["roofline"]["nominal_step_frac"] > 0
    assert d["config"]["device_bvh_build_s"]["same_bytes_as_host_build"] is True and d["config"]["device_bvh_build_s"]["device"] >= 0  # the tree built again on the GPU: the host builder's bytes
    assert d["config"]["device_bvh_build_s"]["build_upload"]["same_scene_in_hbm_as_build_then_upload"] is True  # ... and built + laid out there in one call: the same scene in HBM
    assert d["config"]["host_scene_upload_s"]["layout_on"] == "device"
    assert d["config"]["oracle_counters_match"] is None  # (only the default job has committed oracle counters; this micro-job says so)
    ss = d["config"]["steady_state"]  # the same kernels with the queue kept full, beside the metric (never instead of it)
    assert ss["Mrays/s"] > 0 and ss["iterations"] == 12 and ss["queue_size"] == d["config"]["queue_size"]
    r = d["roofline"]
    assert 0 < r["frac"] <= 1 and r["bound"] in ("hbm", "valu-issue", "salu-issue") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    if bench.find_rocprof():
        assert r.get("pmc_source", "").startswith("live"), r  # the counters were collected beside this very run
        assert r["traffic"] is not None and 0 < r["hbm_counter_frac"] <= 1 and 0 < r["valu_issue_frac"] <= 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and c["trace_same_ray_set"]["port"]["value"] > 0
    # timings of this micro-job sit at the resolution of the clock and of the line's rounding: presence and sign only
    # (round 3: `bvh_build_s.port > 0` on a 0.2 ms build rounded to 3 places was a coin flip that hid every parity test)
    assert c["whole_path_first_iterations"]["value"] >= 0 and c["bvh_build_s"]["port"] >= 0 and "port_us" in c["bvh_build_s"]


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher in the environment: the parent starts the two ranks itself (gloo
    rehearsal: both on device 0, rows dealt y % 2 == rank), strong scaling = the same 4 spp in total, and rank 0's line
    carries the one-GPU time of the same job.  bench.py itself asserts that every pixel of the combined frame holds
    exactly spp_total completed paths."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo", "--workload", "c1", "--width", "320", "--height", "180",
           "--queue", "32768", "--spp", "4", "--no-reference-queue"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = _bench_line(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["spp_total"] == 4 and d["value"] > 0
    s = d["config"]["strong_scaling"]
    assert s["one_gpu_ms"] > 0 and s["speedup_vs_1gpu"] > 0 and abs(s["efficiency_vs_1gpu"] - s["speedup_vs_1gpu"] / 2) < 1e-3


def _free_port():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


@pytest.mark.gpu
def test_native_exchange_preflight():
    """bench.py checks the native exchange (tyr_dist_*) in a child of every rank before the rank touches its GPU, so that a
    hang, a crash or a wrong frame costs the child and not the measurement.  One rank on one GPU: verified (exit 0).  Two
    ranks on the one GPU of this box: RCCL refuses the second rank on a device -- the failure path: both children exit 1,
    within their bounds."""
    import tempfile

    script = os.path.join(ROOT, "bench.py")
    store = os.path.join(tempfile.mkdtemp(prefix="tyr_pf_"), "store1")  # the children rendezvous through a file (no port to agree on)
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, script, "--dist-preflight", "--preflight-store", store], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    store = os.path.join(tempfile.mkdtemp(prefix="tyr_pf_"), "store2")
    procs = [subprocess.Popen([sys.executable, script, "--dist-preflight", "--preflight-store", store], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT,
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", TYR_BENCH_PREFLIGHT_ONE_DEVICE="1")) for r in range(2)]
    outs = [q.communicate(timeout=300) for q in procs]
    assert [q.returncode for q in procs] == [1, 1], [o[1][-1500:] for o in outs]


@pytest.mark.gpu
def test_bench_ranks_run_the_preflight_and_fall_back():
    """the whole path under the driver's launcher form: every rank spawns its pre-flight child first (forced here although the
    backend is gloo: TYR_BENCH_PREFLIGHT_ONE_DEVICE), the children fail as two ranks on one device must, and the bench
    still delivers its line over torch.distributed"""
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TYR_BENCH_PREFLIGHT_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo", "--workload", "c1", "--width", "320", "--height", "180", "--queue", "32768", "--spp", "4", "--no-reference-queue"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = _bench_line(p.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and "torch.distributed" in d["config"]["sharding"]


@pytest.mark.gpu
def test_bench_nccl_on_a_box_where_rccl_cannot_work_falls_back_and_says_why():
    """`bench.py --gpus 2 --backend nccl` on the one GPU of a test box: RCCL refuses two ranks on one device -- the
    library's exchange and torch's nccl backend alike -- which is what an RCCL that cannot initialise looks like to the
    driver's 8-GPU run.  The children's pre-flight finds out, the ranks combine over gloo (host-staged), and the line is
    well-formed and says why (config.combine.fallback_reason, config.backend)."""
    import torch

    if torch.cuda.device_count() > 1:
        pytest.skip("more than one GPU: RCCL would work")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "nccl", "--workload", "c1", "--width", "320", "--height", "180", "--queue", "32768", "--spp", "4", "--no-reference-queue"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    d = _bench_line(p.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["spp_total"] == 4
    c = d["config"]["combine"]
    assert c["native_combine"] is False and "gloo" in d["config"]["backend"] and "nccl" in d["config"]["backend"]
    assert c["fallback_reason"] and "pre-flight" in c["fallback_reason"] and "gloo" in c["fallback_reason"]


def test_bench_refuses_to_run_without_a_gpu():
    """the product path has no CPU fallback: on a box without a GPU bench.py says so instead of measuring something else"""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c1", "--width", "64", "--height", "64", "--steps", "1", "--warmup", "0", "--pmc", "off"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "MI355X" in (p.stderr + p.stdout)
