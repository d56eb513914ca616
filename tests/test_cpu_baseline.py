"""bench.py's CPU baseline leg (oracle/cpu_baseline.py) on its own, on the CPU: the same function the bench line's
`cpu_baseline` block comes from, on a workload small enough for the CPU suite -- so that a c1 micro-run of bench.py on
the GPU box is not the only thing that exercises it."""
from conftest import built_scene


def test_cpu_baseline_block_is_well_formed_and_self_consistent(orc):
    from oracle.cpu_baseline import cpu_baseline

    sc, _, _ = built_scene("cornell_soup2k")
    W, H, spp = 96, 64, 2
    c = cpu_baseline(sc, W, H, W * H * spp, 2, sc.triangle_materials, spp)
    assert c["unit"] == "Mrays/s" and c["cores"] == 1 and c["kind"] in ("port", "reference") and c["value"] > 0
    t = c["trace_same_ray_set"]
    assert t["port"]["value"] > 0 and t["port"]["seconds"] > 0 and t["port"]["hits"] > 0
    if "reference" in t:  # oracle/_ref is built: the reference's own traversal on the same rays, and its own builder on the same triangles
        assert c["kind"] == "reference" and c["value"] == t["reference"]["value"]
        assert t["reference"]["hits"] == t["port"]["hits"] and t["reference"]["distances_bit_identical_to_port"] is True
        assert c["bvh_build_s"]["reference"] > 0 and c["bvh_build_s"]["reference_nodes_identical_to_port"] is True
    w = c["whole_path_first_iterations"]
    assert w["value"] > 0 and w["kind"] == "port" and "first 2 wavefront iterations" in w["sample"]
    b = c["bvh_build_s"]
    assert b["port"] > 0 and abs(b["port_us"] - b["port"] * 1e6) < 1.0  # full precision: a small tree builds in well under a millisecond
    assert c["host_cpus"] >= 1
