#!/usr/bin/env python3
"""tools/make_traffic_json.py <workload> <summary.txt> -- turn tools/pmc_traffic.sh's summary into
profiles/traffic_<workload>.json (read by bench.py for roofline.traffic): HBM bytes per launch of the production
extend kernel, corrected as MI355X_MICROARCH.md section HBM prescribes (FETCH_SIZE x2 on gfx950, WRITE_SIZE as is;
both counters are in KB)."""
import json
import os
import re
import sys

wl, path = sys.argv[1], sys.argv[2]
queue = int(sys.argv[3]) if len(sys.argv) > 3 else 1920 * 1080 * 8  # bench.py's default queue at 1080p, 8 spp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kernel, vals = None, {}
cur = None
for line in open(path):
    m = re.match(r"== (.*)", line)
    if m:
        cur = m.group(1).strip()
        continue
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+/dispatch\s+([0-9.]+)\s+\((\d+) dispatches\)", line)
    if m and cur and cur.startswith("k_trace_flat<12"):
        kernel = cur
        vals[m.group(1)] = (float(m.group(3)), int(m.group(4)))
assert kernel and len(vals) == 2, (kernel, vals)
fetch, n = vals["FETCH_SIZE"]
write, _ = vals["WRITE_SIZE"]
name = f"r01_pmc_traffic_bench_{wl}_final.txt"
out = {
    "workload": wl,
    "queue_size": queue,
    "kernel": kernel,
    "launches": n,
    "FETCH_SIZE_KB_per_launch": fetch,
    "WRITE_SIZE_KB_per_launch": write,
    "hbm_bytes_per_launch_corrected": (2.0 * fetch + write) * 1024.0,
    "correction": "FETCH_SIZE doubled (gfx950 counts 128-B fabric reads as 64 B for wide coalesced streams; the node gathers are an uncalibrated access width), WRITE_SIZE as is -- MI355X_MICROARCH.md section HBM",
    "command": f"tools/pmc_traffic.sh --workload {wl}  (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, over `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-queue`)",
    "source": "profiles/" + name,
}
with open(os.path.join(ROOT, "gpurun_out", f"traffic_{wl}.json"), "w") as f:
    json.dump(out, f, indent=1)
with open(os.path.join(ROOT, "gpurun_out", name), "w") as f:
    f.write(open(path).read())
print(json.dumps(out))
