#!/usr/bin/env python3
"""tools/render_timeline.py <kernel_trace.csv> [renders_from_end=1] -- start / end (us, relative to the render's first
kernel) of every kernel of the last render in a `rocprofv3 --kernel-trace` run of bench.py, one line per launch: what
overlaps what when a render uses two streams (the early shade launch beside the traversal launch)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), int(r.get("Grid_Size_X", 0) or 0)) for r in rows if "tyr::" in r["Kernel_Name"])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prim = [i for i, k in enumerate(ks) if "k_primary" in k[2] and k[4] > 256]  # the launch that generates rays: one per render of bench.py (the queue holds every primary ray); later iterations launch one block (set_wavefront_globals)
i0 = prim[-back]
i1 = prim[-back + 1] if back > 1 else len(ks)
# (bench.py goes on after its last render -- the counting render, the tree built and laid out on the device: the render ends at the
# first kernel that is not one of tyr_render's)
RENDER = ("k_primary", "k_pad_holes", "k_trace_flat", "k_shade", "k_scan_words", "k_extend_spheres", "k_connect_spheres")
for i in range(i0, i1):
    if not any(r in ks[i][2] for r in RENDER):
        i1 = i
        break
t0 = ks[i0][0]
for s, e, n, q, _ in ks[i0:i1]:
    short = n.replace("void ", "").replace("tyr::", "").split("(")[0][:34]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f} us  q{q:>3}  {short}")
print(f"span {(max(k[1] for k in ks[i0:i1]) - t0) / 1e6:.3f} ms")
