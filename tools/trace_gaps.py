#!/usr/bin/env python3
"""tools/trace_gaps.py <kernel_trace.csv> -- idle time between the library's kernels inside the last render of a
`rocprofv3 --kernel-trace` run of bench.py: span, busy, idle, and the gap in front of each kind of kernel (us)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if "tyr::" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"])
prod = [i for i, k in enumerate(ks) if "k_trace_flat<12" in k[2]]  # the traversal launches of the timed renders
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 6
i0 = prod[-n_it]
while "k_primary" not in ks[i0][2] and "k_globals" not in ks[i0][2]:
    i0 -= 1
seq = ks[i0:]
busy = sum(e - s for s, e, _ in seq)
span = seq[-1][1] - seq[0][0]
print(f"last render: {len(seq)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
gaps, prev = {}, None
for s, e, n in seq:
    short = n.replace("void ", "").replace("tyr::", "").split("(")[0][:28]
    if prev:
        gaps.setdefault(prev[1] + " -> " + short, []).append((s - prev[0]) / 1e3)
    prev = (e, short)
for k, v in gaps.items():
    print(f"  {k:64s} {' '.join(f'{x:6.1f}' for x in v)}")
