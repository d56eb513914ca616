#!/usr/bin/env python3
"""tools/isa_blocks.py <kernels.s> <mangled-substring> -- per-basic-block instruction mix of one kernel
(VALU / SALU / memory ops / branches) from `make -C tyrant_amd/csrc asm`; used to size the traversal loop."""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
s = open(path).read().split("\n")
i0 = [i for i, l in enumerate(s) if re.match(r"^_Z\w*:", l) and key in l][0]
end = next(k for k in range(i0 + 10, len(s)) if s[k].strip().startswith(".section"))
body = s[i0:end]
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(body))
blocks, cur = [], ["entry", []]
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), []]
    elif l.startswith("\t") and not l.strip().startswith(".") and not l.strip().startswith(";"):
        cur[1].append(l.strip())
blocks.append(cur)
tv = ts = 0
for name, ins in blocks:
    v = sum(1 for i in ins if i.startswith("v_"))
    sc = sum(1 for i in ins if i.startswith("s_"))
    tv += v
    ts += sc
    mem = [i.split()[0] for i in ins if re.match(r"(global|scratch|ds|buffer|flat)_", i)]
    br = [i.split()[0][2:] + "->" + i.split()[-1] for i in ins if "branch" in i]
    print(f"{name:12s} n={len(ins):4d} valu={v:4d} salu={sc:4d} mem={mem} br={br}")
print(f"total valu={tv} salu={ts}")
