#!/usr/bin/env python3
"""tools/isa_lines.py <unit.s built with -gline-tables-only> <mangled-kernel-substring> [blocks|lines] -- where a kernel's
instructions come from: per basic block (size, vector / scalar / memory instructions, the source lines that contribute most)
or per source line (vector instructions).  Build the listing with
  hipcc <the Makefile's HIPFLAGS> -gline-tables-only -S --cuda-device-only hip/<unit>.hip -o <unit>_g.s"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
mode = sys.argv[3] if len(sys.argv) > 3 else "blocks"
s = open(path).read().split("\n")
files = {}
for l in s:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
i0 = [i for i, l in enumerate(s) if re.match(r"^_Z\w*:", l) and key in l][0]
end = next(k for k in range(i0 + 10, len(s)) if s[k].strip().startswith(".section"))
blk, cur = "entry", ("?", 0)
blocks = collections.OrderedDict()
lines = collections.Counter()
for l in s[i0:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blk = m.group(1)
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    b = blocks.setdefault(blk, {"n": 0, "v": 0, "s": 0, "lines": collections.Counter(), "mem": 0})
    b["n"] += 1
    if t.startswith("v_"):
        b["v"] += 1
        lines[cur] += 1
    if t.startswith("s_"):
        b["s"] += 1
    if re.match(r"(global|flat|ds|scratch|buffer)_", t):
        b["mem"] += 1
    b["lines"][cur] += 1
if mode == "lines":
    print("vector instructions", sum(lines.values()))
    for (f, ln), c in lines.most_common(60):
        print(f"{f}:{ln}  {c}")
else:
    for k, b in blocks.items():
        if b["n"] >= 20:
            top = ", ".join(f"{f}:{ln} x{c}" for (f, ln), c in b["lines"].most_common(4))
            print(f"{k:10s} n={b['n']:4d} valu={b['v']:4d} salu={b['s']:4d} mem={b['mem']:3d} | {top}")
