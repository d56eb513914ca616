#!/bin/bash
# tools/coherence_whatif_pmc.sh -- the bytes behind tools/coherence_whatif.py: one rocprofv3 --pmc FETCH_SIZE pass over the same script (one
# timed extend per order), the traversal kernel's last eight dispatches = as_is, shuffled, morton, morton_xcd, hit, hit_xcd, morton_dir, as_is
set -e
out=${1:-gpurun_out/coh}
mkdir -p "$out"
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE TCC_HIT_sum,TCC_MISS_sum; do
    d="$root/$out/pmc_${ctr%%,*}"
    rm -rf "$d"
    timeout -k 10 600 rocprofv3 --pmc ${ctr//,/ } --output-format csv -d "$d" -o pmc -- python3 "$root/tools/coherence_whatif.py" 2 1 > "$root/$out/pmc_${ctr%%,*}.log" 2>&1
    python3 - "$d" <<'PY'
import csv, glob, os, sys
rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(p)))
names = ("as_is", "shuffled", "morton", "morton_xcd", "hit", "hit_xcd", "morton_dir", "as_is")
for ctr in sorted({r["Counter_Name"] for r in rows}):
    mine = sorted((r for r in rows if "k_trace_flat<12, 768" in r["Kernel_Name"] and r["Counter_Name"] == ctr), key=lambda r: int(r["Dispatch_Id"]))[-8:]
    print(ctr, " ".join(f"{n}={float(r['Counter_Value']):.4g}" for n, r in zip(names, mine)))
PY
done
