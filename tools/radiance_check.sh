#!/bin/bash
# tools/radiance_check.sh -- the bench line's oracle_counters_match AND oracle_radiance_match on every workload that has committed oracle figures
out=${1:-gpurun_out/rad}; mkdir -p "$out"
q="--no-reference-queue --no-cpu-baseline --no-steady-state --no-spread --pmc off"
timeout -k 10 300 python bench.py --steps 5 --warmup 1 $q > "$out/c3.json" 2> "$out/c3.err"
timeout -k 10 300 python bench.py --workload c2 --steps 5 --warmup 1 $q > "$out/c2.json" 2> "$out/c2.err"
timeout -k 10 400 python bench.py --workload c5 --width 3840 --height 2160 --spp 16 --steps 2 --warmup 1 $q > "$out/c5.json" 2> "$out/c5.err"
python3 - "$out" <<'PY'
import json, sys, os
for w in ("c3", "c2", "c5"):
    for l in open(os.path.join(sys.argv[1], w + ".json")):
        if l.startswith("{"):
            c = json.loads(l)["config"]
            print(w, "counters", c.get("oracle_counters_match"), "radiance", c.get("oracle_radiance_match"), (c.get("oracle_radiance") or {}).get("rel_err"))
            f = c.get("framed")
            if f:
                print("  framed counters", f.get("oracle_counters_match"), "radiance", f.get("oracle_radiance_match"), (f.get("oracle_radiance") or {}).get("rel_err"))
PY
