#!/bin/bash
# tools/primary_pmc.sh -- k_primary (the third kernel of a render by time: 0.28 of C3's 5.26 ms) under rocprofv3 --pmc: what bounds it?
out=${1:-gpurun_out/primary_pmc}; mkdir -p "$out"; root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  d="$root/$out/$(echo $grp | cut -d' ' -f1)"; rm -rf "$d"
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$d" -o pmc -- python3 "$root/tools/stream_probe.py" renders=3 > "$d.log" 2>&1
done
python3 - "$root/$out" <<'PY'
import csv, glob, os, sys
c = {}
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(p)) if "k_primary" in r["Kernel_Name"]]
    for name in {r["Counter_Name"] for r in rows}:
        mine = sorted((r for r in rows if r["Counter_Name"] == name), key=lambda r: int(r["Dispatch_Id"]))
        big = [float(r["Counter_Value"]) for r in mine]
        c[name] = max(big)  # the launch that makes the 16.6 M camera rays (the others are one-block launches)
cyc = c["GRBM_GUI_ACTIVE"] / 8
print({k: f"{v:.4g}" for k, v in c.items()})
print(f"k_primary (16.6 M camera rays): {cyc / 2.4e3:.1f} us at 2.4 GHz; vector issue {4 * c['SQ_ACTIVE_INST_VALU'] / (1024 * cyc):.3f}, scalar issue {4 * c['SQ_ACTIVE_INST_SCA'] / (1024 * cyc):.3f}, lanes per vector instruction {c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']):.3f}, waves waiting {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f}; fabric bytes {(2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / 1e6:.0f} MB")
PY
