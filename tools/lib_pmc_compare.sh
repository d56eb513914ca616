#!/bin/bash
# tools/lib_pmc_compare.sh <tag> [...] -- the traversal kernel's counters, product library against tagged builds (TYRANT_HIP_LIBRARY), over
# tools/stream_probe.py (three C3 renders): one rocprofv3 --pmc pass per counter group and library; sums over the k_trace_flat<12, 768u> launches
out=gpurun_out/libpmc; mkdir -p $out; root=$(pwd)
cd /tmp && export TMPDIR=/tmp
groups=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT")
for tag in product "$@"; do
  lib=$root/tyrant_amd/lib/libtyrant_hip.so; [ "$tag" != product ] && lib=$root/tyrant_amd/lib/libtyrant_hip_$tag.so
  i=0
  for g in "${groups[@]}"; do
    d=$root/$out/$tag/g$i; rm -rf $d; mkdir -p $d
    TYRANT_HIP_LIBRARY=$lib timeout -k 10 200 rocprofv3 --pmc $g --output-format csv -d $d -o pmc -- python3 $root/tools/stream_probe.py renders=3 ${PROBE_KNOBS} > $d/log.txt 2>&1 || echo "$tag group $i ($g) failed"
    i=$((i+1))
  done
done
python3 - $root/$out "$@" <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
tags = ["product"] + sys.argv[2:]
tab = collections.OrderedDict()
for t in tags:
    for p in glob.glob(os.path.join(root, t, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if "k_trace_flat<12, 768" in r["Kernel_Name"]:
                tab.setdefault(r["Counter_Name"], collections.defaultdict(float))[t] += float(r["Counter_Value"])
print(f"{'counter':26s}" + "".join(f"{t:>16s}" for t in tags) + "   ratio")
for name, v in tab.items():
    base = v.get("product", 0.0)
    print(f"{name:26s}" + "".join(f"{v.get(t, float('nan')):16.5g}" for t in tags) + "   " + " ".join(f"{v.get(t, 0) / base:.3f}" if base else "-" for t in tags[1:]))
PY
