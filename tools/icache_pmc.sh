# tools/icache_pmc.sh -- the render kernels' instruction-cache counters (SQC_ICACHE_REQ / HITS / MISSES) over tools/stream_probe.py: profiles/r06_pmc_icache.txt
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
d=$root/gpurun_out/icache; rm -rf $d; mkdir -p $d
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $d -o pmc -- python3 $root/tools/stream_probe.py renders=3 > $d/log.txt 2>&1
tail -2 $d/log.txt
python3 - $d <<'PY'
import csv, glob, os, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    if c.get("SQC_ICACHE_REQ", 0) > 1e5:
        print(k, {n: f"{v:.4g}" for n, v in c.items()}, "miss rate", round(c.get("SQC_ICACHE_MISSES", 0) / max(c["SQC_ICACHE_REQ"], 1), 5))
PY
