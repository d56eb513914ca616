#!/bin/bash
# tools/pmc_ab.sh <tag> <counters...> -- one rocprofv3 --pmc pass over tools/ab_traverse.py (1 rep)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
out=$R/gpurun_out/pmcab_$tag
mkdir -p $out
cd /tmp
timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $out -o g -- python3 $R/tools/ab_traverse.py --workload ${WORKLOAD:-c2} --reps 1 --configs "${CONFIGS:-0:16:0:0,1:16:0:16}" > $out/run.log 2>&1 || { echo failed; tail -5 $out/run.log; exit 1; }
python3 $R/tools/pmc_summarize.py $out > $out/summary.txt
