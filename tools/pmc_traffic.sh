#!/bin/bash
# tools/pmc_traffic.sh [bench args] -- HBM-side traffic of the bench's kernels: FETCH_SIZE and WRITE_SIZE in
# two separate rocprofv3 --pmc passes (they do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_traffic
mkdir -p $out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $out -o $c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-queue "$@" > $out/$c.log 2>&1 || { echo "$c failed"; tail -3 $out/$c.log; exit 1; }
done
python3 $R/tools/pmc_summarize.py $out > $out/summary.txt
cat $out/summary.txt
