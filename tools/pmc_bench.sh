#!/bin/bash
# tools/pmc_bench.sh <tag> <workload> <counter group 1> -- <counter group 2> -- ...
# one rocprofv3 --pmc pass per counter group over one bench.py render (--steps 1 --warmup 0, no CPU baseline, no
# second queue size); per-kernel sums under gpurun_out/pmcb_<tag>/summary.txt.  Small groups only: a pass with too many
# counters fails, and --pmc is never combined with tracing (MI355X_MICROARCH.md "rocprofv3 PMC").
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; wl=$2; shift 2
out=$R/gpurun_out/pmcb_$tag
mkdir -p $out
cd /tmp
i=0
group=()
run_group() {
  [ ${#group[@]} -eq 0 ] && return 0
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc "${group[@]}" --output-format csv -d $out -o g$i -- python3 $R/bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-reference-queue > $out/g$i.log 2>&1 || { echo "group $i (${group[*]}) failed"; tail -3 $out/g$i.log; return 1; }
  group=()
}
for a in "$@"; do
  if [ "$a" = "--" ]; then run_group || exit 1; else group+=("$a"); fi
done
run_group || exit 1
python3 $R/tools/pmc_summarize.py $out > $out/summary.txt
cat $out/summary.txt
