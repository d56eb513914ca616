# kernel stats of the bench command itself (no steady-state leg: only the timed shape's launches) and the timeline of one render
set -x
O=gpurun_out/r4n; mkdir -p $O; rm -rf /tmp/prof_c3_bench
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --pmc off --cpu-iterations 0 --no-reference-queue --no-steady-state --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_c3_under_rocprof.json.log 2> $GRAFT_REPO_ROOT/$O/bench_c3_under_rocprof.err )
f=$(find /tmp/prof_c3_bench -name "*kernel_stats.csv" | head -1); cp $f $O/bench_c3_kernel_stats.csv
t=$(find /tmp/prof_c3_bench -name "*kernel_trace.csv" | head -1); python tools/render_timeline.py $t > $O/timeline_c3.txt
head -5 $O/bench_c3_kernel_stats.csv; cat $O/timeline_c3.txt | tail -45
