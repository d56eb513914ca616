# tools/round6_measure.sh -- the round's measurements in one gpurun call (everything lands under gpurun_out/r6m/)
set -x
O=gpurun_out/r6m; mkdir -p $O
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --save-pmc $O/pmc_c3.json > $O/bench_c3.json.log 2> $O/bench_c3.err || exit 1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3_full -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --pmc off --cpu-iterations 0 --no-reference-queue --no-steady-state --no-framed --no-spread > $GRAFT_REPO_ROOT/$O/bench_c3_under_rocprof.json.log 2> $GRAFT_REPO_ROOT/$O/bench_c3_under_rocprof.err ) || exit 1
f=$(find /tmp/prof_c3_full -name "*kernel_stats.csv" | head -1); cp $f $O/bench_c3_kernel_stats.csv
t=$(find /tmp/prof_c3_full -name "*kernel_trace.csv" | head -1); python tools/render_timeline.py $t > $O/timeline_c3.txt
timeout -k 10 300 python bench.py --workload c2 --steps 10 --warmup 2 --save-pmc $O/pmc_c2.json > $O/bench_c2.json.log 2> $O/bench_c2.err || exit 1
timeout -k 10 400 python bench.py --workload c5 --width 3840 --height 2160 --spp 16 --steps 2 --warmup 1 --no-reference-queue --cpu-iterations 1 --save-pmc $O/pmc_c5.json > $O/bench_c5.json.log 2> $O/bench_c5.err || exit 1
timeout -k 10 200 python tools/strong_scaling_model.py > $O/strong_scaling_model_c4.txt 2>&1
ls -la $O
