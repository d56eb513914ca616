cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bl
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r02bl
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c2 -o c2 -- python3 $R/bench.py --workload c2 --pmc off --no-cpu-baseline --no-reference-queue > $R/$O/bench_c2_under_rocprof.json 2> $R/$O/rocprof_c2.err ); rc=$?
grep "k_trace_flat\|k_shade\|k_primary" $O/prof_c2/c2_kernel_stats.csv | cut -c1-150
exit $rc
