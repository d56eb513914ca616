cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ba
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r02ba
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1 && tail -3 $O/pytest.log \
 && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log \
 && timeout -k 10 400 python bench.py --save-pmc $O/pmc_c3.json > $O/bench_c3.json 2> $O/bench_c3.err \
 && ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c3 -o c3 -- python3 $R/bench.py --pmc off --no-cpu-baseline --no-reference-queue > $R/$O/bench_c3_under_rocprof.json 2> $R/$O/rocprof_c3.err ) \
 && timeout -k 10 400 python bench.py --workload c2 --save-pmc $O/pmc_c2.json > $O/bench_c2.json 2> $O/bench_c2.err \
 && timeout -k 10 500 python bench.py --workload c5 --width 3840 --height 2160 --spp 16 --no-cpu-baseline --no-reference-queue --save-pmc $O/pmc_c5.json > $O/bench_c5.json 2> $O/bench_c5.err \
 && ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c5 -o c5 -- python3 $R/bench.py --workload c5 --width 3840 --height 2160 --spp 16 --pmc off --no-cpu-baseline --no-reference-queue > $R/$O/bench_c5_under_rocprof.json 2> $R/$O/rocprof_c5.err ) \
 && for w in c3 c2 c5; do cut -c1-150 $O/bench_$w.json; done
rc=$?
echo "chain rc $rc"
[ $rc -ne 0 ] && { tail -20 $O/pytest.log; tail -5 $O/smoke.log $O/bench_c3.err $O/bench_c2.err $O/bench_c5.err 2>/dev/null; }
exit $rc
