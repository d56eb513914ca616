cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ap
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r02ap
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1 && tail -4 $O/pytest.log \
 && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log \
 && timeout -k 10 400 python bench.py --save-pmc $O/pmc_c3.json > $O/bench_c3.json 2> $O/bench_c3.err \
 && ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c3 -o c3 -- python3 $R/bench.py --pmc off --no-cpu-baseline --no-reference-queue > $R/$O/bench_c3_under_rocprof.json 2> $R/$O/rocprof_c3.err ) \
 && cut -c1-260 $O/bench_c3.json
rc=$?
echo "chain rc $rc"
[ $rc -ne 0 ] && { tail -20 $O/pytest.log; tail -5 $O/smoke.log 2>/dev/null; tail -5 $O/bench_c3.err 2>/dev/null; }
exit $rc
