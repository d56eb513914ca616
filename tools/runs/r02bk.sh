cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bk
O=gpurun_out/r02bk
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 && tail -2 $O/pytest.log \
 && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log \
 && timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err && cut -c1-220 $O/bench_default.json
rc=$?; echo "chain rc $rc"; [ $rc -ne 0 ] && { tail -20 $O/pytest.log; tail -5 $O/smoke.log $O/bench_default.err 2>/dev/null; }; exit $rc
