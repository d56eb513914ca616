cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ay
O=gpurun_out/r02ay
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 && tail -2 $O/pytest.log \
 && ( for i in 1 2 3; do NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py; done; timeout -k 10 120 python tools/render_timing.py ) > $O/timing.txt 2>&1 \
 && grep -v amdgpu $O/timing.txt | grep " N " \
 && timeout -k 10 400 python bench.py --save-pmc $O/pmc_c3.json > $O/bench_c3.json 2> $O/bench_c3.err && cut -c1-200 $O/bench_c3.json
rc=$?; echo "chain rc $rc"; [ $rc -ne 0 ] && { tail -20 $O/pytest.log; tail -5 $O/bench_c3.err 2>/dev/null; }; exit $rc
