cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02as
O=gpurun_out/r02as
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest.log 2>&1 && tail -2 $O/pytest.log \
 && timeout -k 10 400 python bench.py --workload c2 --save-pmc $O/pmc_c2.json > $O/bench_c2.json 2> $O/bench_c2.err \
 && timeout -k 10 500 python bench.py --workload c5 --width 3840 --height 2160 --spp 16 --no-cpu-baseline --no-reference-queue --save-pmc $O/pmc_c5.json > $O/bench_c5.json 2> $O/bench_c5.err \
 && cut -c1-150 $O/bench_c2.json && cut -c1-150 $O/bench_c5.json
rc=$?; echo "chain rc $rc"; [ $rc -ne 0 ] && { tail -20 $O/pytest.log; tail -5 $O/bench_c2.err $O/bench_c5.err 2>/dev/null; }; exit $rc
