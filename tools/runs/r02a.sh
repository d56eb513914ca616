set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
timeout -k 10 400 python bench.py --save-pmc gpurun_out/r02a/pmc_c3.json > gpurun_out/r02a/bench_c3.json 2> gpurun_out/r02a/bench_c3.err; echo "bench rc $?"
tail -c 600 gpurun_out/r02a/bench_c3.err
TYRANT_HIP_LIBRARY=$PWD/tyrant_amd/lib/libtyrant_hip_stats.so timeout -k 10 200 python tools/loop_occupancy.py c3 production=1 > gpurun_out/r02a/loop_occupancy_c3.txt 2>&1; echo "occ rc $?"
cat gpurun_out/r02a/loop_occupancy_c3.txt
