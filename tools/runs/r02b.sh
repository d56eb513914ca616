set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02b/pytest.log
tail -25 gpurun_out/r02b/pytest.log
TYRANT_HIP_LIBRARY=$PWD/tyrant_amd/lib/libtyrant_hip_stats.so timeout -k 10 200 python tools/loop_occupancy.py c3 production=1 > gpurun_out/r02b/loop_occupancy_c3.txt 2>&1; echo "occ rc $?"
cat gpurun_out/r02b/loop_occupancy_c3.txt
