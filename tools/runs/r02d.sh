set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02d/pytest.log
tail -15 gpurun_out/r02d/pytest.log
L=$PWD/tyrant_amd/lib
( timeout -k 10 120 python tools/render_timing.py
  timeout -k 10 120 python tools/render_timing.py traversal_variant=5
  timeout -k 10 120 python tools/render_timing.py traversal_variant=5 min_traversing=24
  timeout -k 10 120 python tools/render_timing.py traversal_variant=5 min_traversing=40
  TYRANT_HIP_LIBRARY=$L/libtyrant_hip_feedw4.so timeout -k 10 120 python tools/render_timing.py traversal_variant=5
  TYRANT_HIP_LIBRARY=$L/libtyrant_hip_feeds10.so timeout -k 10 120 python tools/render_timing.py traversal_variant=5
  TYRANT_HIP_LIBRARY=$L/libtyrant_hip_feedpop4.so timeout -k 10 120 python tools/render_timing.py traversal_variant=5
) > gpurun_out/r02d/ab.txt 2>&1
cat gpurun_out/r02d/ab.txt
TYRANT_HIP_LIBRARY=$L/libtyrant_hip_stats.so timeout -k 10 200 python tools/loop_occupancy.py c3 production=1 traversal_variant=5 > gpurun_out/r02d/loop_occupancy_c3_v5.txt 2>&1
cat gpurun_out/r02d/loop_occupancy_c3_v5.txt
