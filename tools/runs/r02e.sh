set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
L=$PWD/tyrant_amd/lib
( for lib in feedA feedB feedC; do
    TYRANT_HIP_LIBRARY=$L/libtyrant_hip_$lib.so timeout -k 10 120 python tools/render_timing.py traversal_variant=5
  done
  TYRANT_HIP_LIBRARY=$L/libtyrant_hip_feedA.so timeout -k 10 120 python tools/render_timing.py traversal_variant=4 waves_per_simd=4
) > gpurun_out/r02e/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02e/ab.txt
TYRANT_HIP_LIBRARY=$L/libtyrant_hip_feedAs.so timeout -k 10 200 python tools/loop_occupancy.py c3 production=1 traversal_variant=5 > gpurun_out/r02e/loop_occupancy_c3_v5.txt 2>&1
grep -v amdgpu gpurun_out/r02e/loop_occupancy_c3_v5.txt
