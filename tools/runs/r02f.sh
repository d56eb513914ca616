cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
L=$PWD/tyrant_amd/lib
TYRANT_HIP_LIBRARY=$L/libtyrant_hip_stats.so timeout -k 10 200 python tools/launch_tail.py c3 > gpurun_out/r02f/launch_tail_c3.txt 2>&1
TYRANT_HIP_LIBRARY=$L/libtyrant_hip_stats.so timeout -k 10 200 python tools/launch_tail.py c2 > gpurun_out/r02f/launch_tail_c2.txt 2>&1
TYRANT_HIP_LIBRARY=$L/libtyrant_hip_stats.so timeout -k 10 200 python tools/launch_tail.py c3 waves_per_simd=3 > gpurun_out/r02f/launch_tail_c3_w3.txt 2>&1
grep -v amdgpu gpurun_out/r02f/*.txt
