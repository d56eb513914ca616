cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02m
L=$PWD/tyrant_amd/lib
TYR_ANATOMY=1 NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_stats.so timeout -k 10 120 python tools/render_timing.py > gpurun_out/r02m/longest.txt 2>&1
grep -v amdgpu gpurun_out/r02m/longest.txt | grep "iteration [0-5]:\|ms/render"
