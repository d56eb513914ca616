cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02w
TYRANT_HIP_LIBRARY=$PWD/tyrant_amd/lib/libtyrant_hip_shadetiming.so timeout -k 10 120 python tools/shade_phases.py c3 > gpurun_out/r02w/shade_phases_c3.txt 2>&1
grep -v amdgpu gpurun_out/r02w/shade_phases_c3.txt
