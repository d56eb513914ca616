cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02r
L=$PWD/tyrant_amd/lib
TYRANT_HIP_LIBRARY=$L/libtyrant_hip_anatomy.so timeout -k 10 200 python tools/drain_profile.py c3 > gpurun_out/r02r/drain_profile_c3.txt 2>&1
grep -v amdgpu gpurun_out/r02r/drain_profile_c3.txt
