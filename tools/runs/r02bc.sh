cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bc
export TYRANT_HIP_LIBRARY=$PWD/tyrant_amd/lib/libtyrant_hip_anatomy.so
( timeout -k 10 200 python tools/drain_profile.py c3 wide_drain=0
  timeout -k 10 200 python tools/drain_profile.py c3 wide_drain=1 ) > gpurun_out/r02bc/drain.txt 2>&1
grep -v amdgpu gpurun_out/r02bc/drain.txt
