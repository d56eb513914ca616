cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02y
L=$PWD/tyrant_amd/lib
( NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py
  NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_w6.so timeout -k 10 120 python tools/render_timing.py
  NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_w4.so timeout -k 10 120 python tools/render_timing.py
) > gpurun_out/r02y/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02y/ab.txt
