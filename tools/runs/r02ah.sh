cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ah
L=$PWD/tyrant_amd/lib
( TYRANT_HIP_LIBRARY=$L/libtyrant_hip_diag.so timeout -k 10 120 python tools/render_timing.py
  TYRANT_HIP_LIBRARY=$L/libtyrant_hip_skipsky.so timeout -k 10 120 python tools/render_timing.py
  NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_diag.so timeout -k 10 120 python tools/render_timing.py
  NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_skipsky.so timeout -k 10 120 python tools/render_timing.py
) > gpurun_out/r02ah/whatif.txt 2>&1
grep -v amdgpu gpurun_out/r02ah/whatif.txt
