cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02n
L=$PWD/tyrant_amd/lib
( NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py
  NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_touch.so timeout -k 10 120 python tools/render_timing.py
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py static_share=12
  NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_touch.so timeout -k 10 120 python tools/render_timing.py static_share=12
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py stack_lds_depth=16
) > gpurun_out/r02n/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02n/ab.txt
TYR_ANATOMY=1 NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_toucha.so timeout -k 10 120 python tools/render_timing.py 2>&1 | grep -v amdgpu | tail -8
