cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02l
L=$PWD/tyrant_amd/lib
( NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py static_interleave=0
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py static_interleave=1
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py static_interleave=1 static_share=8
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py static_interleave=1 static_share=12
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py static_interleave=1 static_share=15
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py static_interleave=1 static_share=2
) > gpurun_out/r02l/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02l/ab.txt
TYR_ANATOMY=1 NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_anatomy.so timeout -k 10 120 python tools/render_timing.py static_interleave=1 2>&1 | grep -v amdgpu | tail -8
