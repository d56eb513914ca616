cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02av
L=$PWD/tyrant_amd/lib
export NOPROFILE=1
( for i in 1 2 3; do
    TYRANT_HIP_LIBRARY=$L/libtyrant_hip_mainflat.so timeout -k 10 120 python tools/render_timing.py
    TYRANT_HIP_LIBRARY=$L/libtyrant_hip_diag.so timeout -k 10 120 python tools/render_timing.py wide_drain=0
    TYRANT_HIP_LIBRARY=$L/libtyrant_hip_diag.so timeout -k 10 120 python tools/render_timing.py wide_drain=1
  done ) > gpurun_out/r02av/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02av/ab.txt
