cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ai
L=$PWD/tyrant_amd/lib
echo skip pytest
( for lib in sh_base sh_tkt sh_lb sh_both sh_base sh_both; do
    TYRANT_HIP_LIBRARY=$L/libtyrant_hip_$lib.so timeout -k 10 120 python tools/render_timing.py
    NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_$lib.so timeout -k 10 120 python tools/render_timing.py
  done ) > gpurun_out/r02ai/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02ai/ab.txt
