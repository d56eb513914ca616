cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02aj
L=$PWD/tyrant_amd/lib
( for lib in shb4 shb5 shb6 shb4 shb5; do
    TYRANT_HIP_LIBRARY=$L/libtyrant_hip_$lib.so timeout -k 10 120 python tools/render_timing.py
    NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_$lib.so timeout -k 10 120 python tools/render_timing.py
  done ) > gpurun_out/r02aj/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02aj/ab.txt
