cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bi
L=$PWD/tyrant_amd/lib
export NOPROFILE=1
( for i in 1 2 3; do
    for lib in diag prio1 prio3; do TYRANT_HIP_LIBRARY=$L/libtyrant_hip_$lib.so timeout -k 10 120 python tools/render_timing.py; done
  done
  for lib in diag prio3; do TYRANT_HIP_LIBRARY=$L/libtyrant_hip_$lib.so timeout -k 10 120 python tools/render_timing.py 2097152; done ) > gpurun_out/r02bi/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02bi/ab.txt | grep "c3 N"
