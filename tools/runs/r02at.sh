cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02at
export NOPROFILE=1 TYRANT_HIP_LIBRARY=$PWD/tyrant_amd/lib/libtyrant_hip_thin.so
( for tw in 0 4 3 2 0 4; do TYR_THIN_WAVES=$tw timeout -k 10 120 python tools/render_timing.py; echo "^ thin waves $tw (items < 6M)"; done
  for tw in 4 3; do TYR_THIN_WAVES=$tw TYR_THIN_ITEMS=3000000 timeout -k 10 120 python tools/render_timing.py; echo "^ thin waves $tw (items < 3M)"; done
) > gpurun_out/r02at/thin.txt 2>&1
grep -v amdgpu gpurun_out/r02at/thin.txt
