cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02aw
L=$PWD/tyrant_amd/lib
export NOPROFILE=1
( for i in 1 2; do
    TYRANT_HIP_LIBRARY=$L/libtyrant_hip_mainflat.so timeout -k 10 120 python tools/render_timing.py 2097152
    TYRANT_HIP_LIBRARY=$L/libtyrant_hip_diag.so timeout -k 10 120 python tools/render_timing.py 2097152 wide_drain=1
  done ) > gpurun_out/r02aw/ab_2mi.txt 2>&1
grep -v amdgpu gpurun_out/r02aw/ab_2mi.txt
TYRANT_HIP_LIBRARY=$L/libtyrant_hip_mainflat.so timeout -k 10 300 python bench.py --workload c5 --width 3840 --height 2160 --spp 16 --no-cpu-baseline --no-reference-queue --pmc off 2>/dev/null | cut -c1-140
TYRANT_HIP_LIBRARY=$L/libtyrant_hip_diag.so timeout -k 10 300 python bench.py --workload c5 --width 3840 --height 2160 --spp 16 --no-cpu-baseline --no-reference-queue --pmc off 2>/dev/null | cut -c1-140
