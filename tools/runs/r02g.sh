cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
L=$PWD/tyrant_amd/lib
for w in 0 4 3 2; do
TYRANT_HIP_LIBRARY=$L/libtyrant_hip_anatomy.so timeout -k 10 200 python tools/launch_tail.py c3 waves_per_simd=$w > gpurun_out/r02g/launch_tail_c3_w$w.txt 2>&1
done
TYRANT_HIP_LIBRARY=$L/libtyrant_hip_anatomy.so timeout -k 10 200 python tools/launch_tail.py c2 > gpurun_out/r02g/launch_tail_c2.txt 2>&1
grep -v amdgpu gpurun_out/r02g/*.txt
