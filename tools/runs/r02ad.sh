cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ad
export TYRANT_HIP_LIBRARY=$PWD/tyrant_amd/lib/libtyrant_hip_raysteps.so
timeout -k 10 800 python tools/ray_length_probe.py c3 gpurun_out/r02ad/rays_c3.npz > gpurun_out/r02ad/ray_length_c3.txt 2>&1; echo "rc $?"
grep -v amdgpu gpurun_out/r02ad/ray_length_c3.txt
