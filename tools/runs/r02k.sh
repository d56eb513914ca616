cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
L=$PWD/tyrant_amd/lib
TYR_ANATOMY=1 NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_anatomy.so timeout -k 10 120 python tools/render_timing.py > gpurun_out/r02k/anatomy_merged.txt 2>&1
grep -v amdgpu gpurun_out/r02k/anatomy_merged.txt | tail -32
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "c5 or deferred" > gpurun_out/r02k/pytest.log 2>&1; tail -5 gpurun_out/r02k/pytest.log
