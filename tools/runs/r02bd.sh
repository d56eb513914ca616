cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bd
L=$PWD/tyrant_amd/lib
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r02bd/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r02bd/pytest.log
export NOPROFILE=1
( for i in 1 2 3; do
    TYRANT_HIP_LIBRARY=$L/libtyrant_hip_mainflat.so timeout -k 10 120 python tools/render_timing.py
    TYRANT_HIP_LIBRARY=$L/libtyrant_hip_diag.so timeout -k 10 120 python tools/render_timing.py
  done
  TYRANT_HIP_LIBRARY=$L/libtyrant_hip_mainflat.so timeout -k 10 120 python tools/render_timing.py 2097152
  TYRANT_HIP_LIBRARY=$L/libtyrant_hip_diag.so timeout -k 10 120 python tools/render_timing.py 2097152 ) > gpurun_out/r02bd/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02bd/ab.txt | grep " N "
