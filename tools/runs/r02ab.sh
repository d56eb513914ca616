cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ab
L=$PWD/tyrant_amd/lib
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02ab/pytest_parity.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02ab/pytest_parity.log
( for lib in base regroup regroup3; do
    TYRANT_HIP_LIBRARY=$L/libtyrant_hip_$lib.so timeout -k 10 120 python tools/render_timing.py
    NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_$lib.so timeout -k 10 120 python tools/render_timing.py
  done ) > gpurun_out/r02ab/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02ab/ab.txt
