cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02v
L=$PWD/tyrant_amd/lib
( NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py reverse_fresh=0
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py reverse_fresh=1
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py reverse_fresh=0
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py reverse_fresh=1
) > gpurun_out/r02v/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02v/ab.txt
TYR_ANATOMY=1 NOPROFILE=1 TYRANT_HIP_LIBRARY=$L/libtyrant_hip_anatomy.so timeout -k 10 120 python tools/render_timing.py reverse_fresh=1 2>&1 | grep -v amdgpu | grep "iteration \(18\|19\|20\)\|ms/render" | tail -5
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "render or deferred or full_size" > gpurun_out/r02v/pytest.log 2>&1; tail -3 gpurun_out/r02v/pytest.log
