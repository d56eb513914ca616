cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
( timeout -k 10 120 python tools/render_timing.py merge_trace=0
  timeout -k 10 120 python tools/render_timing.py merge_trace=1
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py merge_trace=0
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py merge_trace=1
) > gpurun_out/r02i/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02i/ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "render or deferred or full_size or sharded or edge" > gpurun_out/r02i/pytest.log 2>&1; tail -5 gpurun_out/r02i/pytest.log
