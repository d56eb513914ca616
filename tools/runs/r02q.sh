cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02q
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "colored or colors or area_light or cpp_host or stage_by_stage or render_matches" > gpurun_out/r02q/pytest.log 2>&1; tail -6 gpurun_out/r02q/pytest.log
NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py 2>&1 | grep -v amdgpu
