cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02t
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deferred or block_merge" > gpurun_out/r02t/pytest.log 2>&1; tail -6 gpurun_out/r02t/pytest.log
( for t in 0 6 10 12 16; do NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py block_merge=$t; done ) > gpurun_out/r02t/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02t/ab.txt
