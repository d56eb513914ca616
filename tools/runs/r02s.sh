cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02s
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deferred or drain_repacking" > gpurun_out/r02s/pytest.log 2>&1; tail -6 gpurun_out/r02s/pytest.log
( for t in 0 4 8 12 16; do NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py drain_repack=$t; done ) > gpurun_out/r02s/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02s/ab.txt
