cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02au
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r02au/pytest.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r02au/pytest.log
( for wd in 0 1 0 1; do NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py wide_drain=$wd; done
  timeout -k 10 120 python tools/render_timing.py wide_drain=0
  timeout -k 10 120 python tools/render_timing.py wide_drain=1 ) > gpurun_out/r02au/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02au/ab.txt
