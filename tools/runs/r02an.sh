cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02an
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r02an/pytest.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02an/pytest.log
timeout -k 10 400 python tests/fuzz_parity.py 1500 99 > gpurun_out/r02an/fuzz.txt 2>&1; echo "fuzz rc $?"; grep -c " -> ok" gpurun_out/r02an/fuzz.txt; grep "FAIL" gpurun_out/r02an/fuzz.txt | head -5
( for ra in 0 1 0 1; do
    NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py run_ahead=$ra
    timeout -k 10 120 python tools/render_timing.py run_ahead=$ra
  done
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py 2097152 run_ahead=0
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py 2097152 run_ahead=1
) > gpurun_out/r02an/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02an/ab.txt
