cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ag
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiny or edge" > gpurun_out/r02ag/pytest.log 2>&1; echo "rc $?"; tail -15 gpurun_out/r02ag/pytest.log
