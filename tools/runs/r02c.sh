set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=20 > gpurun_out/r02c/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02c/pytest.log
tail -40 gpurun_out/r02c/pytest.log
