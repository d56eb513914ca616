cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02aq
timeout -k 10 900 python -m pytest tests/test_bench_contract.py -m gpu -x -q --durations=6 > gpurun_out/r02aq/pytest.log 2>&1; rc=$?; tail -25 gpurun_out/r02aq/pytest.log; exit $rc
