cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ax
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02ax/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02ax/pytest.log
timeout -k 10 600 python tests/fuzz_parity.py 3000 4242 > gpurun_out/r02ax/fuzz.txt 2>&1; echo "fuzz rc $?"; grep -c " -> ok" gpurun_out/r02ax/fuzz.txt; grep "FAIL" gpurun_out/r02ax/fuzz.txt | head -5
