cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02al
timeout -k 10 1000 python tests/fuzz_parity.py 500 20261004 > gpurun_out/r02al/fuzz_500.txt 2>&1; echo "rc $?"
grep -v amdgpu gpurun_out/r02al/fuzz_500.txt | grep -c " -> ok"; grep -v amdgpu gpurun_out/r02al/fuzz_500.txt | grep "FAIL\|failures\|Error" | head -20
