cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02be
( timeout -k 10 300 python tools/soak.py 400 c3 && timeout -k 10 300 python tools/soak.py 400 c3 2097152 && timeout -k 10 300 python tools/soak.py 300 c2 && timeout -k 10 300 python tools/soak.py 300 c2 2097152 ) > gpurun_out/r02be/soak.txt 2>&1; rc=$?
grep -v amdgpu gpurun_out/r02be/soak.txt | grep "renders of\|Error\|assert" ; exit $rc
