cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ak
timeout -k 10 600 python tools/strong_scaling_model.py > gpurun_out/r02ak/strong_scaling_model.txt 2>&1; echo "rc $?"
grep -v amdgpu gpurun_out/r02ak/strong_scaling_model.txt
