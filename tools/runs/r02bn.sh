cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bn
export NOPROFILE=1
( for i in 1 2 3; do
    timeout -k 10 120 python tools/render_timing.py static_interleave=1
    timeout -k 10 120 python tools/render_timing.py static_interleave=2
  done
  timeout -k 10 120 python tools/render_timing.py 2097152 static_interleave=1
  timeout -k 10 120 python tools/render_timing.py 2097152 static_interleave=2 ) > gpurun_out/r02bn/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02bn/ab.txt | grep " N "
