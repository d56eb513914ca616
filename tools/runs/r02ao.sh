cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ao
( for ra in 0 1 0 1 0 1; do
    NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py run_ahead=$ra
  done
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py 2097152 run_ahead=0
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py 2097152 run_ahead=1
) > gpurun_out/r02ao/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02ao/ab.txt
