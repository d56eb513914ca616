cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02bb
export NOPROFILE=1
( timeout -k 10 120 python tools/render_timing.py
  for kv in static_share=8 static_share=10 static_share=14 static_share=15 refill_min_idle=8 refill_min_idle=12 refill_min_idle=24 min_traversing=24 min_traversing=40 ticket_chunk=128; do
    timeout -k 10 120 python tools/render_timing.py $kv
  done
  timeout -k 10 120 python tools/render_timing.py ) > gpurun_out/r02bb/sweep.txt 2>&1
grep -v amdgpu gpurun_out/r02bb/sweep.txt | grep " N "
