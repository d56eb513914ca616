cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02o
( for t in "static_share=12" "static_share=12 refill_min_idle=8" "static_share=12 refill_min_idle=24" "static_share=12 min_traversing=24" "static_share=12 min_traversing=40" "static_share=12 waves_per_simd=4" "static_share=12 ticket_chunk=128" "static_share=10" "static_share=14"; do
  NOPROFILE=1 timeout -k 10 120 python tools/render_timing.py $t
done ) > gpurun_out/r02o/ab.txt 2>&1
grep -v amdgpu gpurun_out/r02o/ab.txt
