cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02af
export NOPROFILE=1
( timeout -k 10 120 python tools/render_timing.py
  for sh in 4 8 12 16; do timeout -k 10 120 python tools/render_timing.py stagger_share=$sh; done
  for sh in 8 16 24; do timeout -k 10 120 python tools/render_timing.py stagger_share=$sh static_share=8; done
  for sh in 8 16; do timeout -k 10 120 python tools/render_timing.py stagger_share=$sh stagger_stay=2; done
  timeout -k 10 120 python tools/render_timing.py
) > gpurun_out/r02af/stagger.txt 2>&1
grep -v amdgpu gpurun_out/r02af/stagger.txt
