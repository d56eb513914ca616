cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
timeout -k 10 300 python tools/shard_probe.py c3 1 2 3 4 6 8 > gpurun_out/r02h/shards_c3.txt 2>&1
timeout -k 10 300 python tools/shard_probe.py c2 1 2 4 > gpurun_out/r02h/shards_c2.txt 2>&1
timeout -k 10 300 python tools/shard_probe.py c3 1 2 4 overlap_connect=1 > gpurun_out/r02h/shards_c3_ov1.txt 2>&1
grep -v amdgpu gpurun_out/r02h/*.txt
