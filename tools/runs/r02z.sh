cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02z
timeout -k 10 500 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --no-reference-queue > gpurun_out/r02z/bench_c4_2ranks_gloo.json 2> gpurun_out/r02z/bench_c4_2ranks_gloo.err; echo "rc $?"
tail -c 400 gpurun_out/r02z/bench_c4_2ranks_gloo.err
cut -c1-1500 gpurun_out/r02z/bench_c4_2ranks_gloo.json
