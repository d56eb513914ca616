cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ar
export MASTER_ADDR=127.0.0.1 TYR_BENCH_PREFLIGHT_ONE_DEVICE=1
( time timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 0 --backend gloo --workload c1 --width 320 --height 180 --queue 32768 --spp 4 --no-reference-queue ) > gpurun_out/r02ar/out.txt 2> gpurun_out/r02ar/err.txt
echo rc $?
grep -v "amdgpu.ids" gpurun_out/r02ar/err.txt | tail -40
