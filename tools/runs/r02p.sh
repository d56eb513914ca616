cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02p
export TMPDIR=/tmp
R=$PWD
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r02p/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02p/pytest.log
tail -10 gpurun_out/r02p/pytest.log
# C3 (default line) + PMC json + kernel stats of the same command
timeout -k 10 400 python bench.py --save-pmc gpurun_out/r02p/pmc_c3.json > gpurun_out/r02p/bench_c3.json 2> gpurun_out/r02p/bench_c3.err; echo "bench c3 rc $?"
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02p/prof_c3 -o c3 -- python3 $R/bench.py --pmc off --no-cpu-baseline --no-reference-queue > $R/gpurun_out/r02p/bench_c3_under_rocprof.json 2> $R/gpurun_out/r02p/rocprof_c3.err ); echo "rocprof c3 rc $?"
# C2
timeout -k 10 400 python bench.py --workload c2 --save-pmc gpurun_out/r02p/pmc_c2.json > gpurun_out/r02p/bench_c2.json 2> gpurun_out/r02p/bench_c2.err; echo "bench c2 rc $?"
# C5 at its quoted size on one GPU
timeout -k 10 600 python bench.py --workload c5 --width 3840 --height 2160 --spp 16 --no-cpu-baseline --no-reference-queue --save-pmc gpurun_out/r02p/pmc_c5.json > gpurun_out/r02p/bench_c5.json 2> gpurun_out/r02p/bench_c5.err; echo "bench c5 rc $?"
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02p/prof_c5 -o c5 -- python3 $R/bench.py --workload c5 --width 3840 --height 2160 --spp 16 --pmc off --no-cpu-baseline --no-reference-queue > $R/gpurun_out/r02p/bench_c5_under_rocprof.json 2> $R/gpurun_out/r02p/rocprof_c5.err ); echo "rocprof c5 rc $?"
for f in gpurun_out/r02p/bench_c*.json; do echo $f; cut -c1-260 $f; done
timeout -k 10 120 python tools/concurrency_probe.py c3 > gpurun_out/r02p/concurrency_c3.txt 2>&1; grep -v amdgpu gpurun_out/r02p/concurrency_c3.txt
