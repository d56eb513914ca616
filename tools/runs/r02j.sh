cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02j/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02j/pytest.log
tail -14 gpurun_out/r02j/pytest.log
timeout -k 10 400 python bench.py --save-pmc gpurun_out/r02j/pmc_c3.json > gpurun_out/r02j/bench_c3.json 2> gpurun_out/r02j/bench_c3.err; echo "bench rc $?"
R=$PWD
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02j/prof_c3 -o c3 -- python3 $R/bench.py --pmc off --no-cpu-baseline --no-reference-queue > $R/gpurun_out/r02j/bench_c3_under_rocprof.json 2> $R/gpurun_out/r02j/rocprof_c3.err ); echo "rocprof rc $?"
ls gpurun_out/r02j/prof_c3 gpurun_out/r02j/prof_c3/* | head
cut -c1-600 gpurun_out/r02j/bench_c3.json
