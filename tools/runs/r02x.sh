cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02x
export TMPDIR=/tmp
R=$PWD
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r02x/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02x/pytest.log
tail -10 gpurun_out/r02x/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02x/smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r02x/smoke.log
timeout -k 10 400 python bench.py --save-pmc gpurun_out/r02x/pmc_c3.json > gpurun_out/r02x/bench_c3.json 2> gpurun_out/r02x/bench_c3.err; echo "bench c3 rc $?"
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02x/prof_c3 -o c3 -- python3 $R/bench.py --pmc off --no-cpu-baseline --no-reference-queue > $R/gpurun_out/r02x/bench_c3_under_rocprof.json 2> $R/gpurun_out/r02x/rocprof_c3.err ); echo "rocprof c3 rc $?"
cut -c1-200 gpurun_out/r02x/bench_c3.json
