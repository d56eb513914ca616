# tools/round6_final_checks.sh -- the checks at HEAD in one gpurun call: the whole GPU suite (timed), smoke(), 500 fuzz cases, the default bench line
set -o pipefail
O=gpurun_out/r6final; mkdir -p $O
( time timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=6 ) > $O/gputest.txt 2>&1; echo "gputest rc $?" >> $O/gputest.txt
tail -12 $O/gputest.txt
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout -k 10 600 python3 tests/fuzz_parity.py 500 2026 > $O/fuzz_500.txt 2>&1; echo "fuzz rc $?" >> $O/fuzz_500.txt; tail -2 $O/fuzz_500.txt
timeout -k 10 400 python3 bench.py > $O/bench_default.json.log 2> $O/bench_default.err; echo "bench rc $?"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6final/bench_default.json.log') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['ms_per_step_spread'], d['config']['oracle_counters_match'], d['config']['framed']['Mrays/s'], d['config']['framed']['oracle_counters_match'], d['config']['spread']['ms_p50'], d['roofline']['nominal_step_frac'], d['roofline']['fabric_vs_gather_ceiling'])
PY
