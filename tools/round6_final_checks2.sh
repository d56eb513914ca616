# tools/round6_final_checks2.sh -- round 6, second set at HEAD: the C2 and C5 bench lines against their committed oracle counters, a 300-render
# C3 soak, 1000 more fuzz cases
set -o pipefail
O=gpurun_out/r6final2; mkdir -p $O
timeout -k 10 300 python3 bench.py --workload c2 --steps 10 --warmup 2 --pmc off --cpu-iterations 1 > $O/bench_c2.json.log 2> $O/bench_c2.err || exit 1
timeout -k 10 400 python3 bench.py --workload c5 --width 3840 --height 2160 --spp 16 --steps 2 --warmup 1 --no-reference-queue --cpu-iterations 1 --pmc off > $O/bench_c5.json.log 2> $O/bench_c5.err || exit 1
python3 - <<'PY'
import json
for w in ('c2','c5'):
    d=json.loads([l for l in open(f'gpurun_out/r6final2/bench_{w}.json.log') if l.startswith('{')][-1])
    print(w, d['value'], d['ms_per_step'], 'oracle_counters_match', d['config']['oracle_counters_match'])
PY
timeout -k 10 300 python3 tools/soak.py 300 c3 > $O/soak_c3_300.txt 2>&1; echo "soak rc $?" >> $O/soak_c3_300.txt; tail -3 $O/soak_c3_300.txt
timeout -k 10 900 python3 tests/fuzz_parity.py 1000 606 > $O/fuzz_1000.txt 2>&1; echo "fuzz rc $?" >> $O/fuzz_1000.txt; tail -2 $O/fuzz_1000.txt
