#!/bin/bash
# tools/knob_ab.sh <rounds> "<knobs A>" "<knobs B>" [extra bench args] -- bench.py with two sets of --tune knobs, interleaved
# rounds on one box (run-to-run spread is ~1 %; box-to-box 2-3 %): Mrays/s and ms per render of every run, then min / median.
# knobs: "name=value name=value" ("" = defaults).  e.g.  tools/knob_ab.sh 5 "fold_prologue=0" ""
rounds=$1; A=$2; B=$3; shift 3
tune() { for kv in $1; do printf -- "--tune %s " "$kv"; done; }
run() { local k=$1; shift; python bench.py --steps 20 --warmup 3 --pmc off --no-cpu-baseline --no-steady-state $(tune "$k") "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['config']['reference_queue_size']['Mrays/s'] if 'reference_queue_size' in d['config'] else 0, d['config'].get('oracle_counters_match'))"; }
for i in $(seq $rounds); do
  a=$(run "$A" "$@"); b=$(run "$B" "$@")
  echo "round $i   A [$A]: $a    B [$B]: $b"
done
