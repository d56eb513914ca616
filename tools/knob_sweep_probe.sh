# tools/knob_sweep_probe.sh <rounds> "<knobs A>" "<knobs B>" ... -- tools/stream_probe.py (4 C3 renders, the fastest counts) with each
# knob set in turn, <rounds> interleaved rounds on one box, then min / median per set ("" = the defaults)
rounds=$1; shift
for r in $(seq $rounds); do
  for k in "$@"; do
    echo -n "[${k:-defaults}] "; timeout -k 10 100 python3 tools/stream_probe.py renders=4 $k 2>&1 | grep "^render" | sort -t: -k2 -n | head -1 | sed 's/render [0-9]*: //; s/ ms.*//'
  done
done | python3 -c "
import sys, collections, statistics
d = collections.OrderedDict()
for l in sys.stdin:
    t, v = l.rsplit(' ', 1); d.setdefault(t, []).append(float(v))
for t, v in d.items():
    print(f'{t:44s} min {min(v):.3f}  median {statistics.median(v):.3f}  (' + ' '.join(f'{x:.3f}' for x in v) + ')')
"
