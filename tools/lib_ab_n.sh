# tools/lib_ab_n.sh <rounds> <tag> [<tag> ...] -- like tools/lib_ab.sh, <rounds> interleaved rounds, then min / median per library (for effects of ~1 %)
rounds=$1; shift
for r in $(seq $rounds); do
  for tag in product "$@"; do
    lib=$PWD/tyrant_amd/lib/libtyrant_hip.so
    [ "$tag" != product ] && lib=$PWD/tyrant_amd/lib/libtyrant_hip_$tag.so
    echo -n "$tag "; TYRANT_HIP_LIBRARY=$lib timeout -k 10 100 python3 tools/stream_probe.py renders=4 ${PROBE_KNOBS} 2>&1 | grep "^render" | sort -t: -k2 -n | head -1 | sed 's/render [0-9]*: //; s/ ms.*//'
  done
done | python3 -c "
import sys, collections, statistics
d = collections.OrderedDict()
for l in sys.stdin:
    t, v = l.split(); d.setdefault(t, []).append(float(v))
for t, v in d.items():
    print(f'{t:10s} min {min(v):.3f}  median {statistics.median(v):.3f}  ({len(v)} rounds: ' + ' '.join(f'{x:.3f}' for x in v) + ')')
"
