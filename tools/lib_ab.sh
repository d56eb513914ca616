# tools/lib_ab.sh <tag> [<tag> ...] -- C3 renders (tools/stream_probe.py, launch-per-iteration path) with the product library and
# with tagged what-if builds of it (make tagged TAG=...), back to back on one box: the last of four renders each, twice over
for round in 1 2; do
  for tag in product "$@"; do
    lib=$PWD/tyrant_amd/lib/libtyrant_hip.so
    [ "$tag" != product ] && lib=$PWD/tyrant_amd/lib/libtyrant_hip_$tag.so
    echo -n "$tag: "; TYRANT_HIP_LIBRARY=$lib timeout -k 10 100 python3 tools/stream_probe.py renders=4 ${PROBE_KNOBS} 2>&1 | grep "^render" | sort -t: -k2 -n | head -1
  done
done
