# tools/shade_ab.sh <tag> [<tag> ...] -- the product library against tagged builds of it on one box: parity (the staged GPU tests with the
# tagged library), then C3 renders back to back (tools/lib_ab.sh) and the stages' hipEvent times of one profiled render each
set -u
for tag in "$@"; do
  echo "== parity, $tag"; TYRANT_HIP_LIBRARY=$PWD/tyrant_amd/lib/libtyrant_hip_$tag.so timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
done
bash tools/lib_ab.sh "$@"
PROBE_KNOBS="queue=2097152" bash tools/lib_ab.sh "$@"
for tag in product "$@"; do
  lib=$PWD/tyrant_amd/lib/libtyrant_hip.so
  [ "$tag" != product ] && lib=$PWD/tyrant_amd/lib/libtyrant_hip_$tag.so
  echo -n "$tag: "; TYRANT_HIP_LIBRARY=$lib timeout -k 10 100 python3 tools/stream_probe.py renders=3 profile=1 2>&1 | grep "^stages"
done
