echo "== parity, product"; timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_stream_tail.py -m gpu -x -q 2>&1 | tail -2
bash tools/lib_ab.sh before
PROBE_KNOBS="queue=2097152" bash tools/lib_ab.sh before
for tag in product before; do
  lib=$PWD/tyrant_amd/lib/libtyrant_hip.so
  [ "$tag" != product ] && lib=$PWD/tyrant_amd/lib/libtyrant_hip_$tag.so
  echo -n "$tag: "; TYRANT_HIP_LIBRARY=$lib timeout -k 10 100 python3 tools/stream_probe.py stream_tail=0 renders=3 profile=1 2>&1 | grep "^stages"
done
