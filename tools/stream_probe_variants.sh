export TYR_PROBE_DEBUG=1 TYRANT_HIP_LIBRARY=$PWD/tyrant_amd/lib/libtyrant_hip_shtime.so
for kn in "stream_trace_per_cu=4 stream_shade_per_cu=1" "stream_trace_per_cu=2 stream_shade_per_cu=2" "stream_trace_per_cu=1 stream_shade_per_cu=1"; do
  echo "== $kn"; timeout -k 10 100 python3 tools/stream_probe.py stream_tail=1 run_ahead=0 renders=2 $kn | tail -2
done
