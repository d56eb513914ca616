# tools/steal_whatif.sh -- round 6: the drain's hand-offs (-DTYR_WIDE_STEAL, hip/traverse_flat.hip wide_drain_steal) in one gpurun call:
# (1) the staged parity tests with a guarded build of it (a wave that makes no progress gives up with kErrNoProgress instead of
# holding the GPU), (2) the per-wave anatomy of a C3 render's drains with and without, (3) C3 renders back to back at both queue sizes.
set -o pipefail
O=gpurun_out/steal; mkdir -p $O
L=$PWD/tyrant_amd/lib
echo "== parity, guarded steal build" > $O/parity.txt
TYRANT_HIP_LIBRARY=$L/libtyrant_hip_stealg.so timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider >> $O/parity.txt 2>&1
echo "parity rc $?" >> $O/parity.txt
grep -q "NoProgress\|no progress" $O/parity.txt && { echo "guard tripped: stopping"; exit 1; }
for t in anatomy steal_anatomy; do
  TYRANT_HIP_LIBRARY=$L/libtyrant_hip_$t.so TYR_ANATOMY=2 timeout -k 10 200 python3 bench.py --pmc-child --workload c3 --width 1920 --height 1080 --spp 8 --queue 0 > $O/$t.out 2> $O/$t.txt || exit 1
done
bash tools/lib_ab_n.sh 5 steal > $O/ab_16M.txt 2>&1 || exit 1
PROBE_KNOBS="queue=2097152" bash tools/lib_ab_n.sh 3 steal > $O/ab_2Mi.txt 2>&1 || exit 1
cat $O/ab_16M.txt $O/ab_2Mi.txt; tail -3 $O/parity.txt
