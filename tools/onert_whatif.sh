# tools/onert_whatif.sh -- round 6: the four-lanes-per-ray drain with ONE memory round trip per step (-DTYR_WIDE_ONE_RT), alone and under
# the hand-offs of -DTYR_WIDE_STEAL (tagged builds of the experiment source: onert, st12, st12rt, st6rt, st0rt), back to back on one box
O=gpurun_out/onert; mkdir -p $O
bash tools/lib_ab_n.sh 5 onert st12 st12rt st6rt st0rt > $O/ab_16M.txt 2>&1 || exit 1
PROBE_KNOBS="queue=2097152" bash tools/lib_ab_n.sh 3 onert st12 st12rt st6rt st0rt > $O/ab_2Mi.txt 2>&1 || exit 1
cat $O/ab_16M.txt $O/ab_2Mi.txt
for t in onert_an st12rt_an st6rt_an; do
  TYRANT_HIP_LIBRARY=$PWD/tyrant_amd/lib/libtyrant_hip_$t.so TYR_ANATOMY=2 timeout -k 10 150 python3 bench.py --pmc-child --workload c3 --width 1920 --height 1080 --spp 8 --queue 0 --tune fold_spheres=0 > /dev/null 2> $O/$t.txt || exit 1
done
