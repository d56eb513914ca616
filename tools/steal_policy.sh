# tools/steal_policy.sh -- round 6: hand-off policies of the -DTYR_WIDE_STEAL what-if back to back on one box (tagged builds st_*), and
# the per-wave anatomy of each (st_*_an, fold_spheres=0 so that shade leaves the anatomy records alone)
O=gpurun_out/steal; mkdir -p $O
TAGS="st_a0n2 st_a6n1 st_a6n2 st_a12n2 st_a12n3"
bash tools/lib_ab_n.sh 4 $TAGS > $O/policy_16M.txt 2>&1 || exit 1
PROBE_KNOBS="queue=2097152" bash tools/lib_ab_n.sh 3 $TAGS > $O/policy_2Mi.txt 2>&1 || exit 1
for t in $TAGS; do
  TYRANT_HIP_LIBRARY=$PWD/tyrant_amd/lib/libtyrant_hip_${t}_an.so TYR_ANATOMY=2 timeout -k 10 150 python3 bench.py --pmc-child --workload c3 --width 1920 --height 1080 --spp 8 --queue 0 --tune fold_spheres=0 > /dev/null 2> $O/${t}_an.txt || exit 1
done
cat $O/policy_16M.txt $O/policy_2Mi.txt
