set -o pipefail
O=gpurun_out/r06b; mkdir -p $O
bash tools/lib_ab_n.sh 4 lm2 lm3 lm4 lm6 > $O/leafmin_16M.txt 2>&1 || exit 1
PROBE_KNOBS="queue=2097152" bash tools/lib_ab_n.sh 3 lm2 lm3 lm4 lm6 > $O/leafmin_2Mi.txt 2>&1 || exit 1
cat $O/leafmin_16M.txt $O/leafmin_2Mi.txt
bash tools/knob_sweep_probe.sh 5 "fresh_shade=1" "fresh_shade=0" > $O/fresh_shade_ab.txt 2>&1 || exit 1
cat $O/fresh_shade_ab.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider --durations=8 > $O/gputest_new.txt 2>&1; echo "gputest rc $?" >> $O/gputest_new.txt
tail -15 $O/gputest_new.txt
