set -o pipefail
O=gpurun_out/r06a; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gputest.txt 2>&1; echo "gputest rc $?" >> $O/gputest.txt
tail -3 $O/gputest.txt
bash tools/lib_ab_n.sh 4 r5 > $O/ab_vs_r5_16M.txt 2>&1 || exit 1
PROBE_KNOBS="queue=2097152" bash tools/lib_ab_n.sh 3 r5 > $O/ab_vs_r5_2Mi.txt 2>&1 || exit 1
cat $O/ab_vs_r5_16M.txt $O/ab_vs_r5_2Mi.txt
timeout -k 10 300 python3 bench.py > $O/bench_default.json.log 2> $O/bench_default.err; echo "bench rc $?"
tail -c 600 $O/bench_default.json.log
