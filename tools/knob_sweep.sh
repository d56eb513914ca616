# tools/knob_sweep.sh -- C3 renders (tools/stream_probe.py, best of 4) over a grid of the traversal's launch-shape knobs
for ri in 8 16 24 32; do for mt in 16 32 48; do
  echo -n "refill_min_idle=$ri min_traversing=$mt: "; timeout -k 10 100 python3 tools/stream_probe.py renders=4 refill_min_idle=$ri min_traversing=$mt 2>&1 | grep "^render" | sort -t: -k2 -n | head -1
done; done
for ss in 8 10 12 14; do echo -n "static_share=$ss: "; timeout -k 10 100 python3 tools/stream_probe.py renders=4 static_share=$ss 2>&1 | grep "^render" | sort -t: -k2 -n | head -1; done
for sn in 32 48 64; do echo -n "staged_nodes=$sn: "; timeout -k 10 100 python3 tools/stream_probe.py renders=4 staged_nodes=$sn 2>&1 | grep "^render" | sort -t: -k2 -n | head -1; done
