# tools/quiet_cu_probe.sh -- one C3 render by the -DTYR_LAUNCH_ANATOMY -DTYR_WHATIF_QUIET_CUS=16 build (make tagged TAG=quiet ...):
# per-wave feed-phase microseconds per trip on the CUs where only five waves work against the others (TYR_ANATOMY=2 printout)
TYRANT_HIP_LIBRARY=$PWD/tyrant_amd/lib/libtyrant_hip_quiet.so TYR_ANATOMY=2 timeout -k 10 120 python3 tools/stream_probe.py stream_tail=0 fold_spheres=0 renders=2 2>&1 | grep -E "iteration|feed phase|render"
