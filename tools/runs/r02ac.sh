cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ac
L=$PWD/tyrant_amd/lib
R=$PWD
export TMPDIR=/tmp NOPROFILE=1 PYTHONPATH=$R
cd /tmp
for lib in base regroup; do
  export TYRANT_HIP_LIBRARY=$L/libtyrant_hip_$lib.so
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r02ac/$lib -o g1 -- python3 $R/tools/render_timing.py > $R/gpurun_out/r02ac/$lib.log 2>&1 || { echo "$lib failed"; tail -5 $R/gpurun_out/r02ac/$lib.log; exit 1; }
done
cd $R
for lib in base regroup; do echo "== $lib"; python tools/pmc_summarize.py gpurun_out/r02ac/$lib | grep -A9 "== k_shade"; done
