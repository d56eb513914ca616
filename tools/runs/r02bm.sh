cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh final_c3 --pmc off --no-reference-queue --steps 1 --warmup 0 > gpurun_out/pmc_final_c3.log 2>&1; rc=$?
python tools/pmc_summarize.py gpurun_out/pmc_final_c3 > gpurun_out/pmc_final_c3_summary.txt 2>&1
grep -A45 "== k_trace_flat" gpurun_out/pmc_final_c3_summary.txt
tail -3 gpurun_out/pmc_final_c3.log
exit 0
