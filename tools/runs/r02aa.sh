cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh shade_c3 --pmc off --no-reference-queue --steps 1 --warmup 0 > gpurun_out/pmc_shade_c3.log 2>&1 || { tail -5 gpurun_out/pmc_shade_c3.log; exit 1; }
python tools/pmc_summarize.py gpurun_out/pmc_shade_c3 > gpurun_out/pmc_shade_c3_summary.txt
grep -A40 "== k_shade" gpurun_out/pmc_shade_c3_summary.txt
