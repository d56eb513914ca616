#!/bin/bash
# tools/pmc_run.sh <tag> <bench args...> -- rocprofv3 PMC passes (one counter group per run, no tracing
# domains combined with --pmc) over one bench.py render; outputs under gpurun_out/pmc_<tag>/
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp
i=0
for group in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM" \
  "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES" \
  "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ" \
  "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d $out -o g$i -- python3 $R/bench.py "$@" --no-cpu-baseline > $out/g$i.log 2>&1 || { echo "group $i failed"; tail -5 $out/g$i.log; exit 1; }
done
ls $out
