#!/bin/bash
# Counter passes over the large prefilter (2048^2 x 10 mips): instruction mix, L1 / texture-address side, L2 side.
# One rocprofv3 run per pass, counters within the per-block limits.   usage: scripts/gpu_pmc_prefilter.sh <out-prefix>
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
cd /tmp
PASS=1
run() { timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pass$PASS -- python3 $R/scripts/gpu_prefilter_only.py > /dev/null 2>&1; PASS=$((PASS+1)); }
run SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
run TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/gpu_prefilter_only.py > /dev/null 2>&1
cd $R
{
  for d in $OUT/pass*; do python scripts/pmc_summary.py $(ls $d/*/*counter_collection.csv) "prefilter_kernel<1" "prefilter_levels_kernel"; done
  grep -h prefilter $OUT/trace/*/*kernel_stats.csv
} > $OUT/summary.txt
cat $OUT/summary.txt
