#!/bin/bash
# Counter passes over a few 4K frames (one rocprofv3 run per pass, counters within the per-block limits).
# usage: scripts/gpu_pmc_frame.sh <out-prefix> [distance | stress]
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
DIST=${2:-5.0}
cd /tmp
run() { timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pass$PASS -- python3 $R/scripts/gpu_one_frame.py $DIST 3 > /dev/null 2>&1; PASS=$((PASS+1)); }
PASS=1
run SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
run SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_IFETCH SQ_BUSY_CU_CYCLES
run TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE
cd $R
for d in $OUT/pass*; do python scripts/pmc_summary.py $(ls $d/*/*counter_collection.csv) resolve_surface_fast resolve_sky raster_kernel setup_kernel; done > $OUT/summary.txt
