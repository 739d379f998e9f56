#!/bin/bash
# VALU counters of the IBL kernels (one rocprofv3 run per pass): the large prefilter, then the whole
# precompute at the reference's sizes.   usage: scripts/gpu_pmc_ibl.sh <out-prefix>
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
cd /tmp
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/prefilter -- python3 $R/scripts/gpu_prefilter_only.py > /dev/null 2>&1
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prefilter_trace -- python3 $R/scripts/gpu_prefilter_only.py > /dev/null 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/init -- python3 $R/scripts/gpu_one_frame.py 5.0 1 > /dev/null 2>&1
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/init_trace -- python3 $R/scripts/gpu_one_frame.py 5.0 1 > /dev/null 2>&1
cd $R
{
  echo "== prefilter 2048^2 x 10 mips x 1024 samples"; python scripts/pmc_summary.py $(ls $OUT/prefilter/*/*counter_collection.csv) prefilter
  grep -h prefilter $OUT/prefilter_trace/*/*kernel_stats.csv
  echo "== precompute at the reference's sizes"; python scripts/pmc_summary.py $(ls $OUT/init/*/*counter_collection.csv) prefilter irradiance brdf equirect mip
  grep -hE "prefilter|irradiance|brdf|equirect|mip" $OUT/init_trace/*/*kernel_stats.csv
} > $OUT/summary.txt
