#!/bin/bash
# HBM-side traffic of the frame kernels from the L2's memory-side request counters, one rocprofv3
# run per pass (reads, then writes), over the benchmark's own frames.
# (--passes-only: every launch of the frame kernels in the run is then a whole pass of the default number of views)
# usage: scripts/gpu_traffic.sh <out-prefix>
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/rd -- python3 $R/bench.py --no-cpu-baseline --no-prefilter --steps 20 --warmup 4 --passes-only > $OUT.rd.json 2> $OUT.rd.err
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/wr -- python3 $R/bench.py --no-cpu-baseline --no-prefilter --steps 20 --warmup 4 --passes-only > $OUT.wr.json 2> $OUT.wr.err
cd $R
python3 scripts/traffic_summary.py $(ls $OUT/rd/*/*counter_collection.csv) $(ls $OUT/wr/*/*counter_collection.csv) > $OUT.traffic.json
cat $OUT.traffic.json
