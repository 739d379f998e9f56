#!/bin/bash
# Kernel-trace summary of the default bench command: bash scripts/gpu_profile_bench.sh <tag>
# Leaves gpurun_out/<tag>_bench.json (the bench line under the profiler) and gpurun_out/<tag>_bench_kernel_stats.csv.
set -e
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
STATS=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp "$STATS" $R/gpurun_out/${TAG}_bench_kernel_stats.csv
head -12 $R/gpurun_out/${TAG}_bench_kernel_stats.csv
