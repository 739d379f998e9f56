#!/bin/bash
# Kernel durations of one context drawing passes alone (no other pass in flight): bash scripts/gpu_profile_pass.sh <tag> [views]
# Leaves gpurun_out/<tag>_pass_kernel_stats.csv.
set -e
TAG=${1:-pass}
V=${2:-4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --in-flight 1 --views $V --no-prefilter --no-cpu-baseline > $R/gpurun_out/${TAG}_pass.json 2> $R/gpurun_out/${TAG}_pass.err
STATS=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp "$STATS" $R/gpurun_out/${TAG}_pass_kernel_stats.csv
head -8 $R/gpurun_out/${TAG}_pass_kernel_stats.csv
