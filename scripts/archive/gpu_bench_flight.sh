#!/bin/bash
# Frames in flight against hardware queues per process: bash scripts/gpu_bench_flight.sh "queues:in_flight" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for spec in "$@"; do
  q=${spec%%:*}; k=${spec#*:}
  if ! out=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --no-cpu-baseline --no-prefilter --in-flight $k 2> gpurun_out/flight_${q}_$k.err); then echo "$spec FAILED"; exit 1; fi
  echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues $q in flight $k: ms_per_step', round(d['ms_per_step']*1000,2), 'us')"
done
