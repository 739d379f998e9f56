#!/bin/bash
# Runs the frame benchmark several times and prints ms_per_step / single_frame_ms of each run (run-to-run spread is a few per cent).
# usage: scripts/gpu_bench_repeat.sh <n> [bench args...]
N=${1:-3}; shift
for i in $(seq 1 $N); do
  timeout -k 10 300 python bench.py --no-prefilter --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step %.5f single_frame_ms %.5f surface_ms %.5f' % (d['ms_per_step'], d['config']['single_frame_ms'], d['kernel_ms']['resolve_surface_fast_kernel']))" || exit 1
done
