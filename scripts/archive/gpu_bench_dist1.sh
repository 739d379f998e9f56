#!/bin/bash
# The benchmark's multi-rank code path (process group, barriers, RCCL) rehearsed with one rank on a one-GPU box.
# usage: scripts/gpu_bench_dist1.sh [bench args...]
PORT=$((29500 + RANDOM % 400))
PBR_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 1 --no-prefilter --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('dist ms_per_step %.5f single_frame_ms %.5f' % (d['ms_per_step'], d['config']['single_frame_ms']))"
