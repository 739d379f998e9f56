#!/bin/bash
# The orbit benchmark through its multi-rank code path with one rank (process group, barriers, reductions), at the
# default number of steps, next to the plain one-rank run: what RCCL's presence costs a rank.
# usage: bash scripts/gpu_dist_rehearsal.sh [extra bench.py arguments]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
PORT=$((29500 + RANDOM % 400))
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_step', round(d['ms_per_step']*1000,2), 'us')"; }
timeout -k 10 200 python bench.py --no-cpu-baseline --no-prefilter "$@" 2> /dev/null | show plain || exit 1
PBR_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 1 --no-cpu-baseline --no-prefilter "$@" 2> /dev/null | show dist || exit 2
timeout -k 10 200 python bench.py --no-cpu-baseline --no-prefilter "$@" 2> /dev/null | show plain || exit 1
PBR_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((PORT+1)) bench.py --gpus 1 --no-cpu-baseline --no-prefilter "$@" 2> /dev/null | show dist || exit 2
