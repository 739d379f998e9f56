#!/bin/bash
# GPU tests, then the benchmark's default run and its multi-rank code paths rehearsed with one rank (strips pipelined and
# serial, the orbit with the sharded prefilter's exchange).  Every step under its own time limit; stops at the first failure.
# usage: scripts/gpu_round3_check.sh OUTDIR
OUT=${1:-gpurun_out/check}
mkdir -p $OUT
PORT=$((29500 + RANDOM % 400))
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/tests.log 2>&1
rc=$?; tail -3 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err || exit 1
PBR_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 1 --mode strips --steps 20 --warmup 4 > $OUT/strips_dist1.json 2> $OUT/strips_dist1.err || exit 2
PBR_STRIPS_SERIAL=1 PBR_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((PORT+1)) bench.py --gpus 1 --mode strips --steps 20 --warmup 4 > $OUT/strips_dist1_serial.json 2> $OUT/strips_dist1_serial.err || exit 3
PBR_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((PORT+2)) bench.py --gpus 1 --steps 20 --warmup 20 --no-cpu-baseline > $OUT/orbit_dist1.json 2> $OUT/orbit_dist1.err || exit 4
echo all ok
