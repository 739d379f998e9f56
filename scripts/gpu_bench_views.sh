#!/bin/bash
# Frame time of the orbit for combinations of passes in flight and views per pass: [BENCH_ARGS="--steps 20 --warmup 5"] scripts/gpu_bench_views.sh "K:V K:V ..."
set -e
mkdir -p gpurun_out
list=${1:-"4:1 2:2 2:4 3:4 4:2 1:4 3:2"}
for kv in $list; do
  k=${kv%%:*}; v=${kv##*:}
  timeout -k 10 120 python bench.py --in-flight $k --views $v --no-cpu-baseline --no-prefilter $BENCH_ARGS > gpurun_out/bench_views_${k}_${v}.json 2> gpurun_out/bench_views_err.log
  python - $k $v <<'PY'
import json, sys
k, v = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/bench_views_{k}_{v}.json").read().strip().splitlines()[-1])
print(f"in-flight {k} views {v}: {d['ms_per_step']*1e3:.1f} us per frame, surface launch alone {d['roofline']['launch_ms']*1e3:.1f} us, single frame {d['config']['single_frame_ms']*1e3:.1f} us", flush=True)
PY
done
