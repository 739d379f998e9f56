#!/bin/bash
# The previous round's tree (build/r02tree, exported from git and built) against the working tree on the same box,
# twice each, alternating.
for i in 1 2; do
  for t in build/r02tree .; do
    out=$( cd $t && timeout -k 10 200 python bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-prefilter 2>/dev/null ) || exit 1
    echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', 'ms_per_step', round(d['ms_per_step']*1000,2), 'surface alone', round(d['roofline']['launch_ms']*1000,2), 'single frame', round(d['config']['single_frame_ms']*1000,2))"
  done
done
