#!/bin/bash
# Runs bench.py against several builds of the library (scripts/build_variant.sh) and settings, one line each:
#   bash scripts/gpu_bench_variants.sh "name:ENV=VALUE,ENV2=VALUE:library" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; lib=${rest#*:}
  envline=$(echo "$envs" | tr ',' ' ')
  if ! out=$(env $envline PBR355_LIBRARY=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-prefilter 2> gpurun_out/variant_$name.err); then echo "$name FAILED"; exit 1; fi
  echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', 'ms_per_step', round(d['ms_per_step']*1000,2), 'us; surface alone', round(d['roofline']['launch_ms']*1000,2), 'us; single frame', round(d['config']['single_frame_ms']*1000,2))"
done
