#!/bin/bash
# A/B of environment knobs on one workload: usage scripts/gpu_env_ab.sh WORKLOAD PRECISION "ENV1=.. ENV2=.." "ENV.." ...
WL=$1; PR=$2; shift 2
for E in "$@"; do
  env $E python bench.py --workload $WL --precision $PR --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > /tmp/ab.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('[$E]', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms', {k: round(v) for k, v in d['roofline']['per_step_us_by_kernel_class'].items() if v})"
done
