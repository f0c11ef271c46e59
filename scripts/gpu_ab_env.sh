#!/bin/bash
# A/B of an environment switch on the lattice (fp32) and sphere2500 (or the workload:precision list in $WLS):
# usage gpu_ab_env.sh "VAR=a" "VAR=b" ...
for E in "$@"; do
  for W in ${WLS:-grid:400x250:1000000:f32 sphere2500:f64}; do
    WL=${W%:*}; PR=${W##*:}
    env $E timeout -k 10 200 python bench.py --workload $WL --precision $PR --steps 40 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/ab.json 2>gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$E', '$WL', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms', {k: round(v) for k, v in d['roofline']['per_step_us_by_kernel_class'].items() if v and k.startswith('big')}, d['errors'][-1])"
  done
done
