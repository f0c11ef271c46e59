#!/bin/bash
# lattice (fp32) and sphere2500 (fp64) step time against RR_PGO_FLOW (levels of at most that many big fronts run as k_big_flow)
mkdir -p gpurun_out
for F in ${FLOWS:-0 1 2 4 8 16 32}; do
  for W in grid:400x250:1000000:f32 sphere2500:f64; do
    WL=${W%:*}; PR=${W##*:}
    RR_PGO_FLOW=$F timeout -k 10 200 python bench.py --workload $WL --precision $PR --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/sweep.json 2>gpurun_out/sweep.err || { tail -5 gpurun_out/sweep.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/sweep.json')); print('flow=$F', '$WL', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms', {k: round(v) for k, v in d['roofline']['per_step_us_by_kernel_class'].items() if v and k.startswith('big')})"
  done
done
