#!/bin/bash
# quick GPU round: parity tests, then the lattice (fp32) and sphere2500 / intel bench lines and the lattice timeline -> gpurun_out/
# usage: scripts/gpu_quick.sh TAG [notests]
TAG=$1
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1; fi
for W in grid:400x250:1000000:f32 sphere2500:f64 intel:f64 m3500:f64; do
  WL=${W%:*}; PR=${W##*:}
  python bench.py --workload $WL --precision $PR --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/bench_${WL%%:*}_$TAG.json 2>/dev/null || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_${WL%%:*}_$TAG.json')); print('$WL', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms', {k: round(v) for k, v in d['roofline']['per_step_us_by_kernel_class'].items() if v}, d['errors'][-1])"
done
bash scripts/gpu_timeline.sh $TAG 400 250 1000000 f32 && python3 scripts/timeline_levels.py gpurun_out/timeline_$TAG.json
