#!/bin/bash
# first GPU contact of k_big_flow: bit-identity against the launch sequence, then the lattice / sphere2500 step with and without it
# usage: scripts/gpu_flow_check.sh TAG
TAG=${1:-flow}
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "flow_launch" > gpurun_out/pytest_flow_$TAG.log 2>&1
rc=$?
tail -15 gpurun_out/pytest_flow_$TAG.log
[ $rc -ne 0 ] && exit $rc
for F in 32 0; do
  for W in grid:400x250:1000000:f32 sphere2500:f64; do
    WL=${W%:*}; PR=${W##*:}
    RR_PGO_FLOW=$F timeout -k 10 200 python bench.py --workload $WL --precision $PR --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/bench_${WL%%:*}_${TAG}_flow$F.json 2>gpurun_out/bench_${WL%%:*}_${TAG}_flow$F.err || { tail -5 gpurun_out/bench_${WL%%:*}_${TAG}_flow$F.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/bench_${WL%%:*}_${TAG}_flow$F.json')); print('flow=$F', '$WL', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms', {k: round(v) for k, v in d['roofline']['per_step_us_by_kernel_class'].items() if v}, d['errors'][-1])"
  done
done
