#!/bin/bash
# small graphs A/B on one box: parity tests, then intel / m3500 / dlr bench lines for a list of environments -> gpurun_out/$TAG/
# usage: scripts/gpu_small.sh TAG [notests] ; environments in $ENVS (";"-separated, "-" = default)
TAG=$1
mkdir -p gpurun_out/$TAG
if [ "$2" != "notests" ]; then timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; tail -3 gpurun_out/$TAG/pytest.log; fi
IFS=';' read -ra EV <<< "${ENVS:--;RR_PGO_LDS_FLOW=0}"
for W in ${WORKLOADS:-intel m3500 dlr}; do
  for E in "${EV[@]}"; do
    N=$(echo "$E" | tr ' =' '__')
    if [ "$E" = "-" ]; then E=""; fi
    env $E timeout -k 10 120 python bench.py --workload $W --precision ${PREC:-f64} --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/$TAG/bench_${W}_$N.json 2>gpurun_out/$TAG/bench_${W}_$N.err || { echo "FAILED $W $E"; tail -3 gpurun_out/$TAG/bench_${W}_$N.err; continue; }
    python3 -c "
import json; d=json.loads(open('gpurun_out/$TAG/bench_${W}_$N.json').read().strip().splitlines()[-1]); print('$W', '[$E]', round(d['value'],1), 'it/s', round(d['ms_per_step']*1000,1), 'us', {k: round(v,1) for k, v in d['roofline']['per_step_us_by_kernel_class'].items() if v}, d['errors'][-1], 'launches', d['launches_per_step'])"
  done
done
