#!/bin/bash
# One GPU-box round trip: parity tests, smoke, bench (intel + 1M-edge lattice), rocprof summaries -> gpurun_out/
TAG=${1:-r01}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/pytest_gpu_$TAG.log
python __graft_entry__.py smoke 2>&1 | tail -1 | tee gpurun_out/smoke_$TAG.log
export TMPDIR=/tmp
# --- intel.g2o fp64 (BASELINE configs[1])
python bench.py > gpurun_out/bench_intel_$TAG.json 2> gpurun_out/bench_intel_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_intel_$TAG -- python3 bench.py --no-cpu-baseline --no-secondary > gpurun_out/prof_intel_$TAG.log 2>&1
cp $(find gpurun_out/prof_intel_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_intel_$TAG.csv
# --- M3500 + sphere2500 bench lines (no profile)
python bench.py --workload m3500 --no-cpu-baseline --no-secondary > gpurun_out/bench_m3500_$TAG.json 2>/dev/null
python bench.py --workload sphere2500 --no-cpu-baseline --no-secondary > gpurun_out/bench_sphere2500_$TAG.json 2>/dev/null
python bench.py --workload dlr --no-cpu-baseline --no-secondary > gpurun_out/bench_dlr_$TAG.json 2>/dev/null
python bench.py --precision mixed --no-cpu-baseline --no-secondary > gpurun_out/bench_intelmixed_$TAG.json 2>/dev/null
# --- 1M-edge lattice fp32 (BASELINE configs[3]); rocprof needs plain launches (graph replay of 800 nodes crashes it)
python bench.py --workload grid:400x250:1000000 --precision f32 --steps 50 --warmup 5 > gpurun_out/bench_grid_$TAG.json 2> gpurun_out/bench_grid_$TAG.err
RR_PGO_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_grid_$TAG -- python3 scripts/gpu_grid_prof.py 400 250 1000000 f32 5 > gpurun_out/prof_grid_$TAG.log 2>&1
cp $(find gpurun_out/prof_grid_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_grid_$TAG.csv
find gpurun_out -name "*kernel_trace.csv" -delete
for f in gpurun_out/bench_*_$TAG.json; do echo "== $f"; python3 -c "
import json,sys
d=json.load(open('$f'))
r=d.get('roofline') or {}
print(d['value'], d['unit'], '| ms/step', round(d['ms_per_step'],3), '| roofline', r.get('kernel'), r.get('bound'), round(r.get('achieved',0),2), r.get('unit'), 'frac', round(r.get('frac',0),4), '| cpu', (d.get('cpu_baseline') or {}).get('value'))
"; done
head -6 gpurun_out/kernel_stats_intel_$TAG.csv | cut -c1-160
head -12 gpurun_out/kernel_stats_grid_$TAG.csv | cut -c1-160
