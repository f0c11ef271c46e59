#!/bin/bash
# r05 evidence round: tests, smoke, the default bench line, bench lines + rocprofv3 kernel stats of every BASELINE config, the
# critical paths of the dataflow factorisation (stamps build), the constructor closure, the lattice's per-level timeline
# -> gpurun_out/ (copied into profiles/ as <TAG>_*).  usage: scripts/gpu_round_r05.sh TAG
TAG=${1:-r05z}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_full_$TAG.log 2>&1; tail -3 gpurun_out/pytest_gpu_full_$TAG.log | tee gpurun_out/pytest_gpu_$TAG.log   # (the full log stays: a failure keeps its message)
python __graft_entry__.py smoke 2>&1 | tail -1 | tee gpurun_out/smoke_$TAG.log
echo "== default bench line"; date
timeout -k 10 900 python bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err || echo "default bench FAILED"
date
for W in intel:f64 m3500:f64 dlr:f64 sphere2500:f64 intel:mixed; do
  WL=${W%:*}; PR=${W##*:}
  timeout -k 10 300 python bench.py --workload $WL --precision $PR --no-cpu-baseline --no-secondary > gpurun_out/bench_${WL}_${PR}_$TAG.json 2>/dev/null
  rm -rf /tmp/prof_$WL
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$WL -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --precision $PR --no-cpu-baseline --no-secondary > /tmp/prof_$WL.log 2>&1 )
  cp $(find /tmp/prof_$WL -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_${WL}_${PR}_$TAG.csv
  echo "$WL $PR done"
done
# the cross-level launch of the small graphs with fronts beyond LDS, A/B against one launch per level
for W in sphere2500 torus3D; do timeout -k 10 200 python scripts/gpu_env_ab.py $W f64 RR_PGO_FLOW_XL=0 - ; done > gpurun_out/xl_ab_$TAG.txt 2>&1
for N in intel input_M3500_g2o dlr sphere2500; do RR_PGO_ANALYZE_TIMES=1 timeout -k 10 120 python scripts/time_closure.py $N > gpurun_out/closure_${N}_$TAG.txt 2>&1; done
for N in intel input_M3500_g2o dlr sphere2500; do RR_PGO_ANALYSIS_CACHE=0 RR_PGO_ANALYZE_TIMES=1 timeout -k 10 120 python scripts/time_closure.py $N > gpurun_out/closure_uncached_${N}_$TAG.txt 2>&1; done
# the dissection of the small graphs: level sets / coordinate cuts only (RR_PGO_ML_ND=0, r01 - r04) against the multilevel bisection
( echo "# GN it/s, fp64: RR_PGO_ML_ND=0 = the level-set / coordinate-cut dissection of r01 - r04, '-' = the default (multilevel bisection with a minimum-cover separator); scripts/gpu_env_ab.py"
  for W in intel input_M3500_g2o dlr parking-garage sphere2500 torus3D; do timeout -k 10 200 python scripts/gpu_env_ab.py $W f64 RR_PGO_ML_ND=0 - 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/ml_nd_ab_$TAG.txt
timeout -k 10 300 python bench.py --workload grid:400x250:1000000 --precision f32 --steps 50 --warmup 5 --no-secondary > gpurun_out/bench_grid_f32_$TAG.json 2> gpurun_out/bench_grid_$TAG.err
rm -rf /tmp/prof_grid
( cd /tmp && RR_PGO_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_grid -- python3 $GRAFT_REPO_ROOT/scripts/gpu_grid_prof.py 400 250 1000000 f32 5 > /tmp/prof_grid.log 2>&1 )
cp $(find /tmp/prof_grid -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_grid_f32_$TAG.csv
bash scripts/gpu_timeline.sh $TAG 400 250 1000000 f32 && python3 scripts/timeline_levels.py gpurun_out/timeline_$TAG.json > gpurun_out/lattice_levels_$TAG.txt
cat gpurun_out/lattice_levels_$TAG.txt
for f in gpurun_out/bench_*_$TAG.json; do echo "== $f"; python3 -c "
import json,sys
d=json.load(open('$f'))
r=d.get('roofline') or {}
print(d['value'], d['unit'], '| ms/step', round(d['ms_per_step'],4), '| roofline', r.get('kernel'), r.get('bound'), round(r.get('achieved',0),2), r.get('unit'), 'frac', round(r.get('frac',0),4), '| cpu', (d.get('cpu_baseline') or {}).get('value'), '| closure', d.get('closure_ms'), (d.get('cpu_baseline') or {}).get('closure_ms'))
for s in d.get('secondary', []):
    print('   secondary', s.get('workload', '?')[:40], s.get('dtype'), s.get('parallelism'), round(s.get('value', 0), 1), round(s.get('ms_per_step', 0), 3), s.get('error', ''))
"; done
