#!/bin/bash
# r06 evidence round, part A: tests, smoke, the default bench line (as the driver runs it AND with the default step count),
# bench lines + rocprofv3 kernel stats of every BASELINE config, the kernel timeline of optimize() calls, the constructor
# closure cached / uncached, the lattice's per-level timeline, the A/B of the device-side loop against the host loop, the
# two-rank rehearsal of bench.py's N > 1 plan.   -> gpurun_out/ (copied into profiles/ as <TAG>_*).   usage: scripts/gpu_round_r06.sh TAG
TAG=${1:-r06z}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_full_$TAG.log 2>&1; tail -3 gpurun_out/pytest_gpu_full_$TAG.log | tee gpurun_out/pytest_gpu_$TAG.log
python __graft_entry__.py smoke 2>&1 | tail -1 | tee gpurun_out/smoke_$TAG.log
echo "== default bench line"; date
timeout -k 10 900 python bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err || echo "default bench FAILED"
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_$TAG.json 2> gpurun_out/bench_driver_$TAG.err || echo "driver-shaped bench FAILED"
date
for W in intel:f64 m3500:f64 dlr:f64 sphere2500:f64 intel:mixed; do
  WL=${W%:*}; PR=${W##*:}
  timeout -k 10 300 python bench.py --workload $WL --precision $PR --no-cpu-baseline --no-secondary > gpurun_out/bench_${WL}_${PR}_$TAG.json 2>/dev/null
  rm -rf /tmp/prof_$WL
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$WL -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --precision $PR --no-cpu-baseline --no-secondary > /tmp/prof_$WL.log 2>&1 )
  cp $(find /tmp/prof_$WL -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_${WL}_${PR}_$TAG.csv
  python3 scripts/kernel_stats_real.py $(find /tmp/prof_$WL -name "*kernel_trace.csv" | head -1) > gpurun_out/kernel_working_launches_${WL}_${PR}_$TAG.txt
  echo "$WL $PR done"
done
# the kernel timeline of optimize(10) calls from the initial state (what the headline times)
rm -rf /tmp/opt_trace
( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/opt_trace -- python3 $GRAFT_REPO_ROOT/scripts/gpu_opt_trace.py intel 6 > /dev/null 2>&1 )
python3 scripts/gpu_opt_trace.py analyse $(find /tmp/opt_trace -name "*kernel_trace.csv" | head -1) > gpurun_out/optimize_timeline_intel_$TAG.txt
# the device-side loop against the loop with one host round trip per iteration (RR_PGO_SYNC_OPTIMIZE=1), through optimize()
( echo "# GN it/s through rr_pgo_optimize (bench.py, K = 200 iterations of optimize(10) calls from the initial state): the loop on the device (default) / one host round trip per iteration"
  for W in intel m3500 dlr sphere2500; do
    for E in "" "RR_PGO_SYNC_OPTIMIZE=1"; do
      echo -n "$W ${E:-default}: "; env $E timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s', round(d['ms_per_step']*1e3,2), 'us/step; optimize(10):', round(d['optimize10_ms'],3), 'ms for', len(d['errors'])-1, 'iterations; iterate_async', round(d['iterate_async']['ms_per_step']*1e3,2), 'us/step')"
    done
  done ) > gpurun_out/optimize_loop_ab_$TAG.txt 2>&1
cat gpurun_out/optimize_loop_ab_$TAG.txt
for N in intel input_M3500_g2o dlr sphere2500; do RR_PGO_ANALYZE_TIMES=1 timeout -k 10 120 python scripts/time_closure.py $N > gpurun_out/closure_${N}_$TAG.txt 2>&1; done
for N in intel input_M3500_g2o dlr sphere2500; do RR_PGO_ANALYSIS_CACHE=0 RR_PGO_ANALYZE_TIMES=1 timeout -k 10 120 python scripts/time_closure.py $N > gpurun_out/closure_uncached_${N}_$TAG.txt 2>&1; done
timeout -k 10 300 python bench.py --workload grid:400x250:1000000 --precision f32 --steps 50 --warmup 5 --no-secondary > gpurun_out/bench_grid_f32_$TAG.json 2> gpurun_out/bench_grid_$TAG.err
timeout -k 10 300 python bench.py --workload grid:400x250:1000000 --precision mixed --steps 50 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/bench_grid_mixed_$TAG.json 2>> gpurun_out/bench_grid_$TAG.err
rm -rf /tmp/prof_grid
( cd /tmp && RR_PGO_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_grid -- python3 $GRAFT_REPO_ROOT/scripts/gpu_grid_prof.py 400 250 1000000 f32 5 > /tmp/prof_grid.log 2>&1 )
cp $(find /tmp/prof_grid -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_grid_f32_$TAG.csv
bash scripts/gpu_timeline.sh $TAG 400 250 1000000 f32 && python3 scripts/timeline_levels.py gpurun_out/timeline_$TAG.json > gpurun_out/lattice_levels_$TAG.txt
cat gpurun_out/lattice_levels_$TAG.txt
# bench.py's N > 1 plan with two ranks on this one GPU (gloo, collectives staged through host memory): a rehearsal, not a scaling number
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29433 bench.py --gpus 2 --steps 20 --warmup 5 --collectives host-staged > gpurun_out/bench_2ranks_host_staged_$TAG.json 2> gpurun_out/bench_2ranks_host_staged_$TAG.err; echo "two-rank rehearsal rc=$?"
for f in gpurun_out/bench_*_$TAG.json; do echo "== $f"; python3 -c "
import json,sys
d=json.load(open('$f'))
r=d.get('roofline') or {}
print(d['value'], d['unit'], '| ms/step', round(d['ms_per_step'],4), '| roofline', r.get('kernel'), r.get('bound'), round(r.get('achieved',0),2), r.get('unit'), 'frac', round(r.get('frac',0),4), 'traffic', r.get('traffic'), '| cpu', (d.get('cpu_baseline') or {}).get('value'), '| closure cached/uncached', d.get('closure_cached_ms'), d.get('closure_uncached_ms'), (d.get('cpu_baseline') or {}).get('closure_ms'))
for s in d.get('secondary', []):
    print('   secondary', s.get('workload', '?')[:40], s.get('dtype'), s.get('parallelism'), round(s.get('value', 0), 1), round(s.get('ms_per_step', 0), 3), 'cpu', (s.get('cpu_baseline') or {}).get('value'), 'traffic', (s.get('roofline') or {}).get('traffic'), s.get('error', ''))
"; done
