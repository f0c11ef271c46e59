#!/bin/bash
# HBM traffic counters (separate --pmc passes, kernel-trace only) of every BASELINE config: intel / M3500 / sphere2500 fp64
# (scripts/prof_g2o.py: plain launches of whole iterations) and the 1M-edge lattice fp32 (scripts/gpu_grid_prof.py).
# -> gpurun_out/pmc_<TAG>_<workload>_<COUNTER>.txt (scripts/pmc_summary.py), copied into profiles/<TAG>_<workload>_<COUNTER>.txt;
# scripts/pmc_traffic_json.py TAG turns them into profiles/pmc_traffic.json.       usage: scripts/gpu_pmc.sh TAG
TAG=${1:-r06z}
export TMPDIR=/tmp RR_PGO_NO_GRAPH=1
cd /tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  for W in intel input_M3500_g2o sphere2500; do
    S=${W/input_M3500_g2o/m3500}
    rm -rf /tmp/pmc_$S
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$S -- python3 $R/scripts/prof_g2o.py $W 8 > /dev/null 2>&1
    python3 $R/scripts/pmc_summary.py $(find /tmp/pmc_$S -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_${TAG}_${S}_$C.txt
    echo "pmc $C $S done"
  done
  rm -rf /tmp/pmc_grid
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_grid -- python3 $R/scripts/gpu_grid_prof.py 400 250 1000000 f32 3 > /dev/null 2>&1
  python3 $R/scripts/pmc_summary.py $(find /tmp/pmc_grid -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_${TAG}_grid_$C.txt
  echo "pmc $C grid done"
done
cd $R
head -8 gpurun_out/pmc_${TAG}_*.txt | cut -c1-170
