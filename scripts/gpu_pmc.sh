#!/bin/bash
# HBM traffic counters (separate --pmc passes, kernel-trace only) for intel fp64 and the 1M-edge lattice fp32.
TAG=${1:-r01}
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_intel_$C -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 2 > /dev/null 2>&1
  echo "pmc $C intel done"
  python3 $R/scripts/pmc_summary.py $(find $R/gpurun_out/pmc_intel_$C -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_${TAG}_intel_$C.txt
  RR_PGO_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_grid_$C -- python3 $R/scripts/gpu_grid_prof.py 400 250 1000000 f32 3 > /dev/null 2>&1
  echo "pmc $C grid done"
  python3 $R/scripts/pmc_summary.py $(find $R/gpurun_out/pmc_grid_$C -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_${TAG}_grid_$C.txt
  rm -rf $R/gpurun_out/pmc_intel_$C $R/gpurun_out/pmc_grid_$C
done
cd $R
head -8 gpurun_out/pmc_${TAG}_*.txt | cut -c1-170
