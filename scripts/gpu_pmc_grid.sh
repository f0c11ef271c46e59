#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the lattice's kernels under an environment setting: usage gpu_pmc_grid.sh TAG [VAR=val ...]
TAG=$1; shift
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_grid_$C
  env "$@" RR_PGO_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_grid_$C -- python3 $R/scripts/gpu_grid_prof.py 400 250 1000000 f32 3 > /dev/null 2>&1
  python3 $R/scripts/pmc_summary.py $(find /tmp/pmc_grid_$C -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_${TAG}_grid_$C.txt
  head -4 $R/gpurun_out/pmc_${TAG}_grid_$C.txt | cut -c1-170
done
