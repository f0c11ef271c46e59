#!/bin/bash
# every candidate elimination tree of a graph with fronts beyond LDS, pinned one at a time (RR_PGO_ND_LEAF + RR_PGO_AMALG_NP +
# RR_PGO_JOIN_SEPARATORS), measured beside the product's own pick ("-"): does the cost model pick the measured best there too?
# usage: scripts/gpu_tree_sweep.sh TAG WORKLOAD...
TAG=$1; shift
for W in "$@"; do
  CFGS="-"
  for L in 250 150 100 70 50; do for NP in 16 32 72; do for J in 1 0; do CFGS="$CFGS RR_PGO_ND_LEAF=$L,RR_PGO_AMALG_NP=$NP,RR_PGO_JOIN_SEPARATORS=$J"; done; done; done
  timeout -k 10 500 python scripts/gpu_env_ab.py $W f64 $CFGS 2>&1 | grep -v amdgpu.ids
done > gpurun_out/tree_sweep_$TAG.txt
sort -k6 -n -r gpurun_out/tree_sweep_$TAG.txt | cut -c1-110 | head -40
