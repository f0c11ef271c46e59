#!/bin/bash
# A/B of library builds (librr_pgo_<tag>.so) on the lattice with every level of at most 32 fronts in the flow kernel
for L in "$@"; do
  for TASKS in 16384 100000000; do
    RR_PGO_FLOW_TASKS=$TASKS timeout -k 10 200 python3 scripts/ab_grid.py rustrobotics_amd/librr_pgo_$L.so 400 250 1000000 f32 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$L tasks<=$TASKS: /"
  done
done
