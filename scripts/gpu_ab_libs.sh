#!/bin/bash
# A/B of library builds rustrobotics_amd/librr_pgo_<tag>.so on the lattice (fp32): usage gpu_ab_libs.sh tag ...
for L in "$@"; do
  timeout -k 10 200 python3 scripts/ab_grid.py rustrobotics_amd/librr_pgo_$L.so 400 250 1000000 f32 2>&1 | grep -v amdgpu.ids | tail -2 | sed "s/^/$L: /"
done
