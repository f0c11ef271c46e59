#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for w in sphere2500 torus3D parking-garage; do
  for e in 1 0; do
    if [ $e = 1 ]; then export RR_PGO_SEPARATE_DIAG32=1; else unset RR_PGO_SEPARATE_DIAG32; fi
    echo -n "SEPARATE=$e "; python scripts/ab_bench.py rustrobotics_amd/librr_pgo.so $w | tail -1
  done
done
for e in 1 0; do
  if [ $e = 1 ]; then export RR_PGO_SEPARATE_DIAG32=1; else unset RR_PGO_SEPARATE_DIAG32; fi
  echo -n "SEPARATE=$e "; python scripts/ab_grid.py rustrobotics_amd/librr_pgo.so 400 250 1000000 f32 | tail -1
done
