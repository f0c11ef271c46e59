#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for w in sphere2500 torus3D; do
  for l in librr_pgo_base.so librr_pgo.so; do python scripts/ab_bench.py rustrobotics_amd/$l $w | tail -1; done
done
for l in librr_pgo_base.so librr_pgo.so; do python scripts/ab_grid.py rustrobotics_amd/$l 400 250 1000000 f32 | tail -1; done
