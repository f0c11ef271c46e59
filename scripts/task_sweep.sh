#!/bin/bash
for w in intel input_M3500_g2o; do
  echo -n "auto "; python scripts/ab_bench.py rustrobotics_amd/librr_pgo.so $w | tail -1
  for t in 10 15 22 30 45 60 90 120 150 200; do
    echo -n "TASK_US=$t "; RR_PGO_TASK_US=$t python scripts/ab_bench.py rustrobotics_amd/librr_pgo.so $w | tail -1
  done
done
