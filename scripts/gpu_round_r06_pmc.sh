#!/bin/bash
# r06 evidence round, part B: the counter passes.  HBM traffic of every BASELINE config (scripts/gpu_pmc.sh: FETCH_SIZE and
# WRITE_SIZE in separate passes), then one pass per derived counter on the lattice's big-front kernels per level
# (scripts/gpu_pmc_levels.sh).   usage: scripts/gpu_round_r06_pmc.sh TAG
TAG=${1:-r06z}
mkdir -p gpurun_out
bash scripts/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1; tail -3 gpurun_out/pmc_$TAG.log
bash scripts/gpu_pmc_levels.sh $TAG "MfmaUtil MemUnitStalled LDSBankConflict OccupancyPercent" > gpurun_out/pmc_levels_$TAG.log 2>&1; tail -5 gpurun_out/pmc_levels_$TAG.log
