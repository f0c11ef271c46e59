#!/bin/bash
# One GPU-box round trip: parity tests, smoke, bench, rocprof summary -> gpurun_out/
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
