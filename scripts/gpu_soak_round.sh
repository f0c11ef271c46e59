#!/bin/bash
# r05 item 1: the 8-byte payload-as-flag probe (1e9 hand-offs per form), the bit-identity soak at 200 repeats per
# configuration, and the full GPU suite three times over with complete logs.  usage: scripts/gpu_soak_round.sh TAG
TAG=${1:-r05b}
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/handoff_probe.hip -o /tmp/handoff_probe || exit 1
timeout -k 10 300 /tmp/handoff_probe 16300 > gpurun_out/handoff_probe8_$TAG.txt 2>&1; cat gpurun_out/handoff_probe8_$TAG.txt
TMO=900 bash scripts/gpu_soak.sh $TAG 200 || exit 1
for k in 1 2 3; do
  timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_full_${TAG}_$k.log 2>&1
  tail -1 gpurun_out/pytest_gpu_full_${TAG}_$k.log
done
