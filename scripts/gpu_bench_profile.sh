#!/bin/bash
# bench line + rocprofv3 kernel-trace stats of the same command -> gpurun_out/
TAG=${1:-r01}
shift
mkdir -p gpurun_out
python bench.py "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -c 3000 gpurun_out/bench_$TAG.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/prof_$TAG.log 2>&1
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_$TAG.csv
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +50M -delete
head -20 gpurun_out/kernel_stats_$TAG.csv
