#!/bin/bash
# copy the artefacts of an evidence round from gpurun_out/ into profiles/<TAG>_*   usage: scripts/collect_round.sh TAG
TAG=${1:-r06z}
cd "$(dirname "$0")/.."
for f in gpurun_out/*_$TAG.json gpurun_out/*_$TAG.csv gpurun_out/*_$TAG.txt gpurun_out/*_$TAG.log; do
  [ -f "$f" ] || continue
  b=$(basename "$f"); b=${b%_$TAG.*}; ext=${f##*.}
  case "$b" in timeline|pytest_gpu_full|round|pmc|pmc_levels) [ "$ext" = log -o "$b" = timeline ] && continue;; esac
  cp "$f" profiles/${TAG}_$b.$ext
done
for f in gpurun_out/pmc_${TAG}_*_SIZE.txt; do [ -f "$f" ] && cp "$f" profiles/$(basename "$f" | sed "s/^pmc_//"); done
[ -f gpurun_out/pmc_levels_$TAG.txt ] && cp gpurun_out/pmc_levels_$TAG.txt profiles/${TAG}_pmc_derived_by_kernel.txt
ls profiles | grep "^$TAG" | wc -l
