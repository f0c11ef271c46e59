#!/bin/bash
# timeline of the last iteration of the lattice with the Schur passes beside the flow launches: per level the start / end of
# k_big_flow and of k_big_schur_flow (do they overlap?).  usage: scripts/gpu_overlap_timeline.sh TAG NF
TAG=$1; NF=${2:-32}
export RR_PGO_SCHUR_OVERLAP=$NF
bash scripts/gpu_timeline.sh $TAG 400 250 1000000 f32
python3 - gpurun_out/timeline_$TAG.json <<'PY'
import json, sys
t = json.load(open(sys.argv[1]))
lvl = -1
for name, start, dur, gx, gy, gz, wg in t:
    short = name.split('<')[0]
    if short == 'k_big_build': lvl += 1; print()
    if short.startswith('k_big') and 'gemv' not in short and 'solve' not in short:
        print("level %2d  %-18s start %8.1f  end %8.1f  dur %7.1f us  grid %d" % (lvl, short, start / 1e3, (start + dur) / 1e3, dur / 1e3, gx // max(wg, 1)))
PY
