#!/bin/bash
# per-launch timeline of the LAST iteration of a g2o dataset (plain launches) -> gpurun_out/timeline_<tag>.json
# usage: scripts/gpu_timeline_g2o.sh TAG NAME [PRECISION]
TAG=$1; NAME=$2; PREC=${3:-f64}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$TAG
RR_PGO_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -- python3 $GRAFT_REPO_ROOT/scripts/prof_g2o.py $NAME 3 > /tmp/tl_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
F=$(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$F" gpurun_out/timeline_$TAG.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "k_linearize" in n)
out = []
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    n = r["Kernel_Name"]
    short = n.split("(")[0].split("rrpgo::")[-1]
    out.append([short, int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"])])
json.dump(out, open(sys.argv[2], "w"))
print("launches", len(out), "span us", (int(rows[-1]["End_Timestamp"]) - t0) / 1e3)
PY
