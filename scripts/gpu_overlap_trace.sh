#!/bin/bash
# kernel trace of the lattice under an environment setting; prints concurrency figures (scripts/trace_overlap.py) and
# the launch timeline of one top level.  usage: scripts/gpu_overlap_trace.sh "ENV=.. ENV=.."
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ov
export RR_PGO_NO_GRAPH=1
for kv in $1; do export $kv; done
rocprofv3 --kernel-trace --output-format csv -d /tmp/ov -- python3 $GRAFT_REPO_ROOT/scripts/gpu_grid_prof.py 400 250 1000000 f32 3 > /tmp/ov.log 2>&1
F=$(find /tmp/ov -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/trace_overlap.py $F
python3 - $F <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_linearize" in r["Kernel_Name"])
rows = rows[last:]
t0 = int(rows[0]["Start_Timestamp"])
# the level with 8 fronts: print its launches
sel = [r for r in rows if r["Grid_Size_Z"] == "8" or r["Grid_Size_Y"] == "8"]
for r in sel[:70]:
    n = r["Kernel_Name"].split("(")[0].split("rrpgo::")[-1][:28]
    print(f"{n:28s} q={r.get('Queue_Id','?'):>3s} grid=({int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}")
PY
