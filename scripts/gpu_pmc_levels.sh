#!/bin/bash
# derived PMC metrics of the big-front kernels per level (= per number of fronts in the batch) on the 1M-edge lattice
# usage: scripts/gpu_pmc_levels.sh TAG "COUNTER1 COUNTER2 ..."   (one rocprofv3 pass per counter)
TAG=$1; shift
export TMPDIR=/tmp RR_PGO_NO_GRAPH=1
cd /tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_levels_$TAG.txt
: > $OUT
for C in $1; do
  rm -rf /tmp/pl
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pl -- python3 $R/scripts/gpu_grid_prof.py 400 250 1000000 f32 2 > /dev/null 2>&1
  echo "pass $C done"
  python3 - $(find /tmp/pl -name "*counter_collection.csv" | head -1) $C >> $OUT <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    n = r["Kernel_Name"].split("(")[0].split("rrpgo::")[-1]
    if not n.startswith("k_big") and not n.startswith("k_factor") : continue
    gx, gy, wg = int(r.get("Grid_Size_X", r.get("Grid_Size", 1))), int(r.get("Grid_Size_Y", 1)), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256)
    nf = gy if gy > 1 else gx // max(wg, 1)   # fronts of the level (2-D grids) or workgroups = tiles / tasks (1-D grids: k_big_update, k_big_schur, k_big_flow)
    k = (n[:26], nf)
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
print("==", sys.argv[2])
for (n, nf), (v, c) in sorted(acc.items()):
    print(f"  {n:26s} n={nf:6d} dispatches {c:5d} avg {v / c:12.2f}")
PY
done
cd $R; cat $OUT
