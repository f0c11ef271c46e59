#!/bin/bash
# Pull form vs edge-parallel form of the linearisation on the 1M-edge lattice (fp32): kernel times (rocprofv3
# --kernel-trace --stats) and HBM-side traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes) -> gpurun_out/lin_forms_<tag>.txt
TAG=${1:-r02}
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lin_forms_$TAG.txt
: > $OUT
for FORM in pull edges; do
  if [ $FORM = edges ]; then export RR_PGO_EDGE_LINEARIZE=1; else unset RR_PGO_EDGE_LINEARIZE; fi
  export RR_PGO_NO_GRAPH=1
  echo "=== $FORM form: kernel stats (5 iterations)" >> $OUT
  rm -rf /tmp/lf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lf -- python3 $R/scripts/gpu_grid_prof.py 400 250 1000000 f32 5 > /dev/null 2>&1
  python3 -c "
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_lin' in r['Name']: print(r['Name'].split('(')[0][-44:], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1))
" $(find /tmp/lf -name "*kernel_stats.csv" | head -1) >> $OUT
  echo "lin $FORM stats done"
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/lf; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/lf -- python3 $R/scripts/gpu_grid_prof.py 400 250 1000000 f32 3 > /dev/null 2>&1
    python3 $R/scripts/pmc_summary.py $(find /tmp/lf -name "*counter_collection.csv" | head -1) | grep k_lin >> $OUT
    echo "lin $FORM $C done"
  done
done
cd $R
cat $OUT
