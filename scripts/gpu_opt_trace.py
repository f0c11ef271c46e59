"""Diagnostic: a few optimize(10) calls from the initial state (the shape bench.py's headline times), for
`rocprofv3 --kernel-trace`; with an argument `analyse <csv>` prints the timeline of the last call.
usage: gpu_opt_trace.py <dataset> <calls>   |   gpu_opt_trace.py analyse <kernel_trace.csv>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "analyse":
    import csv
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r['Start_Timestamp']))
    name = lambda r: r['Kernel_Name'].split('(')[0].split('<')[0].replace('void rrpgo::', '')
    # calls are separated by the largest gaps: print the last 40 kernels with start offsets
    tail = rows[-40:]
    t0 = int(tail[0]['Start_Timestamp'])
    prev_end = t0
    for r in tail:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(f"{(s - t0) / 1e3:9.2f} us  +gap {(s - prev_end) / 1e3:7.2f}  dur {(e - s) / 1e3:8.2f}  {name(r)}")
        prev_end = e
    sys.exit(0)
from rustrobotics_amd import PoseGraph
g = PoseGraph.new(os.path.join(ROOT, 'tests/golden/g2o', sys.argv[1] + '.g2o'))
s0 = g.state()
for rep in range(int(sys.argv[2])):
    g.set_state(s0)
    g.optimize_count(10)
g.sync()
