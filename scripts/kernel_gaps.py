"""Diagnostic: per-kernel durations and the idle gaps between consecutive kernels of a rocprofv3 --kernel-trace csv.
usage: kernel_gaps.py <kernel_trace.csv>"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].split('(')[0].split('<')[0].replace('void rrpgo::', '')
dur, gap = {}, {}
for a, b in zip(rows, rows[1:]):
    dur.setdefault(name(a), []).append((int(a['End_Timestamp']) - int(a['Start_Timestamp'])) / 1e3)
    gap.setdefault((name(a), name(b)), []).append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
print('kernel: launches, median duration us')
for k, v in sorted(dur.items(), key=lambda kv: -len(kv[1]))[:12]:
    print(f'  {k:28s} {len(v):6d} {st.median(v):8.2f}')
print('gap after -> before: count, median us')
for k, v in sorted(gap.items(), key=lambda kv: -len(kv[1]))[:12]:
    print(f'  {k[0]:24s} -> {k[1]:24s} {len(v):6d} {st.median(v):8.2f}')
