"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, average counter value per dispatch."""
import csv, sys, collections
path, = sys.argv[1:]
rows = list(csv.DictReader(open(path)))
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    k = (r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:48]:
    print(f"{k:62s} {c:12s} dispatches {n:6d}  avg/dispatch {v / n:14.1f}  total {v:16.1f}")
