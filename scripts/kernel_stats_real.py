"""Per kernel of a rocprofv3 --kernel-trace csv: launches, how many of them were EMPTY, and the average duration of the rest.

rr_pgo_optimize keeps one iteration enqueued behind the one that is running; the iteration behind the one that meets the stop
rule (pose_graph_optimization.rs:298-300) finds the stop word set and its factor / solve / update launches return at their
first instruction (a few microseconds each).  `rocprofv3 --stats` averages over ALL launches of a kernel, so its average for
k_factor_flow in a run of bench.py (one such launch per optimize() call, i.e. per six iterations on intel.g2o) sits ~10 % under
the duration of a launch that does work -- the figure bench.py's roofline uses (HIP events around working launches).  This script
separates the two: a launch counts as empty when it is shorter than a quarter of the kernel's median.
usage: kernel_stats_real.py <kernel_trace.csv>"""
import csv, statistics as st, sys
rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rrpgo::", "")
    by.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':44s} {'launches':>8s} {'empty':>6s} {'avg all us':>11s} {'avg working us':>15s} {'median us':>10s}")
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    med = st.median(v)
    work = [x for x in v if x >= 0.25 * med]
    print(f"{n[:44]:44s} {len(v):8d} {len(v) - len(work):6d} {sum(v) / len(v):11.2f} {sum(work) / max(len(work), 1):15.2f} {med:10.2f}")
