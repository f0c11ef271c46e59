"""Concurrency seen in a rocprofv3 kernel trace: sum of kernel durations vs the length of their union,
and how much of k_big_update's time runs while a panel kernel is also running."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ev.sort()
t0 = ev[len(ev) // 2][0]
ev = [e for e in ev if e[0] >= t0]          # second half: past the warm-up
tot = sum(e[1] - e[0] for e in ev)
union, cur_s, cur_e = 0, None, None
for s, e, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
union += cur_e - cur_s
span = max(e[1] for e in ev) - ev[0][0]
print(f"kernels {len(ev)}  sum of durations {tot/1e6:.2f} ms  union {union/1e6:.2f} ms  wall span {span/1e6:.2f} ms")
upd = [(s, e) for s, e, n in ev if "k_big_update" in n]
pan = [(s, e) for s, e, n in ev if "k_big_panel32" in n or "k_big_diag32" in n]
ov = 0; j = 0
for s, e in upd:
    for ps, pe in pan:
        if pe <= s: continue
        if ps >= e: break
        ov += min(e, pe) - max(s, ps)
print(f"k_big_update total {sum(e-s for s,e in upd)/1e6:.2f} ms, of which concurrent with a panel kernel {ov/1e6:.2f} ms; panel kernels total {sum(e-s for s,e in pan)/1e6:.2f} ms")
by = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    k = n.split("(")[0][-40:]; by[k][0] += 1; by[k][1] += e - s
for k, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]: print(f"  {k:42s} n={c:6d} total {d/1e6:8.2f} ms avg {d/c/1e3:8.1f} us")
