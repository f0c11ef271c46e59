"""k_big_update dispatches from a rocprofv3 --pmc counter_collection.csv (SQ counters), longest first.
On gfx950 SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES
counts cycles (MI355X_MICROARCH.md)."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_big_update" in r["Kernel_Name"]]
disp = collections.defaultdict(dict)
for r in rows:
    d = disp[r["Dispatch_Id"]]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["wgs"] = int(r["Grid_Size"]) // 256
    d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ds = sorted(disp.values(), key=lambda d: -d["us"])
def line(tag, group):
    t = collections.Counter()
    for d in group:
        for k, v in d.items(): t[k] += v
    wc = t["SQ_WAVE_CYCLES"] or 1
    return (f"{tag:34s} n={len(group):4d} us={t['us']:9.1f} wgs={t['wgs']:8d} waves={t['SQ_WAVES']:.3g} "
            f"wait_any/wave={t['SQ_WAIT_ANY']/wc:.2f} wait_inst/wave={t['SQ_WAIT_INST_ANY']/wc:.2f} active/wave={t['SQ_ACTIVE_INST_ANY']/wc:.2f} "
            f"mfma_busy_cycles/(4*wave_quadcycles)={t['SQ_VALU_MFMA_BUSY_CYCLES']/(4*wc):.3f} lds_conflict/wave={t['SQ_LDS_BANK_CONFLICT']/wc:.3f} "
            f"wave_quadcycles/wave={wc/max(t['SQ_WAVES'],1):.0f}")
for i, d in enumerate(ds[:6]): print(line(f"#{i} {d['us']:.0f}us {d['wgs']} WGs", [d]))
print(line("dispatches > 150 us", [d for d in ds if d["us"] > 150]))
print(line("dispatches 40..150 us", [d for d in ds if 40 < d["us"] <= 150]))
print(line("dispatches <= 40 us", [d for d in ds if d["us"] <= 40]))
