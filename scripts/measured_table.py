"""The 'Measured' block of DESIGN.md (between its measured:begin / measured:end markers) from the artefacts of a round:
python3 scripts/measured_table.py TAG [--write]   (profiles/<TAG>_bench_*.json, profiles/<TAG>_lattice_levels.txt)"""
import io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = io.StringIO()
_print = print
def print(*a):   # noqa: A001 -- collect the block
    _print(*a, file=out)
rows = [("intel_f64", "intel.g2o fp64 (configs[1], the headline line)"), ("intel_mixed", "intel.g2o mixed (f64 state / gradient, f32 factor)"),
        ("m3500_f64", "input_M3500 fp64 (configs[2])"), ("dlr_f64", "dlr.g2o fp64 (pose-landmark, 17.6 k edges)"),
        ("grid_mixed", "lattice 100 k poses / 1 M edges, mixed = fp64 state + gradient, fp32 factor (configs[3], primary: meets the stop rule)"),
        ("grid_f32", "the same lattice in pure fp32 (configs[3] as BASELINE words it; |dx| floors above the stop rule)"),
        ("sphere2500_f64", "sphere2500 fp64 SE(3) (configs[4])")]
print("| workload | GN it/s through `optimize()` | ms/step | `iterate_async` it/s | launches/step | per-step time by kernel class (µs, HIP events) | roofline of the dominant class |")
print("|---|---|---|---|---|---|---|")
for key, label in rows:
    d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_{key}.json")))
    r = d["roofline"]
    cls = {k: v for k, v in r["per_step_us_by_kernel_class"].items() if v >= 0.5}
    cls_s = ", ".join(f"{k} {v:.0f}" for k, v in sorted(cls.items(), key=lambda kv: -kv[1]))
    roof = f"`{r['kernel']}`: {r['achieved']:.3g} {r['unit']} = {100 * r['frac']:.2g} % of the {r['bound'].upper()} peak ({r['peak']:g})"
    mk = d.get("mfma_kernel")
    if mk:
        roof += f"; `{mk['kernel']}`: {mk['achieved']:.1f} TFLOP/s = {100 * mk['frac']:.1f} % of the fp32 MFMA peak over {mk['launches_per_step']:.0f} launches"
    if r.get("traffic"):
        roof += f"; counter traffic {r['traffic'] / 1e6:.3g} MB per launch" + (f" = {r['traffic_ratio']:.1f} x algorithmic" if r.get("traffic_ratio") else "")
    ia = d.get("iterate_async") or {}
    print(f"| {label} | **{d['value']:.0f}** | {d['ms_per_step']:.3f} | {ia.get('value', 0):.0f} | {d['launches_per_step']} | {cls_s} | {roof} |")
dflt = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_default.json")))
cb = dflt.get("cpu_baseline") or {}
print()
print(f"Default line (`python bench.py`, what the driver runs): {dflt['value']:.0f} {dflt['unit']}, CPU oracle on one core of the same box "
      f"{cb.get('value', 0):.0f} {cb.get('unit', '')} ({cb.get('kind')}; {cb.get('sample', '')[:120]}).")
for s in dflt.get("secondary", []):
    cbs = s.get("cpu_baseline") or {}
    print(f"* secondary: {s.get('workload', '?')[:60]} {s.get('dtype')} {s.get('parallelism')}: {s.get('value', 0):.1f} it/s, {s.get('ms_per_step', 0):.3f} ms"
          + (f"; CPU oracle {cbs['value']:.3g} it/s" + (" (offline)" if cbs.get("offline") else "") if cbs else "")
          + (f"; stopped by the reference's rule: {s['stopped_by_reference_rule']}" if "stopped_by_reference_rule" in s else ""))

print()
print(f"Per tree level of the lattice (rocprofv3 kernel trace of one iteration with plain launches, `profiles/{tag}_lattice_levels.txt`; "
      "`nf` = fronts of the level, launches x count = their total time in µs):")
print()
print("```")
for line in open(os.path.join(ROOT, "profiles", f"{tag}_lattice_levels.txt")):
    if line.startswith("level"):
        print(line.rstrip())
print("```")
block = out.getvalue()
if "--write" in sys.argv:
    path = os.path.join(ROOT, "DESIGN.md")
    doc = open(path).read()
    a, b = doc.index("<!-- measured:begin -->"), doc.index("<!-- measured:end -->")
    open(path, "w").write(doc[:a] + "<!-- measured:begin -->\n" + block + doc[b:])
else:
    _print(block)
