"""Per-level summary of a gpurun_out/timeline_<tag>.json (scripts/gpu_timeline.sh): one line per batch of fronts beyond LDS."""
import json, sys
t = json.load(open(sys.argv[1]))
levels, cur, tot = [], None, {}
for name, start, dur, gx, gy, gz, wg in t:
    short = name.split('<')[0]
    tot[short] = tot.get(short, 0) + dur / 1e3
    if short in ('k_big_zero', 'k_big_build'):
        cur = {'start': start, 'k': {}, 'nf': gy}
        levels.append(cur)
    if cur is not None and short.startswith('k_big') and 'gemv' not in short:
        d = cur['k'].setdefault(short, [0, 0.0, []])
        d[0] += 1; d[1] += dur / 1e3; d[2].append((gx // wg if wg else gx, gy, gz, round(dur / 1e3, 1)))
        cur['end'] = start + dur
for i, l in enumerate(levels):
    print(f"level {i:2d}: nf={l['nf']:4d} span {(l['end']-l['start'])/1e3:7.1f} us  " +
          "  ".join(f"{k[6:]}:{v[0]}x={v[1]:.0f}" for k, v in l['k'].items()))
    if len(sys.argv) > 2:
        for k, v in l['k'].items():
            print("      ", k, v[2])
print({k: round(v) for k, v in tot.items()})
