"""A/B of environment knobs on one workload: python scripts/gpu_ab_env.py WORKLOAD PREC "K=V K=V" "K=V" ...  (each arg one config)"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch  # noqa
    sys.path.insert(0, ROOT)
    from bench import make_graph
    wl, prec, steps = sys.argv[2], sys.argv[3], int(sys.argv[4])
    g = make_graph(wl, prec, 0)
    s0 = g.state()
    best = 1e9
    for rep in range(3):
        g.set_state(s0); g.iterate_async(3); g.sync(); g.set_state(s0)
        t0 = time.perf_counter(); g.iterate_async(steps); g.sync(); best = min(best, (time.perf_counter() - t0) / steps)
    prof = g.profile(5)
    print(json.dumps({"ms": best * 1e3, "classes": {k: round(v[0] / 5 * 1e3, 1) for k, v in prof.items() if v[0] > 0}}))
    sys.exit(0)
wl, prec = sys.argv[1], sys.argv[2]
steps = 200 if not wl.startswith("grid:") else 20
for cfg in sys.argv[3:]:
    env = dict(os.environ)
    for kv in cfg.split():
        if "=" in kv:
            k, v = kv.split("=", 1); env[k] = v
    out = subprocess.run([sys.executable, __file__, "--child", wl, prec, str(steps)], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print("%-50s %s" % (cfg, line[-1] if line else out.stderr[-300:]), flush=True)
