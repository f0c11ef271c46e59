"""Non-interactive counterpart of the reference's example and bench for this path.

  python -m rustrobotics_amd <file.g2o> [--solver GaussNewton|LevenbergMarquardt] [--iterations 50]
                                        [--precision f64|f32|mixed] [--plot]
      = examples/mapping/pose_graph_optimization.rs:49-50   PoseGraph::new(file, solver)?.optimize(50, true, plot)
        (the reference picks file / solver / plot from interactive menus; --plot writes img/{name}-{iteration}-{solver}.svg
        before the first and after every iteration, like :266-268, :294-296)

  python -m rustrobotics_amd <file.g2o> --bench [--repeats 20]
      = benches/graph_slam.rs:9-10   PoseGraph::new("dataset/g2o/intel.g2o", GaussNewton)?.optimize(10, false, false)
        timed end to end like criterion does: parsing, symbolic analysis, device setup and the ten
        Gauss-Newton iterations are ALL inside the timed closure; prints mean / median / min in ms.
"""
import argparse
import statistics
import sys
import time

from .mapping import PoseGraph, PoseGraphSolver


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m rustrobotics_amd", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("file", help="g2o file (SE2, SE2 + XY landmarks, or SE3:QUAT)")
    ap.add_argument("--solver", choices=[s.name for s in PoseGraphSolver], default="GaussNewton")
    ap.add_argument("--iterations", type=int, default=None, help="default 50 (example) / 10 (--bench)")
    ap.add_argument("--precision", choices=["f64", "f32", "mixed"], default="f64")
    ap.add_argument("--plot", action="store_true", help="the example's third menu: write ./img/{name}-{iteration}-{solver}.svg")
    ap.add_argument("--bench", action="store_true", help="time new() + optimize(10, false, false) like benches/graph_slam.rs")
    ap.add_argument("--repeats", type=int, default=20)
    a = ap.parse_args(argv)
    solver = PoseGraphSolver[a.solver]
    if not a.bench:
        graph = PoseGraph.new(a.file, solver, precision=a.precision)
        graph.optimize(50 if a.iterations is None else a.iterations, True, a.plot)
        return 0
    iters = 10 if a.iterations is None else a.iterations
    PoseGraph.new(a.file, solver, precision=a.precision).optimize(iters, False, False)   # warm-up: library load, HIP context
    ms = []
    for _ in range(a.repeats):
        t0 = time.perf_counter()
        errors = PoseGraph.new(a.file, solver, precision=a.precision).optimize(iters, False, False)
        ms.append((time.perf_counter() - t0) * 1e3)
    print(f"graph_slam: new() + optimize({iters}, false, false) on {a.file}: mean {statistics.mean(ms):.3f} ms, "
          f"median {statistics.median(ms):.3f} ms, min {min(ms):.3f} ms over {a.repeats} runs; "
          f"{len(errors) - 1} iterations run, final chi2 {errors[-1]:.9g}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
