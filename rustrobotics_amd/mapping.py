"""Host-side mirror of the reference's `robotics::mapping` public surface
(reference src/mapping/mod.rs:6: `PoseGraph`, `PoseGraphSolver`), on top of the
C ABI of librr_pgo.so.  Same names, argument meaning and error behaviour:

  PoseGraph.new(file_path, solver)        pose_graph_optimization.rs:215-227
  PoseGraph.optimize(num_iterations, log, plot) -> list of chi2   :247-303
  PoseGraph.plot()                        :375-431   img/{name}-{iteration}-{solver:?}.svg

`optimize(n, False, False)` is ONE rr_pgo_optimize call (the loop, its stop rule and the Levenberg-Marquardt decisions
run on the device).  With `log` or `plot` the loop runs here, one iteration at a time through rr_pgo_linearize_solve /
rr_pgo_update / rr_pgo_chi2 in the reference's order, so that the lines are printed and the figures written as the
iterations complete (:258-268, :288-296).
"""
import ctypes as C
import enum

import numpy as np

from . import _lib


class PoseGraphSolver(enum.Enum):
    """pose_graph_optimization.rs:28-32"""
    GaussNewton = 0
    LevenbergMarquardt = 1


class PoseGraphError(RuntimeError):
    """Stands in for the reference's Box<dyn Error>."""

    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code


def _check(rc):
    if rc != 0:
        raise PoseGraphError(rc, _lib.load().rr_pgo_last_error().decode())


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class PoseGraph:
    def __init__(self, handle, solver, name=""):
        self._h = handle
        self.solver = solver
        self.name = name
        self.iteration = 0

    # -- constructors -----------------------------------------------------------
    @classmethod
    def new(cls, file_path, solver=PoseGraphSolver.GaussNewton, precision="f64", device=-1):
        """PoseGraph::new(file_path, solver)."""
        L = _lib.load()
        opt = _lib.Options()
        L.rr_pgo_default_options(C.byref(opt))
        opt.precision = _lib.PRECISIONS[precision]
        opt.device = device
        opt.solver = solver.value
        h = C.c_void_p()
        _check(L.rr_pgo_load_g2o(str(file_path).encode(), C.byref(opt), C.byref(h)))
        import os
        return cls(h, solver, os.path.splitext(os.path.basename(str(file_path)))[0])

    @classmethod
    def from_arrays(cls, node_kind, node_state, edge_kind, edge_from, edge_to, edge_meas, edge_info,
                    solver=PoseGraphSolver.GaussNewton, precision="f64", device=-1, node_id=None,
                    rank=0, world_size=1, sharded=False):
        L = _lib.load()
        keep = [np.ascontiguousarray(node_kind, np.int32), np.ascontiguousarray(node_state, np.float64),
                np.ascontiguousarray(edge_kind, np.int32), np.ascontiguousarray(edge_from, np.int32),
                np.ascontiguousarray(edge_to, np.int32), np.ascontiguousarray(edge_meas, np.float64),
                np.ascontiguousarray(edge_info, np.float64)]
        d = _lib.GraphDesc()
        d.n_nodes = len(keep[0])
        d.node_kind = _ip(keep[0])
        if node_id is not None:
            ids = np.ascontiguousarray(node_id, np.uint32)
            keep.append(ids)
            d.node_id = ids.ctypes.data_as(C.POINTER(C.c_uint32))
        d.node_state = _dp(keep[1])
        d.n_edges = len(keep[2])
        d.edge_kind = _ip(keep[2])
        d.edge_from = _ip(keep[3])
        d.edge_to = _ip(keep[4])
        d.edge_meas = _dp(keep[5])
        d.edge_info = _dp(keep[6])
        opt = _lib.Options()
        L.rr_pgo_default_options(C.byref(opt))
        opt.precision = _lib.PRECISIONS[precision]
        opt.device = device
        opt.solver = solver.value
        opt.rank, opt.world_size = rank, world_size
        opt.sharded = 1 if sharded else 0
        h = C.c_void_p()
        _check(L.rr_pgo_create(C.byref(d), C.byref(opt), C.byref(h)))
        return cls(h, solver)

    @classmethod
    def synthetic_grid(cls, width, height, n_edges=0, seed_meas=42, seed_init=43, **kw):
        """BASELINE config 4 graph, built by the library's own generator."""
        arrays = synthetic_grid_arrays(width, height, n_edges, seed_meas, seed_init)
        return cls.from_arrays(*arrays, **kw)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:   # not during interpreter teardown
            _lib._lib.rr_pgo_destroy(h)
            self._h = None

    # -- fields -------------------------------------------------------------------
    @property
    def len(self):
        return _lib.load().rr_pgo_dim(self._h)

    @property
    def num_nodes(self):
        return _lib.load().rr_pgo_num_nodes(self._h)

    @property
    def num_edges(self):
        return _lib.load().rr_pgo_num_edges(self._h)

    @property
    def anchor_node(self):
        return _lib.load().rr_pgo_anchor_node(self._h)

    def graph_arrays(self):
        """The parsed graph in rr_pgo_graph_desc packing (copies)."""
        L = _lib.load()
        d = _lib.GraphDesc()
        _check(L.rr_pgo_get_graph(self._h, C.byref(d)))
        n, m = d.n_nodes, d.n_edges
        nk = np.ctypeslib.as_array(d.node_kind, (n,)).copy() if n else np.zeros(0, np.int32)
        ek = np.ctypeslib.as_array(d.edge_kind, (m,)).copy() if m else np.zeros(0, np.int32)
        ns = int(sum({0: 3, 1: 2, 2: 7}[int(k)] for k in nk))
        nm = int(sum({0: 3, 1: 2, 2: 7}[int(k)] for k in ek))
        ni = int(sum({0: 6, 1: 3, 2: 21}[int(k)] for k in ek))
        return (nk, np.ctypeslib.as_array(d.node_state, (ns,)).copy(), ek,
                np.ctypeslib.as_array(d.edge_from, (m,)).copy(), np.ctypeslib.as_array(d.edge_to, (m,)).copy(),
                np.ctypeslib.as_array(d.edge_meas, (nm,)).copy(), np.ctypeslib.as_array(d.edge_info, (ni,)).copy())

    # -- the path -------------------------------------------------------------------
    def global_error(self):
        """global_error(&graph), :537-574"""
        out = C.c_double()
        _check(_lib.load().rr_pgo_chi2(self._h, C.byref(out)))
        return out.value

    def linearize_and_solve(self, lam=0.0, lm=False):
        """build_linear_system(lambda)?.solve()?, :271,:371-373"""
        dx = np.zeros(self.len)
        _check(_lib.load().rr_pgo_linearize_solve(self._h, lam, int(lm), _dp(dx)))
        return dx

    def update_nodes(self, dx, sign=1.0):
        """update_nodes(dx), :229-245"""
        dx = np.ascontiguousarray(dx, np.float64)
        if dx.shape != (self.len,):
            raise ValueError("dx has the wrong length")
        _check(_lib.load().rr_pgo_update(self._h, _dp(dx), sign))

    def optimize(self, num_iterations, log=False, plot=False, return_norms=False):
        """optimize(num_iterations, log, plot) -> Vec<f64> of chi2, :247-303"""
        if log or plot:
            return self._optimize_stepwise(num_iterations, log, plot, return_norms)
        L = _lib.load()
        errors = np.zeros(num_iterations + 1)
        norms = np.zeros(max(num_iterations, 1))
        n = C.c_int32()
        _check(L.rr_pgo_optimize(self._h, num_iterations, _dp(errors), C.byref(n), _dp(norms)))
        errors = errors[:n.value]
        self.iteration += n.value - 1
        if return_norms:
            return list(errors), list(norms[:n.value - 1])
        return list(errors)

    def _optimize_stepwise(self, num_iterations, log, plot, return_norms):
        """:247-303 statement by statement, for the calls that print or plot between iterations."""
        lm = self.solver == PoseGraphSolver.LevenbergMarquardt
        tolerance, lam, norms = 1e-4, 0.01, []          # :253-255
        last_error = self.global_error()
        errors = [last_error]
        if log:                                         # :258-265
            print(f"Loaded graph with {self.num_nodes} nodes and {self.num_edges} edges")
            print(f"initial error :{errors[-1]:.5f}", flush=True)
        if plot:                                        # :266-268
            self.plot()
        for i in range(num_iterations):
            self.iteration += 1
            dx = self.linearize_and_solve(lam, lm)      # :271 (lambda reaches the diagonal only for Levenberg-Marquardt, :362-366)
            self.update_nodes(dx)
            norm_dx = float(np.sqrt(np.dot(dx, dx)))    # :273
            error = self.global_error()
            if lm:                                      # :275-282
                if last_error < error:
                    self.update_nodes(dx, -1.0)
                    lam *= 2.0
                else:
                    lam /= 2.0
            last_error = error
            norms.append(norm_dx)
            errors.append(error)
            if log:                                     # :288-293
                print(f"step {i:3} : |dx| = {norm_dx:3.5f}, error = {errors[-1]:3.5f}", flush=True)
            if plot:                                    # :294-296
                self.plot()
            if norm_dx < tolerance:                     # :298-300
                break
        return (errors, norms) if return_norms else errors

    # -- PoseGraph::plot, :375-431 -------------------------------------------------------
    def plot_data(self):
        """What the reference's figure shows: the poses (blue circles), the same poses joined in the order of their ids
        (red line), the landmarks if there are any (red stars).  SE(3) graphs are `todo!()` in the reference (:398-399)."""
        L = _lib.load()
        d = _lib.GraphDesc()
        _check(L.rr_pgo_get_graph(self._h, C.byref(d)))
        n = d.n_nodes
        kinds = np.ctypeslib.as_array(d.node_kind, (n,)) if n else np.zeros(0, np.int32)
        if np.any(kinds == 2):
            raise PoseGraphError(_lib.EUNSUPPORTED, "plot of an SE(3) graph: todo!() in the reference (pose_graph_optimization.rs:398-399)")
        ids = np.ctypeslib.as_array(d.node_id, (n,)).astype(np.int64) if (n and d.node_id) else np.arange(n)
        st = self.state()
        offs = np.concatenate([[0], np.cumsum(np.where(kinds == 0, 3, 2))])[:-1]
        xy = np.stack([st[offs], st[offs + 1]], 1) if n else np.zeros((0, 2))
        pose = kinds == 0
        order = np.argsort(ids[pose], kind="stable")
        return {"poses": xy[pose], "poses_seq": xy[pose][order], "landmarks": xy[~pose],
                "file": f"img/{self.name}-{self.iteration}-{self.solver.name}.svg"}

    def plot(self, directory="."):
        """Writes the figure as SVG (the reference goes through plotpy / matplotlib; here the file is written directly)."""
        import os
        pd = self.plot_data()
        pts = [pd["poses"], pd["landmarks"]]
        allp = np.concatenate([p for p in pts if len(p)]) if any(len(p) for p in pts) else np.zeros((1, 2))
        lo, hi = allp.min(0), allp.max(0)
        span = float(max(hi[0] - lo[0], hi[1] - lo[1], 1e-9))      # equal axes (:425)
        size, margin = 640.0, 40.0
        scale = (size - 2 * margin) / span

        def px(p):
            return margin + (p[0] - lo[0]) * scale, size - margin - (p[1] - lo[1]) * scale

        out = [f'<svg xmlns="http://www.w3.org/2000/svg" width="{size:.0f}" height="{size:.0f}" viewBox="0 0 {size:.0f} {size:.0f}">',
               '<rect width="100%" height="100%" fill="white"/>']
        if len(pd["poses_seq"]):
            path = " ".join(f"{x:.2f},{y:.2f}" for x, y in map(px, pd["poses_seq"]))
            out.append(f'<polyline points="{path}" fill="none" stroke="red" stroke-width="1"/>')
        for p in pd["poses"]:
            x, y = px(p)
            out.append(f'<circle cx="{x:.2f}" cy="{y:.2f}" r="2" fill="blue"/>')
        for p in pd["landmarks"]:
            x, y = px(p)
            out.append(f'<text x="{x:.2f}" y="{y:.2f}" fill="red" font-size="12" text-anchor="middle" dominant-baseline="central">*</text>')
        out.append("</svg>")
        path = os.path.join(directory, pd["file"])
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write("\n".join(out) + "\n")
        return path

    def optimize_count(self, num_iterations):
        """rr_pgo_optimize with the buffers of the previous call: the iterations it executed (len(errors) - 1).
        What bench.py's timed loop calls -- optimize() itself, without this mirror's list building around it."""
        buf = getattr(self, "_opt_buf", None)
        if buf is None or len(buf[0]) < num_iterations + 1:
            errors, norms, n = np.zeros(num_iterations + 1), np.zeros(max(num_iterations, 1)), C.c_int32()
            buf = self._opt_buf = (errors, norms, n, _lib.load().rr_pgo_optimize, _dp(errors), C.byref(n), _dp(norms))
        rc = buf[3](self._h, num_iterations, buf[4], buf[5], buf[6])
        if rc != 0:
            _check(rc)
        done = buf[2].value - 1
        self.iteration += done
        return done

    def restarter(self, state):
        """A callable that puts the handle back into `state` (rr_pgo_set_state with everything bound once)."""
        state = np.ascontiguousarray(state, np.float64).copy()
        assert state.shape == (_lib.load().rr_pgo_state_len(self._h),)
        fn, h, ptr = _lib.load().rr_pgo_set_state, self._h, _dp(state)

        def restart(_keep=state):
            rc = fn(h, ptr)
            if rc != 0:
                _check(rc)
        return restart

    def state(self):
        out = np.zeros(_lib.load().rr_pgo_state_len(self._h))
        _check(_lib.load().rr_pgo_get_state(self._h, _dp(out)))
        return out

    def set_state(self, state):
        state = np.ascontiguousarray(state, np.float64)
        assert state.shape == (_lib.load().rr_pgo_state_len(self._h),)
        _check(_lib.load().rr_pgo_set_state(self._h, _dp(state)))

    # -- inspection / measurement ------------------------------------------------------
    def assemble(self, lam=0.0, lm=False):
        """Assembled normal matrix as (row_node, col_node, [blocks]) and b."""
        L = _lib.load()
        nb, nv = C.c_int32(), C.c_int64()
        _check(L.rr_pgo_assemble(self._h, lam, int(lm), C.byref(nb), None, None, None, None, C.byref(nv), None))
        br = np.zeros(nb.value, np.int32)
        bc = np.zeros(nb.value, np.int32)
        bo = np.zeros(nb.value, np.int64)
        vals = np.zeros(nv.value)
        b = np.zeros(self.len)
        _check(L.rr_pgo_assemble(self._h, lam, int(lm), C.byref(nb), _ip(br), _ip(bc),
                                 bo.ctypes.data_as(C.POINTER(C.c_int64)), _dp(vals), C.byref(nv), _dp(b)))
        return br, bc, bo, vals, b

    # -- sharding ONE graph over ranks (include/rr_pgo.h, "sharding") --------------------------------
    def exchange_info(self, which):
        """(device pointer, element count, element size) of exchange buffer `which` (0: boundary update
        matrices, all-gathered; 1: the two partial sums chi2 and |dx|^2, all-reduced)."""
        ptr, n, es = C.c_void_p(), C.c_int64(), C.c_int32()
        _check(_lib.load().rr_pgo_exchange_buffer(self._h, which, C.byref(ptr), C.byref(n), C.byref(es)))
        return ptr.value, n.value, es.value

    def bind_exchange(self, which, dev_ptr, n_elems):
        _check(_lib.load().rr_pgo_set_exchange_buffer(self._h, which, C.c_void_p(dev_ptr), n_elems))

    def stage(self, stage, lam=0.0, lm=False):
        _check(_lib.load().rr_pgo_stage(self._h, stage, lam, int(lm)))

    def stage_scalars(self):
        chi, nrm = C.c_double(), C.c_double()
        _check(_lib.load().rr_pgo_stage_scalars(self._h, C.byref(chi), C.byref(nrm)))
        return chi.value, nrm.value

    def node_owner(self):
        """rank owning every node, -1 = shared (top separators, anchor); zeros on an unsharded handle"""
        out = np.zeros(self.num_nodes, np.int32)
        _check(_lib.load().rr_pgo_node_owner(self._h, _ip(out)))
        return out

    def stream_ptr(self):
        """hipStream_t of the handle as an integer (torch.cuda.ExternalStream takes it)"""
        return int(_lib.load().rr_pgo_stream(self._h) or 0)

    def iterate_async(self, iters):
        _check(_lib.load().rr_pgo_iterate_async(self._h, iters))

    def sync(self):
        _check(_lib.load().rr_pgo_sync(self._h))

    def stats(self):
        s = _lib.Stats()
        _check(_lib.load().rr_pgo_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _lib.Stats._fields_ if k != "reserved"}

    @staticmethod
    def analyze(file_path, precision="f64"):
        """Host-only: parse + symbolic analysis, the statistics a handle on the file would report (no device needed)."""
        opt = _lib.Options()
        _lib.load().rr_pgo_default_options(C.byref(opt))
        opt.precision = _lib.PRECISIONS[precision]
        s = _lib.Stats()
        _check(_lib.load().rr_pgo_analyze_g2o(str(file_path).encode(), C.byref(opt), C.byref(s)))
        return {k: getattr(s, k) for k, _ in _lib.Stats._fields_ if k != "reserved"}

    def profile(self, iters):
        ms = np.zeros(_lib.NUM_KCLASS)
        n = np.zeros(_lib.NUM_KCLASS, np.int64)
        _check(_lib.load().rr_pgo_profile(self._h, iters, _dp(ms), n.ctypes.data_as(C.POINTER(C.c_int64)), _lib.NUM_KCLASS))
        return {name: (float(ms[i]), int(n[i])) for i, name in enumerate(_lib.KCLASS_NAMES)}


def parse_g2o_arrays(file_path):
    """parse_g2o (g2o.rs:35-143) through the library's loader, returned as the flat arrays `from_arrays` takes
    (needs a device: the loader entry point builds a handle)."""
    return PoseGraph.new(file_path).graph_arrays()


def synthetic_grid_arrays(width, height, n_edges=0, seed_meas=42, seed_init=43):
    L = _lib.load()
    s = C.c_void_p()
    d = _lib.GraphDesc()
    _check(L.rr_pgo_synth_grid(width, height, n_edges, seed_meas, seed_init, C.byref(s), C.byref(d)))
    try:
        n, m = d.n_nodes, d.n_edges
        out = (np.ctypeslib.as_array(d.node_kind, (n,)).copy(), np.ctypeslib.as_array(d.node_state, (3 * n,)).copy(),
               np.ctypeslib.as_array(d.edge_kind, (m,)).copy(), np.ctypeslib.as_array(d.edge_from, (m,)).copy(),
               np.ctypeslib.as_array(d.edge_to, (m,)).copy(), np.ctypeslib.as_array(d.edge_meas, (3 * m,)).copy(),
               np.ctypeslib.as_array(d.edge_info, (6 * m,)).copy())
    finally:
        L.rr_pgo_synth_free(s)
    return out
