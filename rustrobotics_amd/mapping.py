"""Host-side mirror of the reference's `robotics::mapping` public surface
(reference src/mapping/mod.rs:6: `PoseGraph`, `PoseGraphSolver`), on top of the
C ABI of librr_pgo.so.  Same names, argument meaning and error behaviour:

  PoseGraph.new(file_path, solver)        pose_graph_optimization.rs:215-227
  PoseGraph.optimize(num_iterations, log, plot) -> list of chi2   :247-303

`plot` is accepted for signature parity and must be False (plotting is a side
output of the reference, SURVEY.md L0, out of scope).
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
        if plot:
            raise PoseGraphError(_lib.EUNSUPPORTED, "plotting is out of scope of this backend")
        L = _lib.load()
        errors = np.zeros(num_iterations + 1)
        norms = np.zeros(max(num_iterations, 1))
        n = C.c_int32()
        if log:  # :258-265
            print(f"Loaded graph with {self.num_nodes} nodes and {self.num_edges} edges")
        _check(L.rr_pgo_optimize(self._h, num_iterations, _dp(errors), C.byref(n), _dp(norms)))
        errors = errors[:n.value]
        self.iteration += n.value - 1
        if log:
            print(f"initial error :{errors[0]:.5f}")
            for i in range(n.value - 1):  # :288-293
                print(f"step {i:3} : |dx| = {norms[i]:3.5f}, error = {errors[i + 1]:3.5f}")
        if return_norms:
            return list(errors), list(norms[:n.value - 1])
        return list(errors)

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
