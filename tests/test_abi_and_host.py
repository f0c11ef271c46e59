"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/rr_pgo.h
declares, the loader's failure modes, the synthetic generator, and the symbolic analysis
(checked by a host-only multifrontal walk of the same tables the HIP kernels use)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, g2o_path

CSRC = os.path.join(ROOT, "rustrobotics_amd", "csrc")


@pytest.fixture(scope="session")
def lib():
    from rustrobotics_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", CSRC])
    return _lib.load()


def test_exports_match_header(lib):
    from rustrobotics_amd import _lib
    header = open(os.path.join(ROOT, "include", "rr_pgo.h")).read()
    declared = set(re.findall(r"\b(rr_pgo_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name  # dlsym succeeds


@pytest.fixture(scope="session")
def abi_client(tmp_path_factory, lib):
    """tests/native/abi_client.c: include/rr_pgo.h compiled AS C (strict C99, warnings are errors) and linked against
    the library -- the header is the product's contract with compiled callers (the reference's are Rust:
    src/mapping/mod.rs:6, benches/graph_slam.rs:9-10), so it must stand without a C++ compiler or Python around it."""
    from rustrobotics_amd import _lib
    exe = tmp_path_factory.mktemp("abi") / "abi_client"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic-errors", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "abi_client.c"), _lib.LIB_PATH, f"-Wl,-rpath,{libdir}",
                           "-o", str(exe)])
    return str(exe)


def test_struct_layouts_match_the_c_compiler(abi_client):
    """sizeof / offsetof of every field of the three public structs as gcc lays them out (the program also carries
    _Static_asserts for the values INTEGRATION.md's #[repr(C)] structs rely on) against the ctypes mirror."""
    from rustrobotics_amd import _lib
    out = subprocess.run([abi_client, "layout"], capture_output=True, text=True, check=True).stdout
    seen = {}
    for line in out.splitlines():
        struct, field, off, size = line.split()
        seen[(struct, field)] = (int(off), int(size))
    for cname, cls in (("rr_pgo_options", _lib.Options), ("rr_pgo_graph_desc", _lib.GraphDesc), ("rr_pgo_stats", _lib.Stats)):
        assert seen.pop((cname, ".")) == (0, C.sizeof(cls)), cname
        for fname, _ in cls._fields_:
            d = getattr(cls, fname)
            assert seen.pop((cname, fname)) == (d.offset, d.size), (cname, fname)
    assert seen.pop(("enum", "RR_PGO_NUM_KCLASS"))[0] == _lib.NUM_KCLASS == len(_lib.KCLASS_NAMES)
    assert seen.pop(("enum", "RR_PGO_ABI_VERSION"))[0] == _lib.ABI_VERSION == _lib.load().rr_pgo_abi_version()
    assert not seen, f"fields the ctypes mirror does not know: {seen}"


# ---- INTEGRATION.md's Rust binding against the header -------------------------------------------------------------

_C_BASE = {"char": "c_char", "double": "f64", "int32_t": "i32", "int64_t": "i64", "uint32_t": "u32", "uint64_t": "u64",
           "int": "c_int", "void": "c_void"}


def _c_type(text):
    """'const rr_pgo_options *opt' -> ('rr_pgo_options', ['const'])  (pointer levels, outermost last; [] = by value)"""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S).strip()
    m = re.match(r"^(const\s+)?(\w+)\s*((?:\*\s*(?:const\s*)?)*)\s*(\w+)?$", text)
    assert m, text
    base = _C_BASE.get(m.group(2), m.group(2))
    stars = m.group(3).count("*")
    levels = []
    if stars:
        levels = ["const" if m.group(1) else "mut"] + ["mut"] * (stars - 1)   # `const T **` does not occur in the header
    return base, levels


def _rust_type(text):
    text = text.strip().replace("std::ffi::c_void", "c_void")
    levels = []
    while text.startswith("*"):
        kind, text = text[1:].split(None, 1)
        levels.append(kind)
        text = text.strip()
    return text, levels[::-1]   # innermost pointer first, like _c_type


def test_integration_md_rust_binding_matches_the_header():
    """INTEGRATION.md, section 2 (the `extern "C"` block and the #[repr(C)] structs a maintainer of the reference would
    paste into src/mapping/): every function of include/rr_pgo.h, same name, arity, scalar types and pointer
    mutability; every struct field in the same order with the same type."""
    header = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "rr_pgo.h")).read(), flags=re.S)
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rust = md[md.index("// src/mapping/rr_pgo_sys.rs"):]
    rust = rust[:rust.index("```")]
    rust = re.sub(r"//[^\n]*", "", rust)
    # functions
    cfun = {}
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(rr_pgo_\w+)\s*\(([^)]*)\)\s*;", header):
        params = [] if args.strip() in ("", "void") else [_c_type(a) for a in args.split(",")]
        ret = ret.replace("extern", "").strip()
        cfun[name] = (_c_type(ret + " x") if ret != "void" else None, params)
    rfun = {}
    for name, args, ret in re.findall(r"pub fn (rr_pgo_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", rust):
        params = [_rust_type(a.split(":", 1)[1]) for a in args.split(",") if a.strip()]
        rfun[name] = (_rust_type(ret) if ret else None, params)
    assert set(cfun) == set(rfun), set(cfun) ^ set(rfun)
    for name in cfun:
        assert cfun[name] == rfun[name], (name, cfun[name], rfun[name])
    # structs
    for sname in ("rr_pgo_options", "rr_pgo_graph_desc", "rr_pgo_stats"):
        cbody = re.search(r"typedef struct %s \{(.*?)\} %s;" % (sname, sname), header, flags=re.S).group(1)
        cfields = []
        for decl in cbody.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"^(const\s+)?(\w+)\s*(\*?)\s*(.*)$", decl)
            base = _C_BASE.get(m.group(2), m.group(2))
            for var in m.group(4).split(","):
                var = var.strip()
                arr = re.match(r"^(\*?)\s*(\w+)\[(\d+)\]$", var)
                ptr = (m.group(3) == "*") or var.startswith("*")
                vname = arr.group(2) if arr else var.lstrip("* ")
                cfields.append((vname, f"[{base}; {arr.group(3)}]" if arr else base, ["const" if m.group(1) else "mut"] if ptr else []))
        rbody = re.search(r"pub struct %s \{(.*?)\n\}" % sname, rust, flags=re.S).group(1)
        rfields = []
        for fname, ftype in re.findall(r"pub (\w+)\s*:\s*([^,\n]+(?:;\s*\d+\])?)", rbody):
            t, lv = _rust_type(ftype.strip().rstrip(","))
            rfields.append((fname, t, lv))
        assert cfields == rfields, (sname, cfields, rfields)


def test_loader_parses_large_files_in_parts_like_a_sequential_pass(lib, tmp_path):
    """g2o_loader.cpp cuts files beyond 96 KB into up to four runs of whole lines parsed side by side (that
    path had no test -- every failure case above is a one-line file).  A 1 MB file: the first failure IN FILE ORDER wins
    with the line number a sequential pass would print, whichever part it sits in; CRLF line ends and a missing final
    newline parse like the plain file (the reference reads with str::lines(), g2o.rs:52)."""
    from rustrobotics_amd import _lib
    n = 12000
    verts = [f"VERTEX_SE2 {i} {i * 0.5:.6f} {(i % 7) * 0.25:.6f} {(i % 13) * 0.01:.6f}" for i in range(n)]
    edges = [f"EDGE_SE2 {i} {i + 1} 0.500000 0.000000 0.010000 100.000000 0.000000 0.000000 100.000000 0.000000 400.000000" for i in range(n - 1)]
    lines = verts + edges
    assert sum(len(x) + 1 for x in lines) > 1000000
    opt = _lib.Options()
    lib.rr_pgo_default_options(C.byref(opt))
    st = _lib.Stats()

    def analyse(name, text_bytes):
        p = tmp_path / name
        p.write_bytes(text_bytes)
        rc = lib.rr_pgo_analyze_g2o(str(p).encode(), C.byref(opt), C.byref(st))   # host only: parse + symbolic analysis
        return rc, lib.rr_pgo_last_error().decode()

    rc, msg = analyse("ok.g2o", ("\n".join(lines) + "\n").encode())
    assert rc == 0, msg
    blocks = st.nnz_h_blocks
    assert blocks == n + (n - 1)
    for name, data in (("crlf.g2o", ("\r\n".join(lines) + "\r\n").encode()), ("no_final_newline.g2o", "\n".join(lines).encode())):
        rc, msg = analyse(name, data)
        assert rc == 0 and st.nnz_h_blocks == blocks, (name, rc, msg)

    def message_at_line_one(bad_line):
        rc, msg = analyse("one.g2o", (bad_line + "\n").encode())
        assert rc == _lib.EPARSE and msg.startswith("line 1:"), msg
        return msg[len("line 1:"):]

    bad_float = "VERTEX_SE2 999999 0 zero 0"
    tail = message_at_line_one(bad_float)
    for where in (10, len(lines) // 4 + 5, len(lines) // 2 + 7, len(lines) - 3):   # part 0 ... part 3
        broken = list(lines)
        broken[where] = bad_float
        rc, msg = analyse("bad.g2o", ("\n".join(broken) + "\n").encode())
        assert rc == _lib.EPARSE and msg == f"line {where + 1}:{tail}", (where, msg)
    # two failures in different parts: the earlier line is reported; a duplicate vertex id late in the file, with its line
    broken = list(lines)
    broken[len(lines) - 100] = bad_float
    broken[50] = bad_float
    rc, msg = analyse("bad2.g2o", ("\n".join(broken) + "\n").encode())
    assert rc == _lib.EPARSE and msg == f"line 51:{tail}", msg
    broken = list(lines)
    broken[n - 5] = verts[3]          # vertex id 3 again, in a later part than its first occurrence
    rc, msg = analyse("dup.g2o", ("\n".join(broken) + "\n").encode())
    assert rc == _lib.EPARSE and msg.startswith(f"line {n - 4}: duplicate vertex id 3"), msg
    broken[20] = bad_float            # ... and a parse error in part 0 comes first
    rc, msg = analyse("dup2.g2o", ("\n".join(broken) + "\n").encode())
    assert rc == _lib.EPARSE and msg == f"line 21:{tail}", msg


def test_trim_needs_no_device_and_no_handles(lib):
    """rr_pgo_trim gives back what the library keeps between handles (include/rr_pgo.h): callable at any time, also before the
    first handle and on a host without a GPU."""
    lib.rr_pgo_trim.restype = C.c_int
    assert lib.rr_pgo_trim() == 0 and lib.rr_pgo_trim() == 0


def test_integration_md_shim_keeps_the_reference_public_surface():
    """the four public items of robotics::mapping on this path -- PoseGraphSolver, PoseGraph::new,
    PoseGraph::optimize(num_iterations, log, plot), PoseGraph::plot -- with the reference's signatures
    (pose_graph_optimization.rs:28-32, :215, :247-252, :375), in the Rust shim of INTEGRATION.md section 3 AND in the Python
    mirror; the shim's stepping loop calls only entry points the extern block declares."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shim = md[md.index("// src/mapping/pose_graph_optimization.rs"):]
    shim = shim[:shim.index("```")]
    flat = re.sub(r"\s+", " ", re.sub(r"//[^\n]*", "", shim))
    assert "pub enum PoseGraphSolver { GaussNewton, LevenbergMarquardt }" in flat
    assert "pub struct PoseGraph<'a>" in flat and "impl<'a> PoseGraph<'a>" in flat
    assert "pub fn new(file_path: &str, solver: PoseGraphSolver) -> Result<PoseGraph, Box<dyn Error>>" in flat
    assert "pub fn optimize(&mut self, num_iterations: usize, log: bool, plot: bool) -> Result<Vec<f64>, Box<dyn Error>>" in flat
    assert "pub fn plot(&self) -> Result<(), Box<dyn Error>>" in flat
    assert 'format!("img/{}-{}-{:?}.svg", self.name, self.iteration, self.solver)' in flat      # :428
    sys_block = md[md.index("// src/mapping/rr_pgo_sys.rs"):]
    declared = set(re.findall(r"pub fn (rr_pgo_\w+)", sys_block[:sys_block.index("```")]))
    used = set(re.findall(r"sys::(rr_pgo_\w+)\(", shim))
    assert used and used <= declared, used - declared
    assert {"rr_pgo_optimize", "rr_pgo_linearize_solve", "rr_pgo_update", "rr_pgo_chi2", "rr_pgo_get_state", "rr_pgo_get_graph"} <= used
    # the Python mirror: same names, same argument order
    import inspect
    from rustrobotics_amd import PoseGraph, PoseGraphSolver
    assert [s.name for s in PoseGraphSolver] == ["GaussNewton", "LevenbergMarquardt"]
    assert list(inspect.signature(PoseGraph.optimize).parameters)[:4] == ["self", "num_iterations", "log", "plot"]
    assert list(inspect.signature(PoseGraph.new).parameters)[:2] == ["file_path", "solver"]
    assert callable(PoseGraph.plot)


def _load(lib, path):
    from rustrobotics_amd import _lib
    opt = _lib.Options()
    lib.rr_pgo_default_options(C.byref(opt))
    h = C.c_void_p()
    rc = lib.rr_pgo_load_g2o(str(path).encode(), C.byref(opt), C.byref(h))
    return rc, h, lib.rr_pgo_last_error().decode()


def test_loader_failure_modes(lib, tmp_path):
    """Failure cases of g2o.rs:35-143: Err(io) -> EIO; panics/Err(parse) -> EPARSE.
    They are detected before any device work, so they are testable without a GPU."""
    from rustrobotics_amd import _lib
    rc, _, msg = _load(lib, tmp_path / "missing.g2o")
    assert rc == _lib.EIO, msg
    cases = {
        "tab.g2o": "VERTEX_SE2\t0 0 0 0\n",                       # split is on ' ' only (g2o.rs:52)
        "tag.g2o": "VERTEX_SE2 0 0 0 0\nFIX 0\n",                 # unimplemented!() (g2o.rs:138)
        "empty_line.g2o": "VERTEX_SE2 0 0 0 0\n\nVERTEX_SE2 1 1 0 0\n",  # line[0] panics (g2o.rs:53)
        "count.g2o": "VERTEX_SE2 0 0 0\n",                        # slice pattern -> todo!() (g2o.rs:56-58)
        "float.g2o": "VERTEX_SE2 0 0 zero 0\n",                   # parse::<f64>().unwrap() (g2o.rs:30)
        "id.g2o": "VERTEX_SE2 -1 0 0 0\n",                        # u32 parse error (g2o.rs:55)
        "dangling.g2o": "VERTEX_SE2 0 0 0 0\nEDGE_SE2 0 7 1 0 0 1 0 0 1 0 1\n",  # lut.get().unwrap() (:312-313)
    }
    for name, text in cases.items():
        p = tmp_path / name
        p.write_text(text)
        rc, _, msg = _load(lib, p)
        assert rc == _lib.EPARSE, (name, rc, msg)


def test_no_device_is_loud(lib):
    """Without a GPU the library must fail with ENODEVICE, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rustrobotics_amd import _lib
    rc, _, msg = _load(lib, g2o_path("simulation-pose-landmark"))
    assert rc == _lib.ENODEVICE and "no CPU fallback" in msg


def test_synthetic_grid_counts(lib):
    """the 10-offset stencil gives sum (W-|dx|)(H-dy) edges; 400x250 -> 992,860."""
    from rustrobotics_amd import synthetic_grid_arrays
    def closed_form(W, H):
        offs = [(0, 1), (1, 0), (-1, 1), (1, 1), (0, 2), (2, 0), (-2, 1), (-1, 2), (1, 2), (2, 1)]
        return sum((W - abs(dx)) * (H - dy) for dx, dy in offs)
    nk, ns, ek, ef, et, em, ei = synthetic_grid_arrays(40, 25)
    assert len(nk) == 1000 and len(ek) == closed_form(40, 25)
    assert closed_form(400, 250) == 992860
    assert (ef < et).all() and ef[0] == 0           # from = lower index, pose 0 is the anchor
    assert len(set(zip(ef.tolist(), et.tolist()))) == len(ef)  # unique pairs
    a2 = synthetic_grid_arrays(40, 25)
    assert all(np.array_equal(x, y) for x, y in zip((nk, ns, ek, ef, et, em, ei), a2))  # deterministic
    # ground truth is a near-minimum: measurements are consistent with a unit lattice
    assert abs(np.hypot(em[0::3], em[1::3]).max() - np.hypot(2, 2)) < 0.6
    nk2, *_rest = synthetic_grid_arrays(40, 25, closed_form(40, 25) + 100)
    assert len(_rest[1]) == closed_form(40, 25) + 100


def _restated_grid(W, H, n_edges_target=0, seed_meas=42, seed_init=43):
    """SURVEY.md 8(d), BASELINE configs[3], restated in Python from the survey's text, independently of synth_grid.cpp:
    W x H unit lattice, pose index in boustrophedon order (x reversed on odd rows), ground-truth heading 0 on even and
    pi on odd rows; edges cell-major in (y, x) raster order and offset-minor over the ten stencil offsets, `from` = the
    lower pose index, then -- when more edges are asked for -- offset (-2, 2) in raster order; measurement =
    x_from^-1 x_to of the ground truth + N(0, diag(.05, .05, .01)^2), information diag(400, 400, 10000), initial guess =
    ground truth + N(0, diag(.1, .1, .02)^2).  Random stream: splitmix64 from the seed; a normal pair by Box-Muller from
    u1 = ((z >> 11) + 1) / 2^53 (never 0) and u2 = (z' >> 11) / 2^53, cosine branch first, sine branch kept for the next draw."""
    import math
    M64 = (1 << 64) - 1

    class Stream:
        def __init__(self, seed):
            self.s, self.spare = seed & M64, None

        def u64(self):
            self.s = (self.s + 0x9E3779B97F4A7C15) & M64
            z = self.s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
            return z ^ (z >> 31)

        def normal(self):
            if self.spare is not None:
                v, self.spare = self.spare, None
                return v
            u1 = float((self.u64() >> 11) + 1) * (1.0 / 9007199254740992.0)
            u2 = float(self.u64() >> 11) * (1.0 / 9007199254740992.0)
            r, a = math.sqrt(-2.0 * math.log(u1)), 6.283185307179586476925286766559 * u2
            self.spare = r * math.sin(a)
            return r * math.cos(a)

    def index(x, y):
        return y * W + (W - 1 - x if y % 2 else x)

    truth = {}
    for y in range(H):
        for x in range(W):
            truth[index(x, y)] = (float(x), float(y), math.pi if y % 2 else 0.0)
    init = Stream(seed_init)
    state = []
    for k in range(W * H):
        gx, gy, gt = truth[k]
        state += [gx + 0.1 * init.normal(), gy + 0.1 * init.normal(), gt + 0.02 * init.normal()]
    meas_rng = Stream(seed_meas)
    ef, et, em = [], [], []

    def edge(a, b):
        i, j = min(a, b), max(a, b)
        (xi, yi, ti), (xj, yj, tj) = truth[i], truth[j]
        c, s_ = math.cos(ti), math.sin(ti)
        dx, dy = xj - xi, yj - yi
        ef.append(i)
        et.append(j)
        em.extend([c * dx + s_ * dy + 0.05 * meas_rng.normal(), -s_ * dx + c * dy + 0.05 * meas_rng.normal(),
                   (tj - ti) + 0.01 * meas_rng.normal()])

    def full():
        return n_edges_target > 0 and len(ef) >= n_edges_target

    stencil = [(0, 1), (1, 0), (-1, 1), (1, 1), (0, 2), (2, 0), (-2, 1), (-1, 2), (1, 2), (2, 1)]
    for y in range(H):
        for x in range(W):
            for ox, oy in stencil:
                if not full() and 0 <= x + ox < W and y + oy < H:
                    edge(index(x, y), index(x + ox, y + oy))
    if n_edges_target > 0:
        for y in range(H - 2):
            for x in range(2, W):
                if not full():
                    edge(index(x, y), index(x - 2, y + 2))
    return np.array(state), np.array(ef, np.int32), np.array(et, np.int32), np.array(em)


@pytest.mark.parametrize("extra", [0, 100])
def test_synthetic_grid_matches_an_independent_restatement_of_the_survey(lib, extra):
    """The config-4 generator (rr_pgo_synth_grid) bit for bit against a Python restatement of SURVEY.md 8(d): node
    states, edge lists and measurements -- the golden lattice fixtures rest on this stream."""
    from rustrobotics_amd import synthetic_grid_arrays
    W, H = 40, 25
    stencil_edges = sum((W - abs(dx)) * (H - dy) for dx, dy in
                        [(0, 1), (1, 0), (-1, 1), (1, 1), (0, 2), (2, 0), (-2, 1), (-1, 2), (1, 2), (2, 1)])
    target = stencil_edges + extra if extra else 0
    nk, ns, ek, ef, et, em, ei = synthetic_grid_arrays(W, H, target)
    rs, rf, rt, rm = _restated_grid(W, H, target)
    assert len(ef) == (target or stencil_edges)
    assert np.array_equal(ef, rf) and np.array_equal(et, rt)
    assert np.array_equal(ns, rs)            # bit for bit
    assert np.array_equal(em, rm)
    assert (nk == 0).all() and (ek == 0).all()
    assert np.array_equal(ei.reshape(-1, 6), np.tile([400.0, 0, 0, 400.0, 0, 10000.0], (len(ef), 1)))


@pytest.fixture(scope="session")
def mfcheck(tmp_path_factory):
    exe = tmp_path_factory.mktemp("native") / "mf_host_check"
    srcs = [os.path.join(ROOT, "tests", "native", "mf_host_check.cpp")] + [
        os.path.join(CSRC, f) for f in ("symbolic.cpp", "g2o_loader.cpp", "synth_grid.cpp")]
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", CSRC, *srcs, "-o", str(exe)])
    return str(exe)


@pytest.mark.parametrize("args,env", [
    ([g2o_path("simulation-pose-landmark")], {"LEAF": "1000000"}),   # mixed 3/2-dim blocks, pure min degree
    ([g2o_path("intel")], {"LEAF": "1000000"}),
    ([g2o_path("intel")], {"LEAF": "40"}),                          # nested dissection + big fronts
    ([g2o_path("dlr")], {"LEAF": "1000000"}),
    ([g2o_path("input_M3500_g2o")], {"LEAF": "64", "LDS": "38000"}),
    (["grid", "40", "25"], {"LEAF": "64"}),
    ([g2o_path("intel")], {"LEAF": "1000000", "FLOW": "1"}),        # the dataflow schedule (lds_flow.hip.h): ticket order
    ([g2o_path("dlr")], {"LEAF": "1000000", "FLOW": "1"}),
    ([g2o_path("input_M3500_g2o")], {"LEAF": "1000000", "LDS": "19000", "FLOW": "1"}),
    ([g2o_path("simulation-pose-landmark")], {"LEAF": "1000000", "FLOW": "1"}),
    (["grid", "100", "100"], {"LEAF": "48", "LDS": "38000", "FLOW": "1"}),            # dataflow step + levels of fronts beyond LDS
    ([g2o_path("sphere2500")], {"LEAF": "1000", "LDS": "19000", "FLOW": "1"}),
    ([g2o_path("intel")], {"LEAF": "40", "FLOW": "1", "FLOW_OPTIONAL": "1"}),         # a front beyond LDS under an LDS parent: level schedule
    ([g2o_path("intel")], {"LEAF": "1000000", "FLOW": "1", "MERGE_NC": "0", "BALANCE": "0"}),      # neither chain pass (the r03 trees)
    ([g2o_path("input_M3500_g2o")], {"LEAF": "3000", "FLOW": "1", "MERGE_NC": "200", "MERGE_GAIN": "-50", "BALANCE": "15", "AMALG_NP": "16"}),   # both, far beyond their defaults
    ([g2o_path("dlr")], {"LEAF": "1000000", "FLOW": "1", "MERGE_NC": "96", "MERGE_GAIN": "-5", "BALANCE": "12"}),   # 2- and 3-dim nodes moving between fronts
    (["grid", "60", "40"], {"LEAF": "64", "MERGE_NC": "64", "MERGE_GAIN": "-5", "BALANCE": "15"}),                  # ... under fronts beyond LDS
    (["grid", "60", "40"], {"LEAF": "64", "PARTS": "2"}),           # rank-owned subtrees + shared top
    (["grid", "60", "40"], {"LEAF": "64", "PARTS": "4"}),
    (["grid", "100", "100"], {"LEAF": "64", "PARTS": "8"}),
    # r05: the multilevel bisection with a minimum-cover separator (the engine's dissection of graphs of up to 6000 nodes)
    # (CHAIN_MAX: the longest chain of the tree in pivot columns -- 534 / 739 / 807 with minimum degree alone, 360 / 399 / 393 with
    # the bisection: a guard on the quality of the separators, which is what the iteration time of these graphs follows)
    ([g2o_path("intel")], {"LEAF": "50", "ML": "1", "FLOW": "1", "AMALG_NP": "32", "CHAIN_MAX": "400"}),   # the tree the engine picks
    ([g2o_path("dlr")], {"LEAF": "50", "ML": "1", "FLOW": "1", "AMALG_NP": "16", "CHAIN_MAX": "440"}),     # poses + landmarks
    ([g2o_path("input_M3500_g2o")], {"LEAF": "100", "ML": "1", "FLOW": "1", "AMALG_NP": "16", "CHAIN_MAX": "440"}),
    ([g2o_path("sphere2500")], {"LEAF": "50", "ML": "1", "LDS": "19000", "FLOW": "1", "AMALG_NP": "72"}),   # 6 x 6 blocks, fronts beyond LDS
    ([g2o_path("simulation-pose-landmark")], {"LEAF": "8", "ML": "1"}),                             # a graph smaller than the coarsening stops at
    ([g2o_path("sphere2500")], {"LEAF": "32", "ML": "1", "LDS": "19000", "PARTS": "8", "PIN": "1"}),      # ... sharded: the shared schedule is the same on every rank
    ([g2o_path("intel")], {"LEAF": "32", "ML": "1", "PARTS": "4", "PIN": "1"}),
])
def test_symbolic_tables_drive_a_correct_factorization(mfcheck, args, env):
    out = subprocess.run([mfcheck, *args], env={**os.environ, **env}, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr


@pytest.mark.parametrize("shape", ["star", "chain", "two-components", "dense", "hubs", "random-loops", "three-nodes"])
def test_multilevel_dissection_on_awkward_graphs(mfcheck, tmp_path, shape):
    """The multilevel bisection on graphs that are not trajectories: a star (every separator is the hub), a bare chain, two components,
    a dense random graph (no small separator exists), a chain with three hubs, a chain with random long loop closures, three nodes --
    unsharded and over four ranks the tables must still drive a correct factorisation (tests/native/mf_host_check.cpp)."""
    import random
    rnd = random.Random(5)
    n = 400
    chain = [(i, i + 1) for i in range(n - 1)]
    if shape == "star":
        edges = [(0, i) for i in range(1, n)]
    elif shape == "chain":
        edges = chain
    elif shape == "two-components":
        edges = [(i, i + 1) for i in range(n // 2 - 1)] + [(i, i + 1) for i in range(n // 2, n - 1)]
    elif shape == "dense":
        n = 90
        edges = [(i, j) for i in range(n) for j in range(i + 1, n) if rnd.random() < 0.5]
    elif shape == "hubs":
        edges = chain + [(h, i) for h in (5, 200, 390) for i in range(0, n, 3) if abs(i - h) > 1]
    elif shape == "random-loops":
        edges = list(chain)
        while len(edges) < 1000:
            a, b = rnd.randrange(n), rnd.randrange(n)
            if abs(a - b) > 1:
                edges.append((a, b))
    else:
        n, edges = 3, [(0, 1), (1, 2)]
    path = tmp_path / (shape + ".g2o")
    with open(path, "w") as f:
        for i in range(n):
            f.write("VERTEX_SE2 %d %.4f %.4f %.4f\n" % (i, rnd.random() * 10, rnd.random() * 10, rnd.random()))
        for a, b in edges:
            f.write("EDGE_SE2 %d %d 1.0 0.1 0.01 100 0 0 100 0 400\n" % (a, b))
    for env in ({"LEAF": "8", "ML": "1"}, {"LEAF": "16", "ML": "1", "PARTS": "4", "PIN": "1"}):
        out = subprocess.run([mfcheck, str(path)], env={**os.environ, **env}, capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.strip().endswith("OK"), (env, out.stdout + out.stderr)


def test_replayed_dissections_are_the_computed_ones(tmp_path):
    """The candidates of one graph share ONE nested dissection (recorded by the deepest, replayed by the others: symbolic.h NdSplitTable,
    pgo_api.hip): at every depth the engine tries, the analysis that replays must equal the analysis that computes -- ordering,
    supernodes, tree, schedule, estimate (tests/native/nd_replay_check.cpp)."""
    exe = tmp_path / "nd_replay_check"
    srcs = [os.path.join(ROOT, "tests", "native", "nd_replay_check.cpp")] + [os.path.join(CSRC, f) for f in ("symbolic.cpp", "g2o_loader.cpp", "synth_grid.cpp")]
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", CSRC, *srcs, "-lpthread", "-o", str(exe)])
    out = subprocess.run([str(exe), g2o_path("intel"), g2o_path("dlr"), g2o_path("sphere2500"), g2o_path("simulation-pose-landmark")], capture_output=True, text=True)
    assert out.returncode == 0 and "replayed == computed for every depth" in out.stdout, out.stdout + out.stderr


def test_algorithmic_bytes_follow_survey_8d(lib):
    """SURVEY.md 8(d), worked totals for intel.g2o in fp64: linearise 0.90 MB, solve 0.43 + 3 x 1.24 + 0.17 = 4.3 MB with
    nnzblk(L) = 17 193 [probe], update 0.12 MB -- 5.4 MB per iteration with chi2 fused into the linearisation.  The
    statistics bench.py's roofline divides by must be THAT figure (every datum moved once, the factor at its nonzeros), not
    the padded storage of the supernodal panels (r03 reported 7.17 MB).  r05: the multilevel dissection trades fill for a shorter
    critical path -- 18 985 blocks in L against 17 767 for the minimum-degree tree of r01 - r04 and the probe's 17 193 -- so the
    factor / solve figures sit 4 - 7 % above the survey's; the bounds below still exclude padded storage by a wide margin."""
    from rustrobotics_amd import PoseGraph
    s = PoseGraph.analyze(g2o_path("intel"))
    assert s["abi_version"] == 4
    assert s["n_supernodes"] > 0 and s["n_big_fronts"] == 0 and s["n_launches_per_iter"] == 0   # (no device: no engine)
    total = s["bytes_linearize"] + s["bytes_chi2"] + s["bytes_factor"] + s["bytes_solve"] + s["bytes_update"]
    assert abs(total - 5.4e6) <= 0.08 * 5.4e6, total
    assert abs(s["bytes_linearize"] - 0.90e6) <= 0.06 * 0.90e6
    assert abs(s["bytes_factor"] + s["bytes_solve"] - 4.3e6) <= 0.09 * 4.3e6
    assert abs(s["bytes_update"] - 0.12e6) <= 0.05 * 0.12e6
    assert s["stored_factor_bytes"] > s["bytes_factor"] - 0.45e6   # padding only ever adds
    # fp32 halves every scalar but not the 8 index bytes per edge
    s32 = PoseGraph.analyze(g2o_path("intel"), precision="f32")
    assert 0.5 * total < s32["bytes_linearize"] + s32["bytes_factor"] + s32["bytes_solve"] + s32["bytes_update"] < 0.52 * total


def test_offline_pmc_traffic_names_kernels_of_the_newest_kernel_stats():
    """bench.py copies roofline.traffic from profiles/pmc_traffic.json (rocprofv3 --pmc passes taken offline).  That file must
    belong to the same evidence round as the newest kernel statistics in profiles/ and name kernels that ran there -- otherwise
    the traffic ratio in the bench line silently describes kernels that no longer exist."""
    import csv
    import glob
    import json
    import re
    prof = os.path.join(ROOT, "profiles")
    doc = json.load(open(os.path.join(prof, "pmc_traffic.json")))
    tags = sorted({re.match(r"(r\d+z)_kernel_stats_", os.path.basename(f)).group(1) for f in glob.glob(os.path.join(prof, "r*z_kernel_stats_*.csv"))})
    newest = tags[-1]
    assert doc["_tag"] == newest, "pmc_traffic.json is from %s, the newest kernel statistics from %s: re-take the PMC passes (scripts/gpu_pmc.sh)" % (doc["_tag"], newest)
    stats = {"intel": "intel_f64", "grid": "grid_f32", "m3500": "m3500_f64", "sphere2500": "sphere2500_f64"}
    assert {"intel:f64", "m3500:f64", "sphere2500:f64", "grid:400x250:1000000:f32"} <= set(doc)   # no BASELINE config without traffic
    for key, e in doc.items():
        if key.startswith("_"):
            continue
        rows = list(csv.DictReader(open(os.path.join(prof, "%s_kernel_stats_%s.csv" % (newest, stats[key.split(":")[0]])))))
        names = {re.sub(r"<.*", "", r["Name"].replace("void ", "").replace("rrpgo::", "")) for r in rows}
        if key.endswith(":classes"):
            # the attribution of a whole iteration's traffic: every kernel it names ran, and the shares add up
            kernels = {k: v for k, v in e.items() if not k.startswith("_")}
            # (k_copy_words16: the restart of the profiling script's repetitions, not part of an iteration of the bench's statistics)
            assert set(kernels) - {"k_copy_words16"} <= names, (key, sorted(set(kernels) - names))
            assert e["_total_bytes_per_step"] == sum(v["traffic_bytes_per_step"] for v in kernels.values())
            continue
        ran = [k for k in e["kernel"].split("+") if k in names]
        assert ran, (key, e["kernel"], sorted(names))
        # the counters' view of the dominant kernel agrees with the byte model to within what a cache hierarchy can do
        assert e["traffic_bytes_per_launch"] > 0 and e["launches_sampled"] > 0


def test_the_sharded_c_client_compiles_against_the_header(tmp_path):
    """tests/native/shard_client.c (the torch-free driver of the sharded protocol; run on a GPU by tests/test_gpu_parity.py) builds
    with -Werror against include/rr_pgo.h and links the library alone: RCCL is found with dlopen at run time."""
    import shutil
    import subprocess
    from rustrobotics_amd import _lib
    exe = tmp_path / "shard_client"
    subprocess.check_call([shutil.which("gcc") or "gcc", "-std=c99", "-D_DEFAULT_SOURCE", "-Wall", "-Wextra", "-Werror",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "shard_client.c"),
                           _lib.LIB_PATH, "-ldl", f"-Wl,-rpath,{os.path.dirname(_lib.LIB_PATH)}", "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
