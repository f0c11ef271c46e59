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


def test_struct_layouts_match_header(lib):
    from rustrobotics_amd import _lib
    assert C.sizeof(_lib.Options) == 16 * 4
    assert C.sizeof(_lib.GraphDesc) == 80
    assert C.sizeof(_lib.Stats) == 3 * 8 + 6 * 4 + 8 * 8 + 6 * 4


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
    (["grid", "60", "40"], {"LEAF": "64", "PARTS": "2"}),           # rank-owned subtrees + shared top
    (["grid", "60", "40"], {"LEAF": "64", "PARTS": "4"}),
    (["grid", "100", "100"], {"LEAF": "64", "PARTS": "8"}),
])
def test_symbolic_tables_drive_a_correct_factorization(mfcheck, args, env):
    out = subprocess.run([mfcheck, *args], env={**os.environ, **env}, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
