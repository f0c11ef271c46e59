"""Loader for the in-tree HIP library (rustrobotics_amd/librr_pgo.so, C ABI: include/rr_pgo.h).

There is no CPU fallback: if the library is missing this module raises, and every
compute entry point of the library itself fails with RR_PGO_ENODEVICE when no
HIP device is present.
"""
import ctypes as C
import glob
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librr_pgo.so")

OK, EINVAL, EIO, EPARSE, ENODEVICE, ENOTSPD, ENOMEM, EUNSUPPORTED, ETIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7, -8
ABI_VERSION = 4   # RR_PGO_ABI_VERSION this mirror was written against (load() checks the library's)
F64, F32, MIXED = 0, 1, 2
PRECISIONS = {"f64": F64, "f32": F32, "mixed": MIXED}
NUM_KCLASS = 11
KCLASS_NAMES = ("linearize", "factor", "solve", "update", "reduce", "big_assembly", "big_panel", "big_update",
                "mid_factor", "big_solve", "big_flow")

# every symbol include/rr_pgo.h declares (tests check that the .so exports exactly these)
EXPORTS = (
    "rr_pgo_default_options", "rr_pgo_load_g2o", "rr_pgo_create", "rr_pgo_destroy", "rr_pgo_last_error",
    "rr_pgo_num_nodes", "rr_pgo_num_edges", "rr_pgo_dim", "rr_pgo_state_len", "rr_pgo_anchor_node",
    "rr_pgo_get_graph", "rr_pgo_chi2", "rr_pgo_linearize_solve", "rr_pgo_update", "rr_pgo_optimize",
    "rr_pgo_get_state", "rr_pgo_set_state", "rr_pgo_assemble", "rr_pgo_iterate_async", "rr_pgo_sync",
    "rr_pgo_get_stats", "rr_pgo_analyze_g2o", "rr_pgo_abi_version", "rr_pgo_debug_withhold", "rr_pgo_profile", "rr_pgo_synth_grid", "rr_pgo_synth_free",
    "rr_pgo_exchange_buffer", "rr_pgo_set_exchange_buffer", "rr_pgo_stage", "rr_pgo_stage_scalars", "rr_pgo_stream",
    "rr_pgo_node_owner", "rr_pgo_trim",
)


class GraphDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("node_kind", C.POINTER(C.c_int32)), ("node_id", C.POINTER(C.c_uint32)),
        ("node_state", C.POINTER(C.c_double)), ("n_edges", C.c_int32), ("edge_kind", C.POINTER(C.c_int32)),
        ("edge_from", C.POINTER(C.c_int32)), ("edge_to", C.POINTER(C.c_int32)),
        ("edge_meas", C.POINTER(C.c_double)), ("edge_info", C.POINTER(C.c_double)),
    ]


class Options(C.Structure):
    _fields_ = [("precision", C.c_int32), ("device", C.c_int32), ("solver", C.c_int32), ("rank", C.c_int32),
                ("world_size", C.c_int32), ("sharded", C.c_int32), ("reserved", C.c_int32 * 10)]


class Stats(C.Structure):
    _fields_ = [
        ("nnz_h_blocks", C.c_int64), ("nnz_l_scalars", C.c_int64), ("factor_flops", C.c_int64),
        ("n_supernodes", C.c_int32), ("n_levels", C.c_int32), ("n_launches_per_iter", C.c_int32),
        ("max_front", C.c_int32), ("max_pivot_cols", C.c_int32), ("n_big_fronts", C.c_int32),
        ("analyze_ms", C.c_double), ("parse_ms", C.c_double),
        ("bytes_linearize", C.c_double), ("bytes_factor", C.c_double), ("bytes_solve", C.c_double),
        ("bytes_update", C.c_double), ("bytes_chi2", C.c_double), ("big_update_flops", C.c_double),
        ("big_flow_flops", C.c_double), ("stored_factor_bytes", C.c_double), ("abi_version", C.c_int32), ("lds_dataflow", C.c_int32),
    ]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). rustrobotics_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  A process that maps /opt/rocm's copy
    # first (through this library) and torch's later ends up with two HIP runtimes, and the second one finds no
    # device.  So the runtime torch WOULD load is mapped first, by path and without importing torch (an import costs
    # seconds; torch-free callers pay a dlopen): both then resolve to one runtime whichever order they come in.
    # Nothing to do when torch is already imported.  RR_PGO_NO_TORCH_PRELOAD=1 skips this.
    if "torch" not in sys.modules and not os.environ.get("RR_PGO_NO_TORCH_PRELOAD"):
        try:
            spec = importlib.util.find_spec("torch")
            libdir = os.path.join(os.path.dirname(spec.origin), "lib") if spec and spec.origin else None
            # the wheel may ship the runtime under its versioned name only (libamdhip64.so.6, ...)
            hips = sorted(glob.glob(os.path.join(libdir, "libamdhip64.so*"))) if libdir else []
            if hips:
                C.CDLL(hips[0], mode=C.RTLD_GLOBAL)
            elif libdir and os.path.isdir(libdir):
                import torch  # noqa: F401  -- a torch without a HIP runtime of that name: let torch map whatever it uses first
        except (ImportError, OSError, ValueError):
            pass
    L = C.CDLL(LIB_PATH)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.rr_pgo_default_options.argtypes = [C.POINTER(Options)]
    L.rr_pgo_default_options.restype = None
    L.rr_pgo_load_g2o.argtypes = [C.c_char_p, C.POINTER(Options), C.POINTER(vp)]
    L.rr_pgo_create.argtypes = [C.POINTER(GraphDesc), C.POINTER(Options), C.POINTER(vp)]
    L.rr_pgo_destroy.argtypes = [vp]
    L.rr_pgo_destroy.restype = None
    L.rr_pgo_last_error.restype = C.c_char_p
    for name in ("rr_pgo_num_nodes", "rr_pgo_num_edges", "rr_pgo_dim", "rr_pgo_state_len", "rr_pgo_anchor_node"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = C.c_int32
    L.rr_pgo_get_graph.argtypes = [vp, C.POINTER(GraphDesc)]
    L.rr_pgo_chi2.argtypes = [vp, dp]
    L.rr_pgo_linearize_solve.argtypes = [vp, C.c_double, C.c_int, dp]
    L.rr_pgo_update.argtypes = [vp, dp, C.c_double]
    L.rr_pgo_optimize.argtypes = [vp, C.c_int32, dp, ip, dp]
    L.rr_pgo_get_state.argtypes = [vp, dp]
    L.rr_pgo_set_state.argtypes = [vp, dp]
    L.rr_pgo_assemble.argtypes = [vp, C.c_double, C.c_int, ip, ip, ip, C.POINTER(C.c_int64), dp,
                                  C.POINTER(C.c_int64), dp]
    L.rr_pgo_iterate_async.argtypes = [vp, C.c_int32]
    L.rr_pgo_sync.argtypes = [vp]
    L.rr_pgo_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rr_pgo_profile.argtypes = [vp, C.c_int32, dp, C.POINTER(C.c_int64), C.c_int32]
    L.rr_pgo_synth_grid.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.c_uint64, C.c_uint64, C.POINTER(vp),
                                    C.POINTER(GraphDesc)]
    L.rr_pgo_synth_free.argtypes = [vp]
    L.rr_pgo_synth_free.restype = None
    L.rr_pgo_exchange_buffer.argtypes = [vp, C.c_int32, C.POINTER(vp), C.POINTER(C.c_int64), ip]
    L.rr_pgo_set_exchange_buffer.argtypes = [vp, C.c_int32, vp, C.c_int64]
    L.rr_pgo_stage.argtypes = [vp, C.c_int32, C.c_double, C.c_int]
    L.rr_pgo_stage_scalars.argtypes = [vp, dp, dp]
    L.rr_pgo_stream.argtypes = [vp]
    L.rr_pgo_stream.restype = vp
    L.rr_pgo_node_owner.argtypes = [vp, ip]
    L.rr_pgo_analyze_g2o.argtypes = [C.c_char_p, C.POINTER(Options), C.POINTER(Stats)]
    L.rr_pgo_abi_version.restype = C.c_int32
    L.rr_pgo_debug_withhold.argtypes = [vp, C.c_int32]
    if L.rr_pgo_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} speaks ABI version {L.rr_pgo_abi_version()}, this mirror {ABI_VERSION}: rebuild the library")
    _lib = L
    return L
