"""ctypes binding of libfdjac (include/fdjac.h) -- the same stub a Julia `ccall` shim uses.

The library is built in-tree by ``csrc/Makefile`` (``hipcc --offload-arch=gfx950``) into
``finitediff.jl_amd/lib/libfdjac.so``.  There is no fallback: if the shared object is missing
``load()`` raises, and every compute entry point needs a GPU.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "lib", "libfdjac.so")
CSRC = os.path.join(_HERE, "csrc")

FD_OK = 0
FD_ERR_COMM = 8
COMM_ID_BYTES = 128
P2P_HANDLE_BYTES = 64
EPS_COMPUTE, EPS_PRECOMPUTED = 0, 1
TRI_DIAGONALS, TRI_CSC = 0, 1
FORWARD, CENTRAL, COMPLEX = 0, 1, 2
HOST, DEVICE = 0, 1
FDTYPES = {"forward": FORWARD, "central": CENTRAL, "complex": COMPLEX}
STAGES = ("eps", "perturb", "f", "decompress", "total", "exchange")
(INFO_M, INFO_N, INFO_NCOLORS, INFO_NOUTS, INFO_OUT0_LEN, INFO_OUT1_LEN, INFO_OUT2_LEN, INFO_ROW_BEGIN,
 INFO_ROW_END, INFO_NCHUNKS, INFO_SCRATCH_BYTES, INFO_NNZ_LOCAL, INFO_FCALLS_LAST, INFO_ENTRY_BEGIN,
 INFO_SORTED_GATHER, INFO_LINES_DIRECT_X100, INFO_LINES_SORTED_X100, INFO_WINDOW,
 INFO_WIN_OVERREAD_X100, INFO_WINDOW2D, INFO_WIN_PERIOD, INFO_COLRANGE_WG, INFO_SMALL_FUSED, _INFO_23,
 INFO_EPS_CYCLIC, INFO_EPS_NT, _INFO_26, INFO_BUILT_ON_DEVICE, _INFO_28, INFO_LAZY_DIFF, _INFO_30, INFO_BAND_DESC, INFO_LAZY_STORE, INFO_STORE_CSC) = range(34)
LAZY_CAP_IMAG_ONLY, LAZY_CAP_ROW_WINDOW, LAZY_CAP_DIFF, LAZY_CAP_STORE, LAZY_CAP_STORE_CSC, LAZY_CAP_STORE_CSC_BASE, LAZY_CAP_STORE_CSC_COMPLEX, LAZY_CAP_FUSED_EPS, LAZY_CAP_STORE_COLRANGE = 1, 2, 4, 8, 16, 32, 64, 128, 256
PLAN_EPS_CONTIGUOUS, PLAN_COMPLEX_X, PLAN_FINGERPRINT, PLAN_STORE_CSC, PLAN_STORE_CSC_ALWAYS, PLAN_STORE_CSC_ROWS = 1, 2, 4, 8, 16, 32
LAZY_JVP_CAP_QUOTIENT = 1
(F_TRIDIAG, F_TRIDIAG_NL, F_LAP5, F_CLAMP5, F_BLOCKCOUPLED, F_NONSQUARE, F_LAP5_NL, F_LAP7, F_SPARSE) = range(9)
FAMILIES = {"tridiag": F_TRIDIAG, "tridiag_nl": F_TRIDIAG_NL, "lap5": F_LAP5, "clamp5": F_CLAMP5,
            "blockcoupled": F_BLOCKCOUPLED, "nonsquare": F_NONSQUARE, "lap5_nl": F_LAP5_NL, "lap7": F_LAP7}

# int f(fctx, fx, x, nbatch, x_stride, fx_stride, row_begin, row_end, is_complex, stream)
F_LAUNCH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                       C.c_int64, C.c_int, C.c_void_p)

class LazyPoints(C.Structure):
    _fields_ = [("x", C.c_void_p), ("color", C.c_void_p), ("eps", C.c_void_p), ("base_out", C.c_void_p),
                ("color_bytes", C.c_int32), ("c_lo", C.c_int32), ("ncolors", C.c_int32), ("pts", C.c_int32),
                ("is_complex", C.c_int32), ("imag_only", C.c_int32), ("part", C.c_int32), ("nparts", C.c_int32),
                ("diff", C.c_int32), ("store_kind", C.c_int32), ("store", C.c_void_p), ("eps_job", C.c_void_p)]


# int f(fctx, fx, const fd_lazy_points*, fx_stride, row_begin, row_end, stream)
F_LAUNCH_LAZY = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(LazyPoints), C.c_int64, C.c_int64, C.c_int64,
                            C.c_void_p)

class LazyJvpPoints(C.Structure):
    """fd_lazy_jvp_points (include/fdjac.h)."""
    _fields_ = [("x", C.c_void_p), ("v", C.c_void_p), ("eps", C.c_void_p), ("base_out", C.c_void_p), ("central", C.c_int),
                ("reserved0", C.c_int), ("quotient_out", C.c_void_p)]


# int f(fctx, fx, const fd_lazy_jvp_points*, fx_stride, stream)
F_LAUNCH_LAZY_JVP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(LazyJvpPoints), C.c_int64, C.c_void_p)

EXPORTS = (
    "fd_version", "fd_last_error", "fd_ctx_create", "fd_ctx_destroy", "fd_ctx_stream", "fd_ctx_synchronize",
    "fd_plan_create_csc", "fd_plan_create_csc_dense", "fd_plan_create_coo_dense", "fd_plan_create_entries",
    "fd_plan_create_dense", "fd_plan_create_tridiagonal", "fd_plan_create_banded", "fd_plan_create_blockbanded",
    "fd_plan_create_bandedblockbanded", "fd_plan_destroy",
    "fd_plan_info", "fd_jacobian", "fd_jacobian_async", "fd_jacobian_owned_async", "fd_plan_get_epsilons", "fd_plan_fused_trace", "fd_plan_enable_timing", "fd_plan_set_timing_stride",
    "fd_plan_get_timings", "fd_builtin_f_create", "fd_builtin_f_destroy", "fd_builtin_f_counts", "fd_builtin_f_info",
    "fd_stream_copy_gbps", "fd_plan_set_lazy_f", "fd_builtin_f_lazy", "fd_plan_set_lazy_caps", "fd_builtin_f_lazy_caps",
    "fd_jvp_plan_create", "fd_jvp_plan_destroy", "fd_jvp", "fd_jvp_async", "fd_jvp_get_epsilon",
    "fd_jvp_plan_set_lazy_f", "fd_builtin_f_lazy_jvp", "fd_jvp_plan_set_lazy_caps", "fd_builtin_f_lazy_jvp_caps",
    "fd_color_columns_greedy", "fd_color_banded",
    "fd_comm_unique_id", "fd_comm_create", "fd_comm_destroy", "fd_comm_info", "fd_comm_library", "fd_comm_allgather",
    "fd_comm_gatherv", "fd_comm_allreduce_sum", "fd_comm_broadcast", "fd_comm_halo_exchange", "fd_comm_enable_p2p", "fd_comm_p2p_status", "fd_comm_disable_p2p", "fd_f_compile_rows", "fd_f_link_rows_bitcode", "fd_f_compiled_destroy", "fd_f_compiled_counts", "fd_f_compile_log", "fd_f_compile_terms", "fd_f_compiled_row_stores", "fd_plan_row_lists",
    "fd_p2p_create", "fd_p2p_create_loopback", "fd_p2p_loopback_fill", "fd_p2p_loopback_fill_fused", "fd_p2p_local_handle", "fd_p2p_connect", "fd_p2p_destroy", "fd_p2p_info", "fd_p2p_status", "fd_p2p_allgather",
    "fd_p2p_halo_exchange",
    "fd_plan_set_comm", "fd_plan_set_p2p", "fd_plan_set_halo", "fd_plan_eps_partials", "fd_plan_eps_finalize", "fd_plan_set_eps_mode",
    "fd_tridiag_solver_create", "fd_tridiag_solver_destroy", "fd_tridiag_solve_async", "fd_tridiag_solve_interface",
    "fd_tridiag_solve_finish", "fd_plan_create_csc_device", "fd_plan_checksum", "fd_plan_get_timing_samples",
    "fd_plan_eps_shard_range", "fd_plan_matches", "fd_plan_matches_async", "fd_plan_stale", "fd_builtin_f_create_sparse", "fd_tridiag_solver_status", "fd_tridiag_solver_set_policy",
    "fd_banded_solver_create", "fd_banded_solver_destroy", "fd_banded_solver_set_policy", "fd_banded_solver_status", "fd_banded_solve_async",
    "fd_blocktridiag_solver_create", "fd_blocktridiag_solver_destroy", "fd_blocktridiag_solver_set_policy", "fd_blocktridiag_solver_status",
    "fd_blocktridiag_solve_async",
)


# functions that exist once per element type (fd_* = Float64, fd32_* = Float32); everything else is shared
TYPED = (
    "fd_plan_create_csc", "fd_plan_create_csc_dense", "fd_plan_create_coo_dense", "fd_plan_create_entries",
    "fd_plan_create_dense", "fd_plan_create_tridiagonal", "fd_plan_create_banded", "fd_plan_create_blockbanded",
    "fd_plan_create_bandedblockbanded", "fd_plan_destroy", "fd_plan_info", "fd_plan_row_lists", "fd_jacobian", "fd_jacobian_async", "fd_jacobian_owned_async", "fd_plan_set_lazy_f", "fd_plan_set_lazy_caps",
    "fd_plan_get_epsilons", "fd_plan_fused_trace", "fd_plan_enable_timing", "fd_plan_set_timing_stride", "fd_plan_get_timings", "fd_builtin_f_create", "fd_builtin_f_destroy",
    "fd_builtin_f_counts", "fd_builtin_f_info", "fd_builtin_f_lazy", "fd_builtin_f_lazy_caps", "fd_jvp_plan_create", "fd_jvp_plan_destroy",
    "fd_jvp", "fd_jvp_async", "fd_jvp_get_epsilon", "fd_jvp_plan_set_lazy_f", "fd_builtin_f_lazy_jvp",
    "fd_jvp_plan_set_lazy_caps", "fd_builtin_f_lazy_jvp_caps",
    "fd_plan_set_comm", "fd_plan_set_p2p", "fd_plan_set_halo", "fd_plan_eps_partials", "fd_plan_eps_finalize", "fd_plan_set_eps_mode",
    "fd_tridiag_solver_create", "fd_tridiag_solver_destroy", "fd_tridiag_solve_async", "fd_tridiag_solve_interface",
    "fd_tridiag_solve_finish", "fd_plan_create_csc_device", "fd_plan_checksum", "fd_plan_get_timing_samples",
    "fd_plan_eps_shard_range", "fd_plan_matches", "fd_plan_matches_async", "fd_plan_stale", "fd_builtin_f_create_sparse", "fd_tridiag_solver_status", "fd_tridiag_solver_set_policy",
    "fd_banded_solver_create", "fd_banded_solver_destroy", "fd_banded_solver_set_policy", "fd_banded_solver_status", "fd_banded_solve_async",
    "fd_blocktridiag_solver_create", "fd_blocktridiag_solver_destroy", "fd_blocktridiag_solver_set_policy", "fd_blocktridiag_solver_status",
    "fd_blocktridiag_solve_async",
)
EXPORTS = EXPORTS + tuple("fd32_" + n[3:] for n in TYPED)


class TypedLib:
    """View of the loaded library for one element type: ``typed(L, np.float32).fd_jacobian`` is ``L.fd32_jacobian``."""

    def __init__(self, L, f32):
        self._L, self._f32 = L, bool(f32)

    def __getattr__(self, name):
        if self._f32 and name in TYPED:
            return getattr(self._L, "fd32_" + name[3:])
        return getattr(self._L, name)


def typed(L, dtype):
    import numpy as np
    return TypedLib(L, np.dtype(dtype) == np.float32)


class PlanOpts(C.Structure):
    _fields_ = [("fdtype", C.c_int32), ("flags", C.c_int32), ("col_begin", C.c_int64), ("col_end", C.c_int64),
                ("x_begin", C.c_int64), ("x_end", C.c_int64), ("scratch_bytes", C.c_int64),
                ("color_begin", C.c_int64), ("color_end", C.c_int64)]


class PatternArrays(C.Structure):
    """fd_pattern_arrays (include/fdjac.h): the arrays fd_plan_matches compares with what a plan was compiled from."""
    _fields_ = [("idx_a", C.c_void_p), ("len_a", C.c_int64), ("idx_b", C.c_void_p), ("len_b", C.c_int64),
                ("colorvec", C.c_void_p), ("len_color", C.c_int64), ("idx_bytes", C.c_int32), ("idx_base", C.c_int32),
                ("color_bytes", C.c_int32), ("memkind", C.c_int32)]


class FdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfdjac error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile the HIP sources for gfx950 (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s", "-j4"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(args)
    return SO_PATH


_lib = None


def load():
    """dlopen libfdjac.so and declare the prototypes.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise FileNotFoundError(
            "%s is missing: build it with `make -C %s` (hipcc, gfx950). There is no CPU fallback." % (SO_PATH, CSRC))
    try:  # share torch's HIP runtime (same SONAME) so device pointers are interchangeable
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the raw ABI
        pass
    L = C.CDLL(SO_PATH)
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
    pp = C.POINTER(C.c_void_p)
    L.fd_version.restype = i32
    L.fd_last_error.restype = C.c_char_p
    L.fd_ctx_create.argtypes = [i32, vp, pp]
    L.fd_ctx_destroy.argtypes = [vp]
    L.fd_ctx_stream.argtypes = [vp]
    L.fd_ctx_stream.restype = vp
    L.fd_ctx_synchronize.argtypes = [vp]
    po = C.POINTER(PlanOpts)
    L.fd_plan_create_csc.argtypes = [vp, i64, i64, vp, vp, i32, i32, vp, i32, po, pp]
    L.fd_plan_create_csc_device.argtypes = [vp, i64, i64, vp, vp, i32, i32, vp, i32, po, pp]
    L.fd_plan_checksum.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.fd_plan_create_csc_dense.argtypes = [vp, i64, i64, vp, vp, i32, i32, vp, i32, po, pp]
    L.fd_plan_create_coo_dense.argtypes = [vp, i64, i64, vp, vp, i64, i32, i32, vp, i32, po, pp]
    L.fd_plan_create_entries.argtypes = [vp, i64, i64, vp, vp, vp, i64, i64, i32, i32, vp, i32, po, pp]
    L.fd_plan_create_tridiagonal.argtypes = [vp, i64, vp, i32, po, pp]
    L.fd_plan_create_dense.argtypes = [vp, i64, i64, i64, po, pp]
    L.fd_plan_create_banded.argtypes = [vp, i64, i64, i64, i64, vp, i32, po, pp]
    L.fd_plan_create_blockbanded.argtypes = [vp, i64, vp, i64, i64, vp, vp, i32, i32, vp, i32, po, pp]
    L.fd_plan_create_bandedblockbanded.argtypes = [vp, i64, vp, i64, i64, i64, i64, vp, vp, i64, i32, i32, vp, i32, po, pp]
    L.fd_plan_destroy.argtypes = [vp]
    L.fd_plan_matches.argtypes = [vp, C.POINTER(PatternArrays), C.POINTER(i32)]
    L.fd_plan_matches_async.argtypes = [vp, C.POINTER(PatternArrays)]
    L.fd_plan_stale.argtypes = [vp, C.POINTER(i32)]
    L.fd_plan_info.argtypes = [vp, i32, C.POINTER(i64)]
    L.fd_plan_row_lists.argtypes = [vp, pp, pp, pp, C.POINTER(i64), C.POINTER(C.c_uint64)]
    L.fd_jacobian.argtypes = [vp, F_LAUNCH, vp, vp, i32, vp, i32, dbl, dbl, dbl, pp, i32]
    L.fd_jacobian_async.argtypes = [vp, F_LAUNCH, vp, vp, vp, dbl, dbl, dbl, pp]
    L.fd_jacobian_owned_async.argtypes = [vp, vp, F_LAUNCH, vp, vp, vp, dbl, dbl, dbl, pp]
    L.fd_plan_get_epsilons.argtypes = [vp, C.POINTER(dbl)]
    L.fd_plan_fused_trace.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.fd_plan_enable_timing.argtypes = [vp, i32]
    L.fd_plan_set_timing_stride.argtypes = [vp, i32]
    L.fd_plan_get_timings.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64)]
    L.fd_plan_get_timing_samples.argtypes = [vp, i32, C.POINTER(dbl), i64, C.POINTER(i64)]
    L.fd_builtin_f_create.argtypes = [vp, i32, C.POINTER(i64), i32, C.POINTER(F_LAUNCH), pp]
    L.fd_builtin_f_destroy.argtypes = [vp]
    L.fd_builtin_f_create_sparse.argtypes = [vp, i64, i64, vp, vp, i32, i32, C.POINTER(F_LAUNCH), pp]
    L.fd_builtin_f_counts.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.fd_builtin_f_info.argtypes = [vp, C.c_int, C.POINTER(i64)]
    L.fd_stream_copy_gbps.argtypes = [vp, i64, i32, C.POINTER(dbl)]
    L.fd_jvp_plan_create.argtypes = [vp, i64, i64, i32, pp]
    L.fd_jvp_plan_destroy.argtypes = [vp]
    L.fd_jvp.argtypes = [vp, F_LAUNCH, vp, vp, vp, i32, vp, i32, dbl, dbl, dbl, vp, i32]
    L.fd_jvp_async.argtypes = [vp, F_LAUNCH, vp, vp, vp, vp, dbl, dbl, dbl, vp]
    L.fd_jvp_get_epsilon.argtypes = [vp, C.POINTER(dbl)]
    L.fd_color_columns_greedy.argtypes = [i64, i64, vp, vp, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    L.fd_color_banded.argtypes = [i64, i64, i64, C.POINTER(i64), C.POINTER(i64)]
    L.fd_plan_set_lazy_f.argtypes = [vp, F_LAUNCH_LAZY]
    L.fd_builtin_f_lazy.argtypes = [vp, C.POINTER(F_LAUNCH_LAZY)]
    L.fd_plan_set_lazy_caps.argtypes = [vp, i32]
    L.fd_builtin_f_lazy_caps.argtypes = [vp, C.POINTER(i32)]
    L.fd_jvp_plan_set_lazy_f.argtypes = [vp, F_LAUNCH_LAZY_JVP]
    L.fd_builtin_f_lazy_jvp.argtypes = [vp, C.POINTER(F_LAUNCH_LAZY_JVP)]
    L.fd_jvp_plan_set_lazy_caps.argtypes = [vp, i32]
    L.fd_builtin_f_lazy_jvp_caps.argtypes = [vp, C.POINTER(i32)]
    L.fd_comm_unique_id.argtypes = [vp]
    L.fd_comm_create.argtypes = [vp, i32, i32, vp, pp]
    L.fd_comm_destroy.argtypes = [vp]
    L.fd_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.fd_comm_library.restype = C.c_char_p
    L.fd_comm_allgather.argtypes = [vp, vp, i64, i32]
    L.fd_comm_gatherv.argtypes = [vp, vp, i64, vp, C.POINTER(i64), C.POINTER(i64), i32, i32]
    L.fd_comm_allreduce_sum.argtypes = [vp, vp, i64, i32]
    L.fd_comm_broadcast.argtypes = [vp, vp, i64, i32, i32]
    L.fd_comm_halo_exchange.argtypes = [vp, vp, i64, i64, i64, i32]
    L.fd_comm_enable_p2p.argtypes = [vp, i64]
    L.fd_comm_p2p_status.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.fd_comm_disable_p2p.argtypes = [vp]
    L.fd_f_compile_rows.argtypes = [vp, C.c_char_p, C.c_char_p, vp, i64, i64, i64, i32, C.POINTER(F_LAUNCH), C.POINTER(F_LAUNCH_LAZY), C.POINTER(i32), pp]
    L.fd_f_link_rows_bitcode.argtypes = [vp, vp, i64, vp, i64, i64, i64, i32, C.POINTER(F_LAUNCH), C.POINTER(F_LAUNCH_LAZY), C.POINTER(i32), pp]
    L.fd_f_compiled_destroy.argtypes = [vp]
    L.fd_f_compiled_counts.argtypes = [vp, C.POINTER(i64)]
    L.fd_f_compiled_row_stores.argtypes = [vp, C.POINTER(i64)]
    L.fd_f_compile_terms.argtypes = [vp, C.c_char_p, C.c_char_p, vp, i64, i64, i64, i32, vp, vp, C.c_uint64, C.POINTER(F_LAUNCH), C.POINTER(F_LAUNCH_LAZY), C.POINTER(i32), pp]
    L.fd_f_compile_log.restype = C.c_char_p
    L.fd_p2p_create.argtypes = [vp, i32, i32, i64, pp]
    L.fd_p2p_create_loopback.argtypes = [vp, i32, i32, i64, pp]
    L.fd_p2p_loopback_fill.argtypes = [vp, i32, i64, vp, i64]
    L.fd_p2p_loopback_fill_fused.argtypes = [vp, vp, vp, vp, i64]
    L.fd_p2p_local_handle.argtypes = [vp, vp]
    L.fd_p2p_connect.argtypes = [vp, vp]
    L.fd_p2p_destroy.argtypes = [vp]
    L.fd_p2p_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64), C.POINTER(i32)]
    L.fd_p2p_status.argtypes = [vp, C.POINTER(i32)]
    L.fd_p2p_allgather.argtypes = [vp, vp, i64]
    L.fd_p2p_halo_exchange.argtypes = [vp, vp, i64, i64, i64, i32]
    L.fd_plan_eps_shard_range.argtypes = [vp, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    L.fd_plan_set_comm.argtypes = [vp, vp]
    L.fd_plan_set_p2p.argtypes = [vp, vp]
    L.fd_plan_set_halo.argtypes = [vp, i64, i64, i64]
    L.fd_plan_eps_partials.argtypes = [vp, vp, i32, i32, pp, C.POINTER(i64)]
    L.fd_plan_eps_finalize.argtypes = [vp, dbl, dbl, dbl]
    L.fd_plan_set_eps_mode.argtypes = [vp, i32]
    L.fd_blocktridiag_solver_create.argtypes = [vp, i64, i32, pp]
    L.fd_blocktridiag_solver_destroy.argtypes = [vp]
    L.fd_blocktridiag_solver_set_policy.argtypes = [vp, i32]
    L.fd_blocktridiag_solver_status.argtypes = [vp, C.POINTER(i32)]
    L.fd_blocktridiag_solve_async.argtypes = [vp, dbl, dbl, vp, vp, vp]
    L.fd_banded_solver_create.argtypes = [vp, i64, i32, i32, i32, pp]
    L.fd_banded_solver_destroy.argtypes = [vp]
    L.fd_banded_solver_set_policy.argtypes = [vp, i32]
    L.fd_banded_solver_status.argtypes = [vp, C.POINTER(i32)]
    L.fd_banded_solve_async.argtypes = [vp, dbl, dbl, vp, vp, vp]
    L.fd_tridiag_solver_create.argtypes = [vp, i64, i64, i64, i32, pp]
    L.fd_tridiag_solver_destroy.argtypes = [vp]
    L.fd_tridiag_solver_status.argtypes = [vp, C.POINTER(i32)]
    L.fd_tridiag_solver_set_policy.argtypes = [vp, i32]
    L.fd_tridiag_solve_async.argtypes = [vp, dbl, dbl, pp, vp, vp, vp]
    L.fd_tridiag_solve_interface.argtypes = [vp, dbl, dbl, pp, vp, vp]
    L.fd_tridiag_solve_finish.argtypes = [vp, dbl, dbl, pp, vp, vp, i32, i32, vp]
    for name in TYPED:   # the Float32 instantiation has the same prototypes (values behind void*, steps stay double)
        getattr(L, "fd32_" + name[3:]).argtypes = getattr(L, name).argtypes
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("fd_last_error", "fd_ctx_stream", "fd_comm_library", "fd_f_compile_log"):
            fn.restype = i32
    _lib = L
    return L


def check(rc):
    if rc != FD_OK:
        raise FdError(rc, load().fd_last_error().decode("utf-8", "replace"))
