"""ctypes front-end of the CPU oracle (oracle/fd_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.

Everything here speaks the reference's conventions: 1-based Int64 indices,
column-major matrices, colours 1..C.  See fd_oracle.c for the reference
file:line each routine follows.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libfd_oracle.so")
_SO32 = os.path.join(_HERE, "_build", "libfd_oracle32.so")   # the same source with element type float
_SO_OMP = os.path.join(_HERE, "_build", "libfd_oracle_omp.so")   # the same passes, loops split over the host's cores

FORWARD, CENTRAL, COMPLEX = 0, 1, 2
FDTYPES = {"forward": FORWARD, "central": CENTRAL, "complex": COMPLEX}

(PAT_NONE, PAT_CSC_COMMON, PAT_CSC_DENSEJ, PAT_COO_DENSEJ, PAT_COO_TRIDIAG, PAT_BANDED,
 PAT_BLOCKBANDED, PAT_BANDEDBLOCKBANDED, PAT_COO_DENSEJ_ACCUM) = range(9)

F_REAL = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))
F_CPLX = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)

_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)
_f32p = C.POINTER(C.c_float)


def _pattern_fields(fp):
    return [
        ("kind", C.c_int), ("M", C.c_int64), ("N", C.c_int64),
        ("colptr", _i64p), ("rowval", _i64p),
        ("rows_index", _i64p), ("cols_index", _i64p), ("ncoo", C.c_int64),
        ("l", C.c_int64), ("u", C.c_int64),
        ("nblk", C.c_int64), ("blk_sizes", _i64p), ("bl", C.c_int64), ("bu", C.c_int64),
        ("block_starts", _i64p), ("block_strides", _i64p),
        ("out0", fp), ("out1", fp), ("out2", fp), ("out_len", C.c_int64),
        ("lam", C.c_int64), ("mu", C.c_int64),
    ]


class Pattern(C.Structure):
    _fields_ = _pattern_fields(_f64p)


class Pattern32(C.Structure):
    _fields_ = _pattern_fields(_f32p)


class _T:
    """ctypes / numpy types of one element type."""

    def __init__(self, dtype):
        self.f32 = np.dtype(dtype) == np.float32
        self.real = np.float32 if self.f32 else np.float64
        self.cplx = np.complex64 if self.f32 else np.complex128
        self.c_real = C.c_float if self.f32 else C.c_double
        self.fp = _f32p if self.f32 else _f64p
        self.Pattern = Pattern32 if self.f32 else Pattern
        self.F_REAL = C.CFUNCTYPE(None, C.c_void_p, self.fp, self.fp)

    def pf(self, a):
        return a.ctypes.data_as(self.fp) if a is not None else None


def build(force=False):
    """Compile oracle/fd_oracle.c -> oracle/_build/libfd_oracle.so (gcc)."""
    src = os.path.join(_HERE, "fd_oracle.c")
    stale = [so for so in (_SO, _SO32, _SO_OMP) if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)]
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_libs = {}


def lib(dtype=np.float64, omp=False):
    """The oracle for one element type; omp=True: the multi-threaded Float64 build (bench.py's upper-bound line)."""
    T = _T(dtype)
    key = "omp" if omp else T.f32
    if key not in _libs:
        build()
        L = C.CDLL(_SO_OMP if omp else (_SO32 if T.f32 else _SO))
        _f64p, dbl = T.fp, T.c_real          # noqa: F841  (the prototypes below are written in terms of these)
        L.fdo_jacobian_cached.restype = C.c_int
        L.fdo_jacobian_cached.argtypes = [
            C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _f64p, _f64p, _f64p, _f64p, _f64p,
            C.c_void_p, C.c_void_p, _f64p, _i64p, dbl, dbl, dbl,
            C.POINTER(T.Pattern), _i64p]
        L.fdo_default_relstep.restype = dbl
        L.fdo_default_relstep.argtypes = [C.c_int]
        L.fdo_findstructralnz_dense.restype = C.c_int64
        L.fdo_findstructralnz_dense.argtypes = [_f64p, C.c_int64, C.c_int64, _i64p, _i64p]
        L.fdo_jacobian_oop_dense_forward.restype = None
        L.fdo_jacobian_oop_dense_forward.argtypes = [
            C.c_void_p, C.c_void_p, _f64p, C.c_int64, C.c_int64, dbl, dbl,
            dbl, _f64p]
        L.fdo_jvp.restype = C.c_int
        L.fdo_jvp.argtypes = [C.c_int, C.c_void_p, C.c_void_p, _f64p, _f64p, C.c_int64, C.c_int64, _f64p, dbl,
                              dbl, dbl, _f64p, _f64p, _f64p, _f64p]
        L.fdo_build_tridiag_csc.restype = None
        L.fdo_build_tridiag_csc.argtypes = [C.c_int64, _i64p, _i64p]
        _libs[key] = L
    return _libs[key]


def set_omp_threads(n):
    """Threads of the multi-threaded build (lib(omp=True)); the serial builds ignore it."""
    L = lib(omp=True)
    L.fdo_set_num_threads.argtypes = [C.c_int]
    L.fdo_set_num_threads(int(n))
    L.fdo_max_threads.restype = C.c_int
    return int(L.fdo_max_threads())


def default_relstep(fdtype, dtype=np.float64):
    return float(lib(dtype).fdo_default_relstep(FDTYPES[fdtype]))


def _p64(a):
    return a.ctypes.data_as(_i64p) if a is not None else None


def _pf(a):
    return a.ctypes.data_as(_f64p) if a is not None else None


class Fixture:
    """One of the C fixture f!s: (real fn, complex fn, ctx int64 array)."""

    def __init__(self, name, *ctx, dtype=np.float64, omp=False):
        L = lib(dtype, omp=omp)
        self.dtype = np.dtype(dtype)
        self.name = name
        self.ctx = np.asarray(ctx, dtype=np.int64)
        self.f = C.cast(getattr(L, "fdo_f_" + name), C.c_void_p)
        fc = getattr(L, "fdo_fc_" + name, None)
        self.fc = C.cast(fc, C.c_void_p) if fc is not None else None
        self.ctxp = self.ctx.ctypes.data_as(C.c_void_p)

    def __call__(self, x):
        """Evaluate on a numpy vector (real or complex); M inferred by the caller via out=."""
        raise NotImplementedError


class PyF:
    """A Python f!(fx, x) on numpy arrays (real and complex) as oracle callbacks."""

    def __init__(self, fn, M, N, dtype=np.float64):
        self.fn, self.M, self.N = fn, M, N
        self.calls = 0
        self.dtype = np.dtype(dtype)
        T = _T(dtype)

        def _real(_ctx, fxp, xp):
            fx = np.ctypeslib.as_array(fxp, shape=(M,))
            x = np.ctypeslib.as_array(xp, shape=(N,))
            self.calls += 1
            fn(fx, x)

        def _cplx(_ctx, fxp, xp):
            fx = np.ctypeslib.as_array(C.cast(fxp, T.fp), shape=(2 * M,)).view(T.cplx)
            x = np.ctypeslib.as_array(C.cast(xp, T.fp), shape=(2 * N,)).view(T.cplx)
            self.calls += 1
            fn(fx, x)

        self._keep = (T.F_REAL(_real), F_CPLX(_cplx))
        self.f = C.cast(self._keep[0], C.c_void_p)
        self.fc = C.cast(self._keep[1], C.c_void_p)
        self.ctxp = None


def findstructralnz_dense(A):
    A = np.asfortranarray(A, dtype=np.float64)
    m, n = A.shape
    cnt = int(np.count_nonzero(A))
    rows = np.empty(cnt, np.int64)
    cols = np.empty(cnt, np.int64)
    got = lib().fdo_findstructralnz_dense(_pf(A), m, n, _p64(rows), _p64(cols))
    assert got == cnt
    return rows, cols


def tridiag_csc(n):
    colptr = np.empty(n + 1, np.int64)
    rowval = np.empty(3 * n - 2 if n > 1 else 1, np.int64)
    lib().fdo_build_tridiag_csc(n, _p64(colptr), _p64(rowval))
    return colptr, rowval


def jacobian(fdtype, f, x, colorvec, M=None, *, kind=PAT_NONE, colptr=None, rowval=None,
             rows_index=None, cols_index=None, l=0, u=0, blk_sizes=None, bl=0, bu=0,
             block_starts=None, block_strides=None, out_len=None, lam=0, mu=0, f_in=None, relstep=None,
             absstep=None, dir=1.0, cache=None, mutate_x=False, dtype=None, omp=False, runner=False):
    """Run the cached in-place path.  Returns dict(out=..., fcalls=..., x_after=...).

    out: nzval (CSC_COMMON) | dense col-major M x N (NONE / *_DENSEJ) | banded data (l+u+1, N)
         | flat block data | (dl, d, du) for COO_TRIDIAG.
    cache: optional dict of pre-poisoned arrays x1,x2,fx,fx1 (test/cache_reuse_tests.jl).
    """
    dtype = np.dtype(dtype if dtype is not None else getattr(f, "dtype", np.float64))   # eltype(x): float64 | float32
    T = _T(dtype)
    L = lib(dtype, omp=omp)
    _pf = T.pf
    fd = FDTYPES[fdtype]
    x = np.array(x, dtype=T.real) if not mutate_x else x
    N = x.size
    M = N if M is None else M
    colorvec = np.ascontiguousarray(colorvec, dtype=np.int64)
    assert colorvec.size == N, "DimensionMismatch (src/jacobians.jl:516)"
    relstep = default_relstep(fdtype, dtype) if relstep is None else relstep
    absstep = relstep if absstep is None else absstep

    pat = T.Pattern()
    pat.kind, pat.M, pat.N = kind, M, N
    keep = []

    def i64(a):
        a = np.ascontiguousarray(a, dtype=np.int64)
        keep.append(a)
        return a

    out1 = out2 = None
    if kind in (PAT_CSC_COMMON, PAT_CSC_DENSEJ):
        colptr, rowval = i64(colptr), i64(rowval)
        pat.colptr, pat.rowval = _p64(colptr), _p64(rowval)
        nnz = int(colptr[-1] - 1)
        out0 = np.full(nnz if kind == PAT_CSC_COMMON else M * N, np.nan, T.real)
    elif kind in (PAT_COO_DENSEJ, PAT_COO_TRIDIAG, PAT_COO_DENSEJ_ACCUM):
        rows_index, cols_index = i64(rows_index), i64(cols_index)
        pat.rows_index, pat.cols_index, pat.ncoo = _p64(rows_index), _p64(cols_index), rows_index.size
        if kind in (PAT_COO_DENSEJ, PAT_COO_DENSEJ_ACCUM):
            out0 = np.full(M * N, np.nan, T.real)
        else:
            out0 = np.full(N, np.nan, T.real)
            out1 = np.full(max(N - 1, 0), np.nan, T.real)
            out2 = np.full(max(N - 1, 0), np.nan, T.real)
    elif kind == PAT_BANDED:
        pat.l, pat.u = l, u
        out0 = np.full((l + u + 1) * N, np.nan, T.real)
    elif kind in (PAT_BLOCKBANDED, PAT_BANDEDBLOCKBANDED):
        pat.lam, pat.mu = lam, mu
        blk_sizes, block_starts, block_strides = i64(blk_sizes), i64(block_starts), i64(block_strides)
        pat.nblk, pat.blk_sizes, pat.bl, pat.bu = blk_sizes.size, _p64(blk_sizes), bl, bu
        pat.block_starts, pat.block_strides = _p64(block_starts), _p64(block_strides)
        out0 = np.full(out_len, np.nan, T.real)
    else:
        out0 = np.full(M * N, np.nan, T.real)
    pat.out0, pat.out_len = _pf(out0), out0.size
    if out1 is not None:
        pat.out1, pat.out2 = _pf(out1), _pf(out2)

    cache = cache or {}
    x1 = np.array(cache.get("x1", np.zeros(N)), dtype=T.real)
    x2 = np.zeros(N, T.real)
    fx = np.array(cache.get("fx", np.zeros(M)), dtype=T.real)
    fx1 = np.array(cache.get("fx1", np.zeros(M)), dtype=T.real)
    cx1 = np.zeros(N, T.cplx)
    cfx = np.zeros(M, T.cplx)
    fin = None if f_in is None else np.ascontiguousarray(f_in, dtype=T.real)
    fcalls = np.zeros(1, np.int64)

    def call():
        """One finite_difference_jacobian! with the cache arrays above (JacobianCache reuse): only the C call."""
        return L.fdo_jacobian_cached(fd, f.f, f.fc, f.ctxp, _pf(x), _pf(x1), _pf(x2), _pf(fx), _pf(fx1),
                                     cx1.ctypes.data_as(C.c_void_p), cfx.ctypes.data_as(C.c_void_p),
                                     _pf(fin), _p64(colorvec), relstep, absstep, float(dir),
                                     C.byref(pat), _p64(fcalls))

    if runner:   # timing harness: (call, output array) with everything allocated once, like a reused JacobianCache
        call.keep = (keep, x, x1, x2, fx, fx1, cx1, cfx, fin, colorvec, fcalls, pat, out0, out1, out2, f)
        return call, out0
    rc = call()
    if rc != 0:
        raise ValueError("fdtype_error")
    if kind == PAT_COO_TRIDIAG:
        out = (out1, out0, out2)
    elif kind in (PAT_NONE, PAT_CSC_DENSEJ, PAT_COO_DENSEJ, PAT_COO_DENSEJ_ACCUM):
        out = out0.reshape((M, N), order="F")
    elif kind == PAT_BANDED:
        out = out0.reshape((l + u + 1, N), order="F")
    else:
        out = out0
    return {"out": out, "fcalls": int(fcalls[0]), "x_after": x, "x1": x1, "fx": fx, "fx1": fx1}


def jacobian_oop_dense_forward(f, x, M=None, relstep=None, absstep=None, dir=1.0):
    dtype = np.dtype(getattr(f, "dtype", np.float64))
    T = _T(dtype)
    _pf = T.pf
    x = np.ascontiguousarray(x, dtype=T.real)
    N = x.size
    M = N if M is None else M
    relstep = default_relstep("forward", dtype) if relstep is None else relstep
    absstep = relstep if absstep is None else absstep
    J = np.empty(M * N, T.real)
    lib(dtype).fdo_jacobian_oop_dense_forward(f.f, f.ctxp, _pf(x), M, N, relstep, absstep, float(dir), _pf(J))
    return J.reshape((M, N), order="F")


def jvp(fdtype, f, x, v, M=None, f_in=None, relstep=None, absstep=None, dir=1.0):
    """finite_difference_jvp! (src/jvp.jl:238-274) -> dict(jvp=..., eps=...)."""
    dtype = np.dtype(getattr(f, "dtype", np.float64))
    T = _T(dtype)
    _pf = T.pf
    x = np.ascontiguousarray(x, dtype=T.real)
    v = np.ascontiguousarray(v, dtype=T.real)
    N = x.size
    M = N if M is None else M
    relstep = default_relstep(fdtype, dtype) if relstep is None else relstep
    absstep = relstep if absstep is None else absstep
    x1, fx1, out, eps = np.zeros(N, T.real), np.zeros(M, T.real), np.zeros(M, T.real), np.zeros(1, T.real)
    fin = None if f_in is None else np.ascontiguousarray(f_in, dtype=T.real)
    rc = lib(dtype).fdo_jvp(FDTYPES[fdtype], f.f, f.ctxp, _pf(x), _pf(v), M, N, _pf(fin), relstep, absstep, float(dir),
                       _pf(x1), _pf(fx1), _pf(out), _pf(eps))
    if rc != 0:
        raise ValueError("finite_difference_jvp doesn't support :complex-mode finite diff")
    return {"jvp": out, "eps": float(eps[0])}
