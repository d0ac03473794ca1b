"""Runtime-compiled row functors (fd_f_compile_rows, csrc/fdjac_jit.hip): a residual given as SOURCE reaches the one-launch call --
the step-size launch + the column store instantiated for the functor -- and reproduces the bits of the built-in family / the oracle."""
import struct

import numpy as np
import pytest

import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TRIDIAG_NL = """
// test/coloring_tests.jl:5-13 with the nonlinear term of the tridiag_nl fixture: (x[i-1] - 2 x[i]) + x[i+1] + (x[i] x[i]) x[i+1]
struct TridiagNL {
    long long n;
    template <class P> __device__ real_t operator()(long long i, const P &X) const
    {
        // (every coordinate is fetched unconditionally from a clamped index and selected away outside: no load inside a per-lane branch)
        const real_t xi = X(i), xm = X(i > 0 ? i - 1 : i), xp = X(i + 1 < n ? i + 1 : i);
        const real_t a = i > 0 ? xm : (real_t)0, b = i + 1 < n ? xp : (real_t)0;
        real_t v = (a - (real_t)2 * xi) + b;
        v = v + (xi * xi) * b;
        return v;
    }
};
"""


def _dev(a, dt=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")


@pytest.mark.parametrize("fdtype", ["forward", "central"])
def test_jit_tridiag_nl_reproduces_the_builtin_bits(fdtype):
    N = 200_003
    colptr, rowval = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    x = _dev(np.random.default_rng(1).random(N))
    fb = fd.BuiltinF("tridiag_nl", N)
    ref_plan = fd.make_plan(J, J, colors, fdtype)
    ref_plan.set_lazy(fb)
    ref = _dev(np.full(rowval.size, np.nan))
    ref_plan.jacobian(fb, x, [ref])
    fj = fd.JitF(TRIDIAG_NL, "TridiagNL", N, N, params=struct.pack("q", N))
    # (1) the opaque route: the compiled functor as a plain fd_f_launch (materialised points)
    p1 = fd.make_plan(J, J, colors, fdtype)
    o1 = _dev(np.full(rowval.size, np.nan))
    p1.jacobian(fj, x, [o1])
    assert torch.equal(o1.view(torch.int64), ref.view(torch.int64))
    # (2) the one-launch route: the column store instantiated for the functor (FD_PLAN_STORE_CSC_ALWAYS: the pattern is a band)
    p2 = fd.make_plan(J, J, colors, fdtype, store_csc_always=True)
    p2.set_lazy(fj)
    o2 = _dev(np.full(rowval.size, np.nan))
    n0 = fj.launches
    p2.jacobian(fj, x, [o2])
    assert p2.info(fd.lib.INFO_STORE_CSC) == rowval.size and p2.info(fd.lib.INFO_LAZY_STORE) == 1
    assert fj.launches - n0 == 1                       # ONE launch of the functor: no f(x) pass, no hand-over
    assert p2.fcalls_last == (1 + 3 if fdtype == "forward" else 6)
    assert torch.equal(o2.view(torch.int64), ref.view(torch.int64))
    # the same source again: served from the cache (same module), a second functor object works independently
    fj2 = fd.JitF(TRIDIAG_NL, "TridiagNL", N, N, params=struct.pack("q", N))
    o3 = _dev(np.full(rowval.size, np.nan))
    p2.set_lazy(fj2)
    p2.jacobian(fj2, x, [o3])
    assert torch.equal(o3.view(torch.int64), ref.view(torch.int64))


def test_jit_on_a_general_pattern_matches_the_analytic_jacobian_and_float32():
    # a functor on a scattered pattern (the sparse family's residual restated in source form would need its tables; here a 2-D 9-point
    # Moore-neighbourhood residual on a small grid, pattern built on the host): oracle parity to the stated tolerance, Float32 too
    nx, ny = 37, 23
    N = nx * ny
    src = """
    struct Moore {
        long long nx, ny;
        template <class P> __device__ real_t operator()(long long k, const P &X) const
        {
            const long long j = k / nx, i = k - j * nx;
            real_t s = 0;
            bool first = true;
            for (int dj = -1; dj <= 1; ++dj)
                for (int di = -1; di <= 1; ++di) {
                    if (di == 0 && dj == 0) continue;
                    const long long ii = i + di, jj = j + dj;
                    const bool in = ii >= 0 && ii < nx && jj >= 0 && jj < ny;
                    const real_t xv = X(in ? jj * nx + ii : k);
                    const real_t v = in ? (real_t)0.5 * xv : (real_t)0;
                    s = first ? v : s + v;
                    first = false;
                }
            const real_t c = X(k);
            return (s - (real_t)4 * c) + (c * c) * c;
        }
    };
    """
    dense = np.zeros((N, N))
    for k in range(N):
        j, i = divmod(k, nx)
        for dj in (-1, 0, 1):
            for di in (-1, 0, 1):
                ii, jj = i + di, j + dj
                if 0 <= ii < nx and 0 <= jj < ny:
                    dense[k, jj * nx + ii] = 1
    colptr, rowval = P.csc_from_dense(dense)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    colors = fd.matrix_colors(J)
    xh = np.random.default_rng(2).random(N) + 0.2
    # analytic Jacobian: 0.5 at the neighbours, -4 + 3 x_k^2 on the diagonal
    want = np.where(np.eye(N, dtype=bool), -4.0 + 3.0 * xh[None, :] ** 2, 0.5 * dense)
    cols = P.csc_cols(colptr) - 1
    want_nz = want[rowval - 1, cols]
    for dtype, tol in ((np.float64, 2e-6), (np.float32, 2e-2)):
        tdt = torch.float64 if dtype == np.float64 else torch.float32
        fj = fd.JitF(src, "Moore", N, N, params=struct.pack("qq", nx, ny), dtype=dtype)
        x = _dev(xh, tdt)
        for store in (False, True):
            plan = fd.make_plan(J, J, colors, "forward", dtype=dtype, store_csc=store)
            if store:
                plan.set_lazy(fj)
            out = torch.full((rowval.size,), float("nan"), dtype=tdt, device="cuda")
            plan.jacobian(fj, x, [out])
            if store:
                assert plan.info(fd.lib.INFO_LAZY_STORE) == 1
            assert np.max(np.abs(out.cpu().numpy().astype(np.float64) - want_nz)) < tol


def test_jit_errors_are_reported():
    with pytest.raises(fd.lib.FdError) as e:
        fd.JitF("struct Broken { template <class P> __device__ real_t operator()(long long r, const P &X) const { return nonsense; } };",
                "Broken", 10, 10)
    assert e.value.code == 1 and "compil" in str(e.value)
    with pytest.raises(fd.lib.FdError):       # the parameter bytes must be the functor object
        fd.JitF(TRIDIAG_NL, "TridiagNL", 10, 10, params=b"\x00" * 3)


PENTA = """
// a pentadiagonal residual: row i reads x[i-2 .. i+2]; nonlinear in x[i] and x[i+2]
struct Penta {
    long long n;
    template <class P> __device__ real_t operator()(long long i, const P &X) const
    {
        const real_t c = X(i);
        const real_t a2 = X(i > 1 ? i - 2 : i), a1 = X(i > 0 ? i - 1 : i), b1 = X(i + 1 < n ? i + 1 : i), b2 = X(i + 2 < n ? i + 2 : i);
        const real_t m2 = i > 1 ? a2 : (real_t)0, m1 = i > 0 ? a1 : (real_t)0, p1 = i + 1 < n ? b1 : (real_t)0, p2 = i + 2 < n ? b2 : (real_t)0;
        real_t v = ((m2 - (real_t)4 * m1) + (real_t)6 * c) - (real_t)4 * p1;
        v = (v + p2) + (c * c) * p2;
        return v;
    }
};
"""


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("storage", ["csc", "tridiagonal", "banded"])
@pytest.mark.parametrize("N", [200_003, 130, 1])
def test_jit_functor_stores_an_exact_band_itself(fdtype, storage, N):
    # fd_band_store_cols (include/fdjac_device.h) instantiated for the compiled functor: with FD_LAZY_CAP_STORE the plan of an exact
    # band with cyclic colours hands the launcher a fd_band_store -- no compact pattern copy, no index read, ONE launch after the step
    # sizes; CSC nzval, a Tridiagonal's three diagonals and BandedMatrix data; the built-in family's bits
    colors = P.cyclic_colors(N, 3)
    x = _dev(np.random.default_rng(3).random(N) + 0.1)
    fb = fd.BuiltinF("tridiag_nl", N)
    fj = fd.JitF(TRIDIAG_NL, "TridiagNL", N, N, params=struct.pack("q", N))
    assert fj.lazy_caps & fd.lib.LAZY_CAP_STORE

    def build():
        if storage == "csc":
            cp, rv = P.tridiag_csc(N)
            J = fd.SparseMatrixCSC(N, N, cp, rv, None)
            return J, J, [_dev(np.full(rv.size, np.nan))]
        if storage == "tridiagonal":
            J = fd.Tridiagonal(None, torch.empty(N, dtype=torch.float64, device="cuda"), None)
            return J, None, [_dev(np.full(max(N - 1, 0), np.nan)), _dev(np.full(N, np.nan)), _dev(np.full(max(N - 1, 0), np.nan))]
        J = fd.BandedMatrix(torch.zeros((N, 3), dtype=torch.float64, device="cuda").t(), N, 1, 1)
        return J, None, [_dev(np.full(3 * N, np.nan))]

    outs = {}
    for name, f in (("builtin", fb), ("jit", fj)):
        J, sp, out = build()
        plan = fd.make_plan(J, sp, colors, fdtype)
        plan.set_lazy(f)
        n0 = f.counts()[0] if name == "builtin" else f.launches
        plan.jacobian(f, x, out)
        n1 = f.counts()[0] if name == "builtin" else f.launches
        if N >= 3:
            assert plan.info(fd.lib.INFO_LAZY_STORE) == 1 and plan.info(fd.lib.INFO_STORE_CSC) == 0
            assert n1 - n0 == 1, (name, n1 - n0)
        C = int(np.max(colors))
        assert plan.fcalls_last == (1 + C if fdtype == "forward" else 2 * C)
        outs[name] = [o.clone() for o in out]
    for a, b in zip(outs["builtin"], outs["jit"]):
        if storage == "banded" and N > 1:
            # (the two corner slots of BandedMatrix data belong to no entry: whatever the kernels leave there is not compared)
            a, b = a.view(N, 3).clone(), b.view(N, 3).clone()
            a[0, 0] = b[0, 0] = 0.0
            a[N - 1, 2] = b[N - 1, 2] = 0.0
        assert not torch.isnan(b).any() or storage == "banded"
        assert torch.equal(a.view(torch.int64), b.view(torch.int64)), storage


BANDF = """
// a residual whose Jacobian is the exact band (l, u): row i = sum over c = i - l .. i + u of w(i, c) (x_c + x_c^2 / 4), left to right
struct BandF {
    long long n;
    int l, u;
    template <class P> __device__ real_t operator()(long long i, const P &X) const
    {
        real_t s = 0;
        bool first = true;
        for (long long c = i - l; c <= i + u; ++c) {
            const bool in = c >= 0 && c < n;
            const real_t v = X(in ? c : i);
            const real_t t = ((real_t)1 + (real_t)0.125 * (real_t)(int)((i + 3 * c) & 7)) * (v + ((real_t)0.25 * v) * v);
            if (in) { s = first ? t : s + t; first = false; }
        }
        return s;
    }
};
"""


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("storage", ["csc", "banded"])
@pytest.mark.parametrize("lu", [(2, 2), (3, 3), (2, 1), (1, 0), (0, 1), (0, 0), (4, 4), (5, 4)])
def test_jit_band_store_other_bandwidths(fdtype, storage, lu):
    # (1, 1) and (2, 2) are compiled with every functor; other bandwidths with l + u <= 8 on first use (one more small compilation);
    # wider bands are declined and the plan falls back (column store of a CSC pattern, or the hand-over) -- the same bits either way
    l, u = lu
    N = 30_011
    C = l + u + 1
    colors = P.cyclic_colors(N, C)
    x = _dev(np.random.default_rng(4).random(N) + 0.1)
    fj = fd.JitF(BANDF, "BandF", N, N, params=struct.pack("qii", N, l, u))
    if storage == "csc":
        cp, rv = P.banded_csc(N, N, l, u)
        J = sp = fd.SparseMatrixCSC(N, N, cp, rv, None)
        n_out = rv.size
    else:
        J, sp = fd.BandedMatrix(torch.zeros((N, C), dtype=torch.float64, device="cuda").t(), N, l, u), None
        n_out = C * N
    ref_plan = fd.make_plan(J, sp, colors, fdtype)                       # opaque: materialised points, hand-over
    ref = _dev(np.zeros(n_out))
    ref_plan.jacobian(fj, x, [ref])
    plan = fd.make_plan(J, sp, colors, fdtype)
    plan.set_lazy(fj)
    out = _dev(np.zeros(n_out))
    n0 = fj.launches
    plan.jacobian(fj, x, [out])
    # (a BandedMatrix plan recognises cyclic colours through the step-size reduction's test: up to 8 of them)
    if (storage == "csc" and l + u <= 8) or (storage == "banded" and l + u + 1 <= 8):
        assert plan.info(fd.lib.INFO_LAZY_STORE) == 1 and fj.launches - n0 == 1
    assert torch.equal(out.view(torch.int64), ref.view(torch.int64)), (lu, storage)
    if storage == "csc":
        # analytic: d f_r / d x_c = w(r, c) (1 + x_c / 2)
        xh = x.cpu().numpy()
        cols = P.csc_cols(cp) - 1
        want = (1.0 + 0.125 * (((rv - 1) + 3 * cols) & 7)) * (1.0 + 0.5 * xh[cols])
        assert np.max(np.abs(out.cpu().numpy() - want)) < (5e-5 if fdtype == "forward" else 5e-7)


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("window", [(0, 1000), (1, 999), (333, 5000), (4999, 20011), (20010, 20011), (127, 129)])
def test_jit_band_store_on_a_column_window(fdtype, window):
    # a rank's column range (odd and even ends, a single column, ranges inside one wavefront): the compiled functor's band store writes
    # exactly its slice -- the built-in family's bits on the same window
    N = 20011
    cp, rv = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, cp, rv, None)
    colors = P.cyclic_colors(N, 3)
    x = _dev(np.random.default_rng(8).random(N) + 0.1)
    fb = fd.BuiltinF("tridiag_nl", N)
    fj = fd.JitF(TRIDIAG_NL, "TridiagNL", N, N, params=struct.pack("q", N))
    outs = []
    for f in (fb, fj):
        plan = fd.make_plan(J, J, colors, fdtype, col_window=window)
        plan.set_lazy(f)
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(f, x, [out])
        if window[1] - window[0] >= 3:
            assert plan.info(fd.lib.INFO_LAZY_STORE) == 1
        outs.append(out)
    assert not torch.isnan(outs[1]).any() and torch.equal(outs[0].view(torch.int64), outs[1].view(torch.int64))


TRIDIAG_NL_GENERIC = """
// the same residual, generic in the VALUE TYPE of the point (include/fdjac_device.h, "the complex step for row functors"): X(j) yields
// real_t for forward / central differences and fd_cplx<real_t> for the complex step
struct TridiagNLg {
    long long n;
    template <class P> __device__ typename P::value_type operator()(long long i, const P &X) const
    {
        typedef typename P::value_type V;
        const V xi = X(i), xm = X(i > 0 ? i - 1 : i), xp = X(i + 1 < n ? i + 1 : i);
        const V a = i > 0 ? xm : V{}, b = i + 1 < n ? xp : V{};
        V v = (a - (real_t)2 * xi) + b;
        v = v + (xi * xi) * b;
        return v;
    }
};
"""


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_jit_complex_step_opaque_and_column_store(oracle, dtype):
    # src/jacobians.jl:623-648 for a runtime-compiled functor: the functor's complex instantiation is compiled on first use -- as a
    # plain launcher on materialised complex points (the opaque route) and as fd_csc_store_cols_cplx (ONE launch: every stored entry's
    # row at x + i eps e_j, imag / eps).  Bits of the built-in family's complex step; the oracle to the stated tolerance; the same
    # functor still serves forward differences; a functor that names real_t explicitly is refused for the complex step with a message.
    N = 100_003
    t = torch.float64 if dtype == np.float64 else torch.float32
    colptr, rowval = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    xh = np.random.default_rng(8).random(N).astype(dtype)
    x = torch.as_tensor(xh, device="cuda")
    fb = fd.BuiltinF("tridiag_nl", N, dtype=dtype)
    ref_plan = fd.make_plan(J, J, colors, "complex", dtype=dtype)
    ref_plan.set_lazy(fb)
    ref = torch.full((rowval.size,), float("nan"), dtype=t, device="cuda")
    ref_plan.jacobian(fb, x, [ref])
    fj = fd.JitF(TRIDIAG_NL_GENERIC, "TridiagNLg", N, N, params=struct.pack("q", N), dtype=dtype)
    assert fj.lazy_caps & fd.lib.LAZY_CAP_STORE_CSC_COMPLEX
    # (1) opaque
    p1 = fd.make_plan(J, J, colors, "complex", dtype=dtype)
    o1 = torch.full_like(ref, float("nan"))
    p1.jacobian(fj, x, [o1])
    assert p1.fcalls_last == 3
    assert torch.equal(o1, ref)
    # (2) ONE launch through the column store
    p2 = fd.make_plan(J, J, colors, "complex", dtype=dtype, store_csc_always=True)
    p2.set_lazy(fj)
    o2 = torch.full_like(ref, float("nan"))
    n0 = fj.launches
    p2.jacobian(fj, x, [o2])
    assert fj.launches - n0 == 1 and p2.fcalls_last == 3
    assert torch.equal(o2, ref)
    # the oracle (Float64: the complex step is exact to rounding)
    if dtype == np.float64:
        want = oracle.jacobian("complex", oracle.Fixture("tridiag_nl", N), xh, colors, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)["out"]
        assert np.max(np.abs(o2.cpu().numpy() - want) / (1e-12 * np.abs(want) + 1e-300)) <= 1.0 or np.allclose(o2.cpu().numpy(), want, rtol=1e-12, atol=1e-14)
    # forward differences with the same (generic) functor: the bits of the built-in family
    pf = fd.make_plan(J, J, colors, "forward", dtype=dtype, store_csc_always=True)
    pf.set_lazy(fj)
    of = torch.full_like(ref, float("nan"))
    pf.jacobian(fj, x, [of])
    rf_plan = fd.make_plan(J, J, colors, "forward", dtype=dtype)
    rf_plan.set_lazy(fb)
    rf = torch.full_like(ref, float("nan"))
    rf_plan.jacobian(fb, x, [rf])
    assert torch.equal(of, rf)
    # a functor written on real_t: fine for forward differences, refused for the complex step -- loudly, with the reason
    fr = fd.JitF(TRIDIAG_NL, "TridiagNL", N, N, params=struct.pack("q", N), dtype=dtype)
    p3 = fd.make_plan(J, J, colors, "complex", dtype=dtype, store_csc_always=True)
    p3.set_lazy(fr)
    with pytest.raises(fd.lib.FdError) as ei:
        p3.jacobian(fr, x, [torch.full_like(ref, float("nan"))])
    assert "value_type" in str(ei.value)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_row_function_given_as_llvm_bitcode(tmp_path, dtype):
    # fd_f_link_rows_bitcode: what AMDGPU.jl / GPUCompiler would hand over for a Julia closure.  tests/bitcode_user_tridiag_nl.hip is
    # compiled to LLVM bitcode OFFLINE (no library header involved), linked at run time into the library's kernels, and must reproduce
    # the built-in family's bits: forward and central through the band store (ONE launch, no index read) and the column store, the
    # complex step through fd_csc_store_cols_cplx, and all three as an opaque f!.
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    bc = tmp_path / "user.bc"
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bitcode_user_tridiag_nl.hip")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fgpu-rdc", "-emit-llvm", "--offload-device-only", "-c",
                    "-DREAL=%s" % ("double" if dtype == np.float64 else "float"), src, "-o", str(bc)], check=True, capture_output=True)
    N = 150_001
    t = torch.float64 if dtype == np.float64 else torch.float32
    colptr, rowval = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    x = torch.as_tensor(np.random.default_rng(12).random(N).astype(dtype), device="cuda")
    fb = fd.BuiltinF("tridiag_nl", N, dtype=dtype)
    try:
        fu = fd.BitcodeF(bc.read_bytes(), N, N, params=struct.pack("q", N), dtype=dtype)
    except fd.lib.FdError as e:
        if "linking the caller's bitcode failed" in str(e):
            # this PROCESS's hiprtc is the one torch bundles (ROCm 7.0, LLVM 20); /opt/rocm's hipcc writes LLVM 22 bitcode, which an older
            # reader refuses ("Not an int attribute (Producer: LLVM22 Reader: LLVM 20)").  The same bitcode is linked and checked in a process
            # that uses /opt/rocm's own runtime: tests/test_gpu_edge.py::test_c_client_links_a_row_function_given_as_bitcode
            pytest.skip("hipcc's bitcode is newer than the LLVM of the hiprtc loaded into this process (torch's): covered by the C client")
        raise
    assert fu.lazy_caps & fd.lib.LAZY_CAP_STORE_CSC and fu.lazy_caps & fd.lib.LAZY_CAP_STORE
    for fdtype in ("forward", "central", "complex"):
        ref_plan = fd.make_plan(J, J, colors, fdtype, dtype=dtype)
        ref_plan.set_lazy(fb)
        ref = torch.full((rowval.size,), float("nan"), dtype=t, device="cuda")
        ref_plan.jacobian(fb, x, [ref])
        # opaque: materialised points, the linked function behind a plain fd_f_launch
        p0 = fd.make_plan(J, J, colors, fdtype, dtype=dtype)
        o0 = torch.full_like(ref, float("nan"))
        p0.jacobian(fu, x, [o0])
        assert torch.equal(o0, ref), (fdtype, "opaque")
        # ONE launch: the column store (any pattern) ...
        p1 = fd.make_plan(J, J, colors, fdtype, dtype=dtype, store_csc_always=True)
        p1.set_lazy(fu, store=True)
        o1 = torch.full_like(ref, float("nan"))
        n0 = fu.launches
        p1.jacobian(fu, x, [o1])
        assert fu.launches - n0 == 1, fdtype
        assert torch.equal(o1, ref), (fdtype, "column store")
        # ... and, for an exact band with cyclic colours, the band store (forward / central)
        if fdtype != "complex":
            p2 = fd.make_plan(J, J, colors, fdtype, dtype=dtype)
            p2.set_lazy(fu)
            assert p2.info(fd.lib.INFO_LAZY_STORE) == 1
            o2 = torch.full_like(ref, float("nan"))
            n0 = fu.launches
            p2.jacobian(fu, x, [o2])
            assert fu.launches - n0 == 1
            assert torch.equal(o2, ref), (fdtype, "band store")


# ---- separable residuals from their TERM alone (fd_f_compile_terms): the row-wise store for user functors ---------------------------
SPARSE_TERMS = """
// the sparse family's term (csrc/fdjac_functor_f.hip, SparseF): w(r, j) * (v + (v / 4) v), w = 1 + ((r + 3 j) mod 8) / 8
struct SparseTerms {
    template <class T> __device__ T term(long long r, long long j, T v) const
    {
        return ((real_t)1 + (real_t)0.125 * (real_t)(int)((r + 3 * j) & 7)) * (v + ((real_t)0.25 * v) * v);
    }
};
"""
SCALED_TERMS = """
struct ScaledTerms {
    double a; long long shift;      // (parameters travel byte for byte)
    template <class T> __device__ T term(long long r, long long j, T v) const
    {
        return ((real_t)a + (real_t)0.125 * (real_t)(int)((r + 3 * j + shift) & 7)) * (v + ((real_t)0.25 * v) * v);
    }
};
"""


def _random_band(M, N, per_col, reach, seed, empty=0.03):
    rng = np.random.default_rng(seed)
    centre = (np.arange(N) * M) // max(N, 1)
    offs = np.sort(rng.integers(-reach, reach + 1, size=(N, per_col)), axis=1)
    rows = centre[:, None] + offs
    keep = (rows >= 0) & (rows < M)
    keep[:, 1:] &= rows[:, 1:] != rows[:, :-1]
    keep[rng.random(N) < empty] = False
    cnt = keep.sum(axis=1)
    colptr = np.empty(N + 1, np.int64)
    colptr[0] = 1
    np.cumsum(cnt, out=colptr[1:])
    colptr[1:] += 1
    return colptr, (rows[keep] + 1).astype(np.int64)


def _sparse_np(M, N, colptr, rowval, a=1.0, shift=0):
    cols = P.csc_cols(colptr) - 1
    rows = rowval - 1
    order = np.lexsort((cols, rows))
    rs, cs = rows[order], cols[order]
    cnt = np.bincount(rs, minlength=M)
    start = np.concatenate([[0], np.cumsum(cnt)])[:-1]
    maxlen = int(cnt.max()) if cnt.size else 0
    w = a + 0.125 * ((rs + 3 * cs + shift) & 7)

    def f(fx, xx):
        t = w * (xx[cs] + (0.25 * xx[cs]) * xx[cs])
        out = np.zeros(M, dtype=xx.dtype)
        for k in range(maxlen):
            sel = np.nonzero(cnt > k)[0]
            out[sel] = t[start[sel] + k] if k == 0 else out[sel] + t[start[sel] + k]
        fx[:] = out
    return f


@pytest.mark.parametrize("fdtype", ["forward", "central"])
# (3000, 12, 40): more entries per tile than its staged run holds; (2000, 3, 700): the widest window; (300, 4, 9): one partial tile
@pytest.mark.parametrize("case", [(300, 4, 9, 1), (5000, 6, 300, 4), (70001, 6, 300, 7), (3000, 12, 40, 8), (2000, 3, 700, 9), (4100, 20, 30, 10)])
def test_terms_functor_row_wise_store_has_the_bits_of_every_other_route_and_the_oracle(monkeypatch, oracle, fdtype, case):
    N, per_col, reach, seed = case
    if seed % 2 == 0:
        monkeypatch.setenv("FDJAC_ROWS_ENTS", "1")      # (half of the cases through the entry-parallel form, fd_csc_store_ents)
    colptr, rowval = _random_band(N, N, per_col, reach, seed)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    colors = fd.matrix_colors(J)
    x = np.random.default_rng(seed).random(N) + 0.1
    xd = _dev(x)
    # the reference bits: the built-in sparse family through the hand-over path (oracle-checked in tests/test_gpu_storetable.py)
    fb = fd.BuiltinF.sparse(N, N, colptr, rowval)
    ph = fd.make_plan(J, J, colors, fdtype)
    ref = _dev(np.full(rowval.size, np.nan))
    ph.jacobian(fb, xd, [ref])
    # the terms functor on a plan that keeps its pattern by rows: ONE launch, row by row
    pr = fd.make_plan(J, J, colors, fdtype, store_rows=True)
    rl = pr.row_lists()
    assert rl["entries"] == rowval.size and rl["row_ptr"] and rl["row_col"] and rl["row_slot"]
    ft = fd.JitTerms(SPARSE_TERMS, "SparseTerms", pr)
    pr.set_lazy(ft)
    a = _dev(np.full(rowval.size, np.nan))
    n0, r0 = ft.launches, ft.row_stores
    pr.jacobian(ft, xd, [a])
    assert pr.info(fd.lib.INFO_LAZY_STORE) == 1
    assert ft.launches - n0 == 1 and ft.row_stores - r0 == 1            # one launch, and it was the row-wise store
    assert torch.equal(a.view(torch.int64), ref.view(torch.int64))
    # the same functor through the column store (a plan without row lists: another serial) and through the opaque route
    pc = fd.make_plan(J, J, colors, fdtype, store_csc=True)
    pc.set_lazy(ft)
    b = _dev(np.full(rowval.size, np.nan))
    r1 = ft.row_stores
    pc.jacobian(ft, xd, [b])
    assert ft.row_stores == r1 and pc.info(fd.lib.INFO_LAZY_STORE) == 1
    assert torch.equal(b.view(torch.int64), ref.view(torch.int64))
    po = fd.make_plan(J, J, colors, fdtype)
    c = _dev(np.full(rowval.size, np.nan))
    po.jacobian(ft, xd, [c])
    assert torch.equal(c.view(torch.int64), ref.view(torch.int64))
    # and the oracle itself, directly (the reference's loop on the numpy restatement of the residual)
    of = oracle.PyF(_sparse_np(N, N, colptr, rowval), N, N)
    want = oracle.jacobian(fdtype, of, x, colors, M=N, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    g = a.cpu().numpy()
    fs = max(float(np.abs(want["fx"]).max()) if "fx" in want else 10.0 * per_col, 1.0)
    atol = 16 * np.finfo(np.float64).eps * fs / float(np.min(np.abs(pr.epsilons())))      # tolerance: 16 ulp of f over the smallest step
    assert np.all(np.abs(g - want["out"]) <= 1e-6 * np.abs(want["out"]) + atol)


def test_terms_functor_parameters_float32_complex_step_and_refusals(oracle):
    N, per_col, reach, seed = 6000, 5, 120, 3
    colptr, rowval = _random_band(N, N, per_col, reach, seed)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    colors = fd.matrix_colors(J)
    x = np.random.default_rng(seed).random(N) + 0.1
    prm = struct.pack("dq", 1.5, 5)
    # parameters, Float64: row-wise store == column store == opaque, and the oracle on the restated residual
    for dtype, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
        xd = _dev(x, tdt)
        pr = fd.make_plan(J, J, colors, "forward", dtype=dtype, store_rows=True)
        ft = fd.JitTerms(SCALED_TERMS, "ScaledTerms", pr, params=prm)
        pr.set_lazy(ft)
        a = torch.full((rowval.size,), float("nan"), dtype=tdt, device="cuda")
        pr.jacobian(ft, xd, [a])
        assert ft.row_stores == 1
        po = fd.make_plan(J, J, colors, "forward", dtype=dtype)
        c = torch.full((rowval.size,), float("nan"), dtype=tdt, device="cuda")
        po.jacobian(ft, xd, [c])
        it = torch.int64 if dtype == np.float64 else torch.int32
        assert torch.equal(a.view(it), c.view(it))
        cols = P.csc_cols(colptr) - 1
        analytic = (1.5 + 0.125 * (((rowval - 1) + 3 * cols + 5) & 7)) * (1.0 + 0.5 * x[cols])
        assert np.max(np.abs(a.cpu().numpy().astype(np.float64) - analytic)) < (1e-5 if dtype == np.float64 else 5e-2)
        if dtype == np.float64:
            of = oracle.PyF(_sparse_np(N, N, colptr, rowval, 1.5, 5), N, N)
            want = oracle.jacobian("forward", of, x, colors, M=N, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
            atol = 16 * np.finfo(np.float64).eps * max(float(np.abs(want["fx"]).max()), 1.0) / float(np.min(np.abs(pr.epsilons())))
            assert np.all(np.abs(a.cpu().numpy() - want["out"]) <= 1e-6 * np.abs(want["out"]) + atol)
            # the complex step: the term instantiated on the complex type, through the column store; exact to rounding
            pz = fd.make_plan(J, J, colors, "complex", store_rows=True)
            fz = fd.JitTerms(SCALED_TERMS, "ScaledTerms", pz, params=prm)
            pz.set_lazy(fz)
            z = _dev(np.full(rowval.size, np.nan))
            pz.jacobian(fz, xd, [z])
            assert np.max(np.abs(z.cpu().numpy() - analytic)) < 1e-13
    # refusals: a plan without row lists; a parameter block of the wrong size
    p0 = fd.make_plan(J, J, colors, "forward", store_csc=True)
    with pytest.raises(fd.lib.FdError):
        p0.row_lists()
    pr = fd.make_plan(J, J, colors, "forward", store_rows=True)
    with pytest.raises(fd.lib.FdError):
        fd.JitTerms(SCALED_TERMS, "ScaledTerms", pr, params=b"\x00" * 3)
    with pytest.raises(fd.lib.FdError):
        fd.JitTerms("struct Bad { template <class T> __device__ T term(long long r, long long j, T v) const { return nonsense; } };", "Bad", pr)


# ---- a compiled row functor storing BlockBandedMatrix data itself (fd_colrange_store_cols) -------------------------------------------
BLOCK_COUPLED = """
// row k of block b: x_k (S_{b-1} + S_b + S_{b+1}) + sin(x_k), S = the block's coordinates added left to right (blocks outside the matrix: none)
struct BlockCoupled {
    long long nb, bs;
    template <class P> __device__ typename P::value_type operator()(long long k, const P &X) const
    {
        typedef typename P::value_type V;
        const long long b = k / bs;
        V tot = V();
        bool first = true;
        for (long long bb = b - 1; bb <= b + 1; ++bb) {
            if (bb < 0 || bb >= nb) continue;
            V s = V();
            for (long long i = 0; i < bs; ++i) { const V v = X(bb * bs + i); s = i == 0 ? v : s + v; }
            tot = first ? s : tot + s;
            first = false;
        }
        const V xk = X(k);
        return xk * tot + sin(xk);
    }
};
"""


@pytest.mark.parametrize("fdtype", ["forward", "central", "complex"])
@pytest.mark.parametrize("case", [(13, 8), (40, 32), (7, 70)])
def test_jit_functor_stores_blockbanded_data_itself(oracle, fdtype, case):
    # BlockBandedMatrix storage (ext/FiniteDiffBlockBandedMatricesExt.jl:44-68): a compiled functor's lazy launcher takes the plan's
    # column-range descriptor -- one wavefront per column, the column's rows across the lanes -- for all three fdtypes; the same
    # functor as an opaque f! (materialised points, decompression launch) gives the bit reference; then the oracle and the analytic J
    nb, bs = case
    N = nb * bs
    lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
    colors = lay.colors()
    xh = np.random.default_rng(12).random(N) - 0.3
    x = _dev(xh)
    Jb = fd.BlockBandedMatrix(None, lay)
    fj = fd.JitF(BLOCK_COUPLED, "BlockCoupled", N, N, params=struct.pack("qq", nb, bs))
    assert fj.lazy_caps & fd.lib.LAZY_CAP_STORE_COLRANGE
    po = fd.make_plan(Jb, Jb, colors, fdtype)
    ref = _dev(np.full(po.out_len(0), np.nan))
    n0 = fj.launches
    po.jacobian(fj, x, [ref])
    opaque_launches = fj.launches - n0
    ps = fd.make_plan(Jb, Jb, colors, fdtype)
    ps.set_lazy(fj)
    out = _dev(np.full(ps.out_len(0), np.nan))
    n0 = fj.launches
    ps.jacobian(fj, x, [out])
    assert ps.info(fd.lib.INFO_LAZY_STORE) == 1
    # ONE storing launch of the functor (forward: + one plain evaluation of f(x) that every column shares); no perturbation, no
    # materialised points, no decompression launch
    assert fj.launches - n0 == (2 if fdtype == "forward" else 1) and opaque_launches >= 1
    assert not torch.isnan(out).any()
    assert torch.equal(out.view(torch.int64), ref.view(torch.int64))
    # the oracle on the numpy restatement of the residual
    blk = np.repeat(np.arange(nb), bs)

    def f_np(fx, xx):
        sig = np.zeros(nb, dtype=xx.dtype)
        for i in range(bs):                                             # (left to right within a block, as the functor adds)
            sig = xx[i::bs][:nb] if i == 0 else sig + xx[i::bs][:nb]
        S = sig.copy()
        S[1:] = sig[:-1] + sig[1:]
        S[:-1] = S[:-1] + sig[1:]
        fx[:] = xx * S[blk] + np.sin(xx)
    want = oracle.jacobian(fdtype, oracle.PyF(f_np, N, N), xh, colors, kind=oracle.PAT_BLOCKBANDED, blk_sizes=lay.blk_sizes, bl=1, bu=1,
                           block_starts=lay.block_starts, block_strides=lay.block_strides, out_len=lay.data_len)
    g = out.cpu().numpy()
    if fdtype == "complex":
        assert np.max(np.abs(g - want["out"])) <= 1e-12 * max(1.0, np.max(np.abs(want["out"])))       # (no cancellation: a few ulp of |J|)
    else:
        fs = float(np.max(np.abs(xh)) * 3 * bs + 1)
        atol = 16 * np.finfo(np.float64).eps * fs / float(np.min(np.abs(ps.epsilons())))                 # 16 ulp of f over the smallest step
        assert np.all(np.abs(g - want["out"]) <= 1e-5 * np.abs(want["out"]) + atol)


# ---- a compiled row functor storing BandedBlockBandedMatrix data itself (fd_bbb_store_cols) ----------------------------------------------
GRID_NL = """
// an nx x ny grid, row k = (i, j): neighbours within `reach` along the grid row and the points straight above / below, nonlinear in x_k
struct GridNL {
    long long nx, ny, reach;
    template <class P> __device__ typename P::value_type operator()(long long k, const P &X) const
    {
        typedef typename P::value_type V;
        const long long j = k / nx, i = k - j * nx;
        const V c = X(k);
        V s = c * c;
        for (long long d = 1; d <= reach; ++d) {
            const bool hw = i - d >= 0, he = i + d < nx;
            const V w = X(hw ? k - d : k), e = X(he ? k + d : k);
            s = s + (hw ? (real_t)(0.5 / d) * w : (real_t)0);
            s = s + (he ? (real_t)(0.25 * d) * e * c : (real_t)0);
        }
        const bool hs = j > 0, hn = j + 1 < ny;
        const V sv = X(hs ? k - nx : k), nv = X(hn ? k + nx : k);
        s = s + (hs ? sv : (real_t)0);
        s = s + (hn ? (real_t)2 * nv : (real_t)0);
        return s;
    }
};
"""


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("case", [(40, 30, 1), (23, 17, 3), (64, 5, 2)])
def test_jit_functor_stores_bandedblockbanded_data_itself(oracle, fdtype, case):
    # BandedBlockBandedMatrix storage with uniform blocks (ext/FiniteDiffBlockBandedMatricesExt.jl:16-42): a compiled functor's lazy launcher
    # fills every slot of every in-band slab in one launch (fd_bbb_store_cols) -- the bits of the same functor as an opaque f! (perturb,
    # batched rows, k_decompress_bbb), oracle parity on the numpy restatement
    nx, ny, reach = case
    N = nx * ny
    lay = P.BandedBlockBandedLayout(np.full(ny, nx), 1, 1, reach, reach)
    colors = lay.colors()
    xh = np.random.default_rng(21).random(N) + 0.1
    x = _dev(xh)
    fj = fd.JitF(GRID_NL, "GridNL", N, N, params=struct.pack("qqq", nx, ny, reach))
    J = fd.BandedBlockBandedMatrix(None, lay)
    po = fd.make_plan(J, J, colors, fdtype)
    ref = _dev(np.full(lay.data_len, np.nan))
    po.jacobian(fj, x, [ref])
    ps = fd.make_plan(J, J, colors, fdtype)
    ps.set_lazy(fj)
    out = _dev(np.full(lay.data_len, np.nan))
    n0 = fj.launches
    ps.jacobian(fj, x, [out])
    assert ps.info(fd.lib.INFO_LAZY_STORE) == 1
    assert fj.launches - n0 == (2 if fdtype == "forward" else 1)          # the storing launch (+ one plain evaluation of f(x) for forward)
    assert not torch.isnan(out).any()
    assert torch.equal(out.view(torch.int64), ref.view(torch.int64))

    def f_np(fx, xx):
        X = xx.reshape(ny, nx)
        s = X * X
        for d in range(1, reach + 1):
            w = np.zeros_like(X)
            e = np.zeros_like(X)
            w[:, d:] = X[:, :-d]
            e[:, :-d] = X[:, d:]
            s = s + (0.5 / d) * w
            s = s + (0.25 * d) * e * X
        sv = np.zeros_like(X)
        nv = np.zeros_like(X)
        sv[1:] = X[:-1]
        nv[:-1] = X[1:]
        fx[:] = ((s + sv) + 2 * nv).reshape(-1)
    want = oracle.jacobian(fdtype, oracle.PyF(f_np, N, N), xh, colors, kind=oracle.PAT_BANDEDBLOCKBANDED, blk_sizes=lay.blk_sizes, bl=1, bu=1,
                           lam=reach, mu=reach, block_starts=lay.block_starts, block_strides=lay.block_strides, out_len=lay.data_len)
    g = out.cpu().numpy()
    atol = 16 * np.finfo(np.float64).eps * 10.0 / float(np.min(np.abs(ps.epsilons())))                  # 16 ulp of f (|f| < 10) over the smallest step
    assert np.all(np.abs(g - want["out"]) <= 1e-5 * np.abs(want["out"]) + atol)


@pytest.mark.parametrize("shape", [(257, 411), (411, 257)])
def test_terms_functor_on_a_non_square_pattern_takes_the_column_store(shape):
    # the row-wise kernel keeps the window of x by ROW index (square, locally banded patterns); any other shape: the same term through
    # fd_sep_rows in the column store -- the bits of the built-in sparse family's hand-over path
    M, N = shape
    colptr, rowval = _random_band(M, N, 4, 9, 31)
    J = fd.SparseMatrixCSC(M, N, colptr, rowval, None)
    colors = fd.matrix_colors(J)
    x = _dev(np.random.default_rng(31).random(N) + 0.1)
    fb = fd.BuiltinF.sparse(M, N, colptr, rowval)
    ph = fd.make_plan(J, J, colors, "forward")
    ref = _dev(np.full(rowval.size, np.nan))
    ph.jacobian(fb, x, [ref])
    pr = fd.make_plan(J, J, colors, "forward", store_rows=True)
    assert pr.row_lists()["entries"] == rowval.size
    ft = fd.JitTerms(SPARSE_TERMS, "SparseTerms", pr)
    pr.set_lazy(ft)
    out = _dev(np.full(rowval.size, np.nan))
    pr.jacobian(ft, x, [out])
    assert ft.row_stores == 0 and pr.info(fd.lib.INFO_LAZY_STORE) == 1
    assert torch.equal(out.view(torch.int64), ref.view(torch.int64))
