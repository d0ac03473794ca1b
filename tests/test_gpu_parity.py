"""GPU parity: libfdjac (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): per stored entry
    |J_gpu - J_cpu| <= 1e-6*|J_cpu| + 16*eps(Float64)*max|f| / |eps_c|
i.e. 1e-6 relative with an epsilon-scaled absolute floor: the oracle perturbs x in place and
un-perturbs it (as the reference does), the GPU perturbs from the pristine x, and a forward
difference amplifies ulp(f) by 1/eps.  The step sizes themselves must agree to 1e-12 relative.
"""
import numpy as np
import pytest

import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

EPS64 = np.finfo(np.float64).eps
FDTYPES = ["forward", "central", "complex"]


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


def _tol_ok(g, c, eps_min, fscale, what):
    g, c = np.asarray(g), np.asarray(c)
    atol = 16 * EPS64 * fscale / abs(eps_min)
    bad = np.abs(g - c) > 1e-6 * np.abs(c) + atol
    assert not bad.any(), "%s: %d entries off, worst %.3e (atol %.1e)" % (
        what, int(bad.sum()), float(np.max(np.abs(g - c))), atol)


def _oracle_eps(x, colors, fdtype):
    if fdtype == "complex":
        return np.full(int(colors.max()), EPS64)
    rel = fd.default_relstep(fdtype)
    out = []
    for c in range(1, int(colors.max()) + 1):
        out.append(max(rel * np.sqrt(np.linalg.norm(x * (colors == c))), rel))
    return np.array(out)


@pytest.mark.parametrize("fdtype,ncalls", [("forward", 4), ("central", 6), ("complex", 3)])
def test_tridiag_csc_reference_fixture(oracle, fdtype, ncalls):
    # test/coloring_tests.jl:33-49 on the device, through the mirrored Julia API
    N = 30
    x = np.random.default_rng(11).random(N)
    colors = np.tile([1, 2, 3], 10)
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.full(rowval.size, np.nan)))
    f = fd.BuiltinF("tridiag", N)
    cache = fd.JacobianCache(_dev(x), fdtype, colorvec=colors, sparsity=J)
    fd.finite_difference_jacobian_b(J, f, _dev(x), cache)
    assert f.fcalls == ncalls
    ref = oracle.jacobian(fdtype, oracle.Fixture("tridiag", N), x, colors, kind=oracle.PAT_CSC_COMMON,
                          colptr=colptr, rowval=rowval)
    got = J.nzval.cpu().numpy()
    eps = cache.last_plan.epsilons()
    assert np.allclose(eps, _oracle_eps(x, colors, fdtype), rtol=1e-12, atol=0)
    _tol_ok(got, ref["out"], np.min(np.abs(eps)), 4.0, "tridiag csc " + fdtype)
    # and the exact answer (-2 / 1), as the reference asserts with isapprox
    dense = P.csc_to_dense(N, N, colptr, rowval, got)
    exact = np.diag(np.full(N, -2.0)) + np.diag(np.ones(N - 1), 1) + np.diag(np.ones(N - 1), -1)
    assert np.linalg.norm(dense - exact) <= 1.5e-8 * np.linalg.norm(exact) * 4


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("N", [1, 2, 3, 257, 100003])
def test_tridiag_csc_sizes_nonlinear(oracle, fdtype, N):
    x = np.random.default_rng(100 + N).random(N)
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.zeros(rowval.size)))
    f = fd.BuiltinF("tridiag_nl", N)
    fd.finite_difference_jacobian_b(J, f, _dev(x), fdtype, colorvec=colors)
    ref = oracle.jacobian(fdtype, oracle.Fixture("tridiag_nl", N), x, colors, kind=oracle.PAT_CSC_COMMON,
                          colptr=colptr, rowval=rowval)
    assert f.fcalls == ref["fcalls"]
    eps_min = np.min(np.abs(_oracle_eps(x, colors, fdtype)))
    _tol_ok(J.nzval.cpu().numpy(), ref["out"], eps_min, 5.0, "tridiag_nl N=%d %s" % (N, fdtype))


@pytest.mark.parametrize("fdtype", FDTYPES)
def test_host_arrays_and_f_in(oracle, fdtype):
    # numpy in / numpy out through the ABI's host path; f_in reuse (src/jacobians.jl:540-545)
    N = 1000
    x = np.random.default_rng(5).random(N)
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, np.zeros(rowval.size))
    f = fd.BuiltinF("tridiag_nl", N)
    ofx = oracle.Fixture("tridiag_nl", N)
    fin = None
    if fdtype == "forward":
        xm, xp = np.concatenate([[0.0], x[:-1]]), np.concatenate([x[1:], [0.0]])
        fin = (xm - 2 * x) + xp + (x * x) * xp
    fd.finite_difference_jacobian_b(J, f, x, fdtype, np.float64, fin, colorvec=colors)
    ref = oracle.jacobian(fdtype, ofx, x, colors, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval, f_in=fin)
    assert f.fcalls == ref["fcalls"]
    _tol_ok(J.nzval, ref["out"], np.min(np.abs(_oracle_eps(x, colors, fdtype))), 5.0, "host path " + fdtype)


@pytest.mark.parametrize("fdtype,ncalls", [("forward", 4), ("central", 6), ("complex", 3)])
def test_dense_J_sparse_pattern(oracle, fdtype, ncalls):
    # test/coloring_tests.jl:51-70
    N = 30
    x = np.random.default_rng(12).random(N)
    colors = np.tile([1, 2, 3], 10)
    colptr, rowval = P.tridiag_csc(N)
    sp = fd.SparseMatrixCSC(N, N, colptr, rowval)
    J = torch.full((N, N), float("nan"), dtype=torch.float64, device="cuda").t()  # column-major
    f = fd.BuiltinF("tridiag", N)
    fd.finite_difference_jacobian_b(J, f, _dev(x), fdtype, colorvec=colors, sparsity=sp)
    assert f.fcalls == ncalls
    ref = oracle.jacobian(fdtype, oracle.Fixture("tridiag", N), x, colors, kind=oracle.PAT_CSC_DENSEJ,
                          colptr=colptr, rowval=rowval)
    _tol_ok(J.cpu().numpy(), ref["out"], np.min(np.abs(_oracle_eps(x, colors, fdtype))), 4.0, "dense J " + fdtype)


@pytest.mark.parametrize("fdtype,ncalls", [("forward", 4), ("central", 6), ("complex", 3)])
def test_tridiagonal_type(oracle, fdtype, ncalls):
    # test/coloring_tests.jl:72-88,94-96
    N = 30
    x = np.random.default_rng(13).random(N)
    colors = np.tile([1, 2, 3], 10)
    J = fd.Tridiagonal(_dev(np.full(N - 1, np.nan)), _dev(np.full(N, np.nan)), _dev(np.full(N - 1, np.nan)))
    f = fd.BuiltinF("tridiag", N)
    fd.finite_difference_jacobian_b(J, f, _dev(x), fdtype, colorvec=colors)
    assert f.fcalls == ncalls
    colptr, rowval = P.tridiag_csc(N)
    ref = oracle.jacobian(fdtype, oracle.Fixture("tridiag", N), x, colors, kind=oracle.PAT_COO_TRIDIAG,
                          rows_index=rowval, cols_index=P.csc_cols(colptr))
    em = np.min(np.abs(_oracle_eps(x, colors, fdtype)))
    for got, want, nm in zip((J.dl, J.d, J.du), ref["out"], ("dl", "d", "du")):
        _tol_ok(got.cpu().numpy(), want, em, 4.0, "Tridiagonal." + nm)


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("family,nx,ny", [("lap5", 40, 30), ("lap5_nl", 38, 11), ("lap5", 6, 3), ("lap5_nl", 2, 9), ("lap5", 300, 200)])  # (the families' lazy launchers need an even nx)
def test_storing_launch_into_bandedblockbanded_data(oracle, fdtype, family, nx, ny):
    # fd_bbb_store (round 5): the 5-point families fill BandedBlockBandedMatrix data -- ny blocks of nx rows, (1,1)/(1,1) bandwidths, the
    # reference's own layout (test/coloring_tests.jl:109-115; ext/FiniteDiffBlockBandedMatricesExt.jl:16-42) -- in f!'s own launch:
    # bits of the decompression path (k_decompress_bbb after the lazy hand-over), oracle parity, 1 + C / 2 C evaluations, dir = -1,
    # columns without a colour (zeros), and an INVALID colouring (the store must not be taken: the plan refuses it)
    lay = P.BandedBlockBandedLayout(np.full(ny, nx), 1, 1, 1, 1)
    N = lay.N
    colors = lay.colors()
    C = int(colors.max())
    xh = np.random.default_rng(nx * 7 + ny).random(N) + 0.1
    x = _dev(xh)
    f = fd.BuiltinF(family, nx, ny)

    def run(cv, store, dir=True):
        J = fd.BandedBlockBandedMatrix(None, lay)
        plan = fd.make_plan(J, J, cv, fdtype)
        plan.set_lazy(f, store=store)
        out = _dev(np.full(lay.data_len, np.nan))
        plan.jacobian(f, x, [out], dir=dir)
        return out, plan

    a, pa = run(colors, True)
    b, pb = run(colors, False)
    assert pa.info(fd.lib.INFO_LAZY_STORE) == 1 and pb.info(fd.lib.INFO_LAZY_STORE) == 0
    assert pa.fcalls_last == pb.fcalls_last == (1 + C if fdtype == "forward" else 2 * C)
    assert not torch.isnan(a).any() and torch.equal(a.view(torch.int64), b.view(torch.int64))
    if family == "lap5" and fdtype == "forward":
        a2, _ = run(colors, True, dir=-1)
        b2, _ = run(colors, False, dir=-1)
        assert torch.equal(a2.view(torch.int64), b2.view(torch.int64))
    if nx * ny <= 2000:
        ref = oracle.jacobian(fdtype, oracle.Fixture(family, nx, ny), xh, colors, kind=oracle.PAT_BANDEDBLOCKBANDED,
                              blk_sizes=lay.blk_sizes, bl=1, bu=1, lam=1, mu=1, block_starts=lay.block_starts, block_strides=lay.block_strides,
                              out_len=lay.data_len)
        em = np.min(np.abs(_oracle_eps(xh, colors, fdtype)))
        _tol_ok(a.cpu().numpy(), ref["out"], em, 8.0, "BBB store " + family)
    # columns without a colour: zeros, everything else unchanged in kind
    cv = colors.copy()
    cv[[0, N // 2, N - 1]] = 0
    a3, _ = run(cv, True)
    b3, _ = run(cv, False)
    assert torch.equal(a3.view(torch.int64), b3.view(torch.int64))
    # an invalid colouring (every column the same colour): the plan does not allow the store
    bad = np.ones(N, dtype=np.int64)
    a4, p4 = run(bad, True)
    b4, _ = run(bad, False)
    assert p4.info(fd.lib.INFO_LAZY_STORE) == 0 and torch.equal(a4.view(torch.int64), b4.view(torch.int64))


@pytest.mark.parametrize("fdtype", ["forward", "central", "complex"])
@pytest.mark.parametrize("kind", ["bidiag_U", "bidiag_L", "diagonal", "symtridiag"])
def test_other_structured_types_of_the_generic_loop(oracle, kind, fdtype):
    # Bidiagonal / Diagonal / SymTridiagonal J (src/jacobians.jl:524-525 + src/iteration_utils.jl:25-32 through their setindex!): the
    # oracle's generic COO loop over the type's structural non-zeros.  SymTridiagonal is this library's EXTENSION (LinearAlgebra's
    # setindex! throws off the diagonal: the reference cannot fill one): ev[i] receives BOTH J[i+1,i] and J[i,i+1] in the loop's order --
    # the later colour's write stays (the rule is restated literally below; there is no reference behaviour to compare with)
    N = 257
    rng = np.random.default_rng(21)
    x = rng.random(N) + 0.2
    a, b = rng.random(N) + 0.5, rng.random(N) + 0.5
    at, bt = _dev(a), _dev(b)
    if kind == "bidiag_U":
        C = 2
        fn = lambda fx, xx: fx.copy_(at.to(xx.dtype) * xx * xx + bt.to(xx.dtype) * xx * torch.cat([xx[1:], torch.zeros_like(xx[:1])]))
        fnp = lambda fx, xx: fx.__setitem__(slice(None), a * xx * xx + b * xx * np.concatenate([xx[1:], np.zeros(1, xx.dtype)]))
        rows = np.concatenate([np.arange(N), np.arange(N - 1)]) + 1
        cols = np.concatenate([np.arange(N), np.arange(1, N)]) + 1
    elif kind == "bidiag_L":
        C = 2
        fn = lambda fx, xx: fx.copy_(at.to(xx.dtype) * xx * xx + bt.to(xx.dtype) * xx * torch.cat([torch.zeros_like(xx[:1]), xx[:-1]]))
        fnp = lambda fx, xx: fx.__setitem__(slice(None), a * xx * xx + b * xx * np.concatenate([np.zeros(1, xx.dtype), xx[:-1]]))
        rows = np.concatenate([np.arange(N), np.arange(1, N)]) + 1
        cols = np.concatenate([np.arange(N), np.arange(N - 1)]) + 1
    elif kind == "diagonal":
        C = 1
        fn = lambda fx, xx: fx.copy_(at.to(xx.dtype) * xx * xx * xx)
        fnp = lambda fx, xx: fx.__setitem__(slice(None), a * xx * xx * xx)
        rows = cols = np.arange(N) + 1
    else:
        C = 3
        fn = fnp = None
        colptr, rowval = P.tridiag_csc(N)
        rows, cols = rowval, P.csc_cols(colptr)
    colors = P.cyclic_colors(N, C)
    order = np.lexsort((rows, cols))            # findstructralnz order: column-major
    rows, cols = np.asarray(rows)[order], np.asarray(cols)[order]
    f = fd.BuiltinF("tridiag_nl", N) if kind == "symtridiag" else fd.TorchF(fn, N, N)
    of = oracle.Fixture("tridiag_nl", N) if kind == "symtridiag" else oracle.PyF(fnp, N, N)
    ref = oracle.jacobian(fdtype, of, x, colors, kind=oracle.PAT_COO_TRIDIAG, rows_index=rows, cols_index=cols)
    dl_w, d_w, du_w = ref["out"]
    nan = lambda n: _dev(np.full(n, np.nan))
    if kind.startswith("bidiag"):
        J = fd.Bidiagonal(nan(N), nan(N - 1), kind[-1])
        want = {"dv": d_w, "ev": du_w if kind[-1] == "U" else dl_w}
    elif kind == "diagonal":
        J = fd.Diagonal(nan(N))
        want = {"diag": d_w}
    else:
        J = fd.SymTridiagonal(nan(N), nan(N - 1))
        ev = np.full(N - 1, np.nan)
        for color in range(1, C + 1):               # the reference's loop order: colour by colour, entries in storage order
            for r, c in zip(rows, cols):
                if colors[c - 1] == color and r != c:
                    ev[min(r, c) - 1] = (dl_w if r > c else du_w)[min(r, c) - 1]
        want = {"dv": d_w, "ev": ev}
    fd.finite_difference_jacobian_b(J, f, _dev(x), fdtype, colorvec=colors)
    em = np.min(np.abs(_oracle_eps(x, colors, fdtype)))
    for nm, w in want.items():
        _tol_ok(getattr(J, nm).cpu().numpy(), w, em, 8.0, kind + "." + nm)
    # the cached form finds its plan again and gives the same bits
    cache = fd.JacobianCache(_dev(x), fdtype, colorvec=colors, sparsity=J)
    first = {nm: getattr(J, nm).clone() for nm in want}
    for _ in range(2):
        fd.finite_difference_jacobian_b(J, f, _dev(x), cache)
    assert all(torch.equal(getattr(J, nm), first[nm]) for nm in want)


@pytest.mark.parametrize("l,u,M,N", [(1, 1, 30, 30), (2, 1, 40, 37), (0, 3, 25, 31), (3, 0, 33, 20)])
def test_banded(oracle, l, u, M, N):
    # test/coloring_tests.jl:90-92 plus rectangular / asymmetric bands
    rng = np.random.default_rng(14)
    x = rng.random(N)
    colors = P.cyclic_colors(N, l + u + 1)
    A = rng.random((M, N))  # linear f(x) = A_band x so every band entry is exercised

    def band_mask():
        i, j = np.meshgrid(np.arange(M), np.arange(N), indexing="ij")
        return (i - j <= l) & (j - i <= u)

    Ab = A * band_mask()
    At = _dev(Ab)
    f = fd.TorchF(lambda fx, xx: fx.copy_((At.to(xx.dtype) @ xx)), M, N)
    data = torch.full((N, l + u + 1), float("nan"), dtype=torch.float64, device="cuda").t()
    J = fd.BandedMatrix(data, M, l, u)
    fd.finite_difference_jacobian_b(J, f, _dev(x), "forward", colorvec=colors)
    of = oracle.PyF(lambda fx, xx: fx.__setitem__(slice(None), Ab @ xx), M, N)
    ref = oracle.jacobian("forward", of, x, colors, M=M, kind=oracle.PAT_BANDED, l=l, u=u)
    got = data.cpu().numpy()
    inside = np.zeros_like(got, dtype=bool)
    for j in range(N):
        for k in range(l + u + 1):
            r = j - u + k
            inside[k, j] = 0 <= r < M
    em = np.min(np.abs(_oracle_eps(x, colors, "forward")))
    _tol_ok(got[inside], ref["out"][inside], em, float(np.abs(Ab @ x).max()) + 1.0, "banded")
    assert np.all(got[~inside] == 0.0)
    assert np.allclose(P.banded_to_dense(got, M, N, l, u), Ab, rtol=0, atol=1e-5)


@pytest.mark.parametrize("fdtype", ["forward", "complex"])
def test_stencil_blockbanded_and_sparse(oracle, fdtype):
    # test/coloring_tests.jl:99-119 : clamped 5-point stencil; block-banded J and sparse J agree
    nx = ny = 100
    N = nx * ny
    x = np.random.default_rng(15).random(N)
    colptr, rowval = P.lap5_csc(nx, ny)
    k = np.arange(N)
    colors9 = 3 * ((k // nx) % 3) + (k % nx) % 3 + 1
    f = fd.BuiltinF("clamp5", nx, ny)
    Js = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.zeros(rowval.size)))
    fd.finite_difference_jacobian_b(Js, f, _dev(x), fdtype, colorvec=colors9)
    refs = oracle.jacobian(fdtype, oracle.Fixture("clamp5", nx, ny), x, colors9, kind=oracle.PAT_CSC_COMMON,
                           colptr=colptr, rowval=rowval)
    em = np.min(np.abs(_oracle_eps(x, colors9, fdtype)))
    _tol_ok(Js.nzval.cpu().numpy(), refs["out"], em, 5.0, "stencil csc")

    lay = P.BlockBandedLayout(np.full(ny, nx), 1, 1)
    colorsbb = lay.colors()
    Jb = fd.BlockBandedMatrix(_dev(np.full(lay.data_len, np.nan)), lay)
    f2 = fd.BuiltinF("clamp5", nx, ny)
    fd.finite_difference_jacobian_b(Jb, f2, _dev(x), fdtype, colorvec=colorsbb)
    assert f2.fcalls == (301 if fdtype == "forward" else 300)
    refb = oracle.jacobian(fdtype, oracle.Fixture("clamp5", nx, ny), x, colorsbb, kind=oracle.PAT_BLOCKBANDED,
                           blk_sizes=lay.blk_sizes, bl=1, bu=1, block_starts=lay.block_starts,
                           block_strides=lay.block_strides, out_len=lay.data_len)
    emb = np.min(np.abs(_oracle_eps(x, colorsbb, fdtype)))
    gotb = Jb.data.cpu().numpy()
    _tol_ok(gotb, refb["out"], emb, 5.0, "stencil block-banded")
    at = lay.index_of(rowval - 1, P.csc_cols(colptr) - 1)
    assert np.linalg.norm(gotb[at] - Js.nzval.cpu().numpy()) <= 1.5e-8 * np.linalg.norm(gotb[at]) * 10


@pytest.mark.parametrize("fdtype", FDTYPES)
def test_stencil_bandedblockbanded(oracle, fdtype):
    # test/coloring_tests.jl:109-115 : BandedBlockBandedMatrix J (ext/FiniteDiffBlockBandedMatricesExt.jl:16-42) on the
    # device through the enumerated-entries plan, vs the oracle's restatement of the same store loop
    nx, ny = 40, 30
    N = nx * ny
    x = np.random.default_rng(16).random(N)
    lay = P.BandedBlockBandedLayout(np.full(ny, nx), 1, 1, 1, 1)
    colors = lay.colors()
    J = fd.BandedBlockBandedMatrix(_dev(np.full(lay.data_len, np.nan)), lay)
    f = fd.BuiltinF("clamp5", nx, ny)
    fd.finite_difference_jacobian_b(J, f, _dev(x), fdtype, colorvec=colors)
    assert f.fcalls == {"forward": 10, "central": 18, "complex": 9}[fdtype]
    ref = oracle.jacobian(fdtype, oracle.Fixture("clamp5", nx, ny), x, colors, kind=oracle.PAT_BANDEDBLOCKBANDED,
                          blk_sizes=lay.blk_sizes, bl=1, bu=1, lam=1, mu=1, block_starts=lay.block_starts,
                          block_strides=lay.block_strides, out_len=lay.data_len)
    got = J.data.cpu().numpy()
    _tol_ok(got, ref["out"], np.min(np.abs(_oracle_eps(x, colors, fdtype))), 5.0, "banded-block-banded " + fdtype)
    # and it agrees with the sparse J of the same problem (the reference's own assertion)
    colptr, rowval = P.lap5_csc(nx, ny)
    Js = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.zeros(rowval.size)))
    fd.finite_difference_jacobian_b(Js, fd.BuiltinF("clamp5", nx, ny), _dev(x), fdtype, colorvec=colors)
    rows, cols, dest = lay.entries()
    pos = dict(zip(zip(rows.tolist(), cols.tolist()), dest.tolist()))
    at = np.array([pos[(int(r), int(c))] for r, c in zip(rowval, P.csc_cols(colptr))])
    assert np.linalg.norm(got[at] - Js.nzval.cpu().numpy()) <= 1.5e-8 * np.linalg.norm(got[at]) * 10


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("case", ["uniform", "ragged", "wide_subbands", "none_and_chunks", "float32"])
def test_bandedblockbanded_structural_plan(oracle, fdtype, case):
    # fd_plan_create_bandedblockbanded (round 4): the plan holds block sizes, bandwidths and slab starts / strides -- no entry list --
    # and k_decompress_bbb finds every slot's row and block by arithmetic.  Same bits as the enumerated-entries plan of the same
    # matrix (round 3's form), and the oracle's restatement of ext/FiniteDiffBlockBandedMatricesExt.jl:16-42.
    rng = np.random.default_rng(41)
    if case == "ragged":
        blk, bl, bu, lam, mu = rng.integers(3, 40, size=37), 2, 1, 2, 3
    elif case == "wide_subbands":
        blk, bl, bu, lam, mu = np.full(25, 12), 0, 2, 5, 4
    else:
        blk, bl, bu, lam, mu = np.full(30, 40), 1, 1, 1, 1
    lay = P.BandedBlockBandedLayout(blk, bl, bu, lam, mu)
    N = lay.N
    colors = lay.colors()
    kw = {}
    if case == "none_and_chunks":
        colors = colors.copy()
        colors[[0, 7, N // 2, N - 1]] = 0
        kw = dict(scratch_bytes=4 * 2 * 16 * N)          # room for a few colours at a time
    dtype = np.float32 if case == "float32" else np.float64
    tdt = torch.float32 if case == "float32" else torch.float64
    x = torch.as_tensor(rng.random(N) + 0.1, dtype=tdt, device="cuda")
    A = torch.as_tensor(rng.random((N, 3)), dtype=tdt, device="cuda")

    def fn(fx, xx):     # some f! (the plan does not care whether the pattern fits it: same arithmetic both ways)
        fx.copy_(A[:, 0].to(xx.dtype) * xx ** 2 + A[:, 1].to(xx.dtype) * xx.roll(1) + A[:, 2].to(xx.dtype) * xx.roll(-3) * xx)

    outs = []
    for as_entries in (False, True):
        J = fd.BandedBlockBandedMatrix(None, lay, as_entries=as_entries)
        plan = fd.make_plan(J, J, colors, fdtype, dtype=dtype, **kw)
        assert plan.out_len(0) == lay.data_len
        out = torch.full((lay.data_len,), float("nan"), dtype=tdt, device="cuda")
        plan.jacobian(fd.TorchF(fn, N, N, dtype=dtype), x, [out])
        outs.append(out)
    assert not torch.isnan(outs[0]).any()
    assert torch.equal(outs[0], outs[1]), case
    if case in ("uniform", "ragged") and dtype == np.float64:
        nx = 40
        if case == "uniform":      # the reference's fixture on this layout, against the oracle
            f = fd.BuiltinF("clamp5", nx, 30)
            Jd = fd.BandedBlockBandedMatrix(_dev(np.full(lay.data_len, np.nan)), lay)
            xh = rng.random(N)
            fd.finite_difference_jacobian_b(Jd, f, _dev(xh), fdtype, colorvec=lay.colors())
            ref = oracle.jacobian(fdtype, oracle.Fixture("clamp5", nx, 30), xh, lay.colors(), kind=oracle.PAT_BANDEDBLOCKBANDED,
                                  blk_sizes=lay.blk_sizes, bl=1, bu=1, lam=1, mu=1, block_starts=lay.block_starts,
                                  block_strides=lay.block_strides, out_len=lay.data_len)
            _tol_ok(Jd.data.cpu().numpy(), ref["out"], np.min(np.abs(_oracle_eps(xh, lay.colors(), fdtype))), 5.0, "bbb structural " + fdtype)


@pytest.mark.parametrize("fdtype", FDTYPES)
def test_nonsquare_with_cache(oracle, fdtype):
    # test/coloring_tests.jl:124-159
    n = 4
    x0 = np.concatenate([np.arange(1, n + 1) + 0.5, np.arange(1, n + 1) + 1.5])
    A = np.zeros((n, 2 * n))
    A[np.arange(n), np.arange(n)] = 1
    A[np.arange(n), np.arange(n) + n] = 1
    colptr, rowval = P.csc_from_dense(A)
    colorvec = np.concatenate([np.full(n, 1), np.full(n, 2)])
    sp = fd.SparseMatrixCSC(n, 2 * n, colptr, rowval, _dev(np.zeros(rowval.size)))
    cache = fd.JacobianCache(_dev(x0.copy()), _dev(np.zeros(n)), _dev(np.zeros(n)), fdtype, sparsity=sp,
                             colorvec=colorvec)
    f = fd.BuiltinF("nonsquare", n)
    fd.finite_difference_jacobian_b(sp, f, _dev(x0), cache)
    assert f.fcalls == {"forward": 3, "central": 4, "complex": 2}[fdtype]
    x1, x2 = x0[:n], x0[n:]
    Jex = np.hstack([np.diag(2 * (x1 - 3) + x2), np.diag(x1 + 2 * (x2 + 4))])
    J2 = P.csc_to_dense(n, 2 * n, colptr, rowval, sp.nzval.cpu().numpy())
    assert np.linalg.norm(J2 - Jex) <= 1e-6 * np.linalg.norm(Jex)
    ref = oracle.jacobian(fdtype, oracle.Fixture("nonsquare", n), x0, colorvec, M=n, kind=oracle.PAT_CSC_COMMON,
                          colptr=colptr, rowval=rowval)
    _tol_ok(sp.nzval.cpu().numpy(), ref["out"], np.min(np.abs(_oracle_eps(x0, colorvec, fdtype))), 200.0, "nonsquare")


DENSE_CASES = [
    (lambda dx, x: dx.copy_(torch.stack([x[0] ** 2 + x[1] ** 2, x[0] + x[1]])), [5.0, 3.0], 2,
     [[1, 1], [1, 1]], [[10, 6], [1, 1]]),
    (lambda dx, x: dx.copy_(torch.stack([x[0] ** 2 + x[1] ** 2, x[0]])), [-3.0, 2.0], 2,
     [[1, 1], [1, 0]], [[-6, 4], [1, 0]]),
    (lambda dx, x: dx.copy_(torch.stack([x[0] ** 2 + x[1] ** 2 - x[0]])), [-3.0, 2.0], 1, [[1, 1]], [[-7, 4]]),
    (lambda dx, x: dx.copy_(torch.stack([x[0] ** 2 + x[1] ** 2 - x[0], x[0] * x[1], x[0] * x[2], x[0]])),
     [-3.0, 2.0, 13.3], 4, [[1, 1, 0], [1, 1, 0], [1, 0, 1], [1, 0, 0]],
     [[-7.0, 4.0, 0], [2.0, -3.0, 0.0], [13.3, 0.0, -3.0], [1.0, 0.0, 0.0]]),
    (lambda dx, x: dx.copy_(torch.stack([x[0] ** 2 + x[1] ** 2])), [5.0, 3.0], 1, [[1, 1]], [[10.0, 6.0]]),
]


@pytest.mark.parametrize("case", range(len(DENSE_CASES)))
def test_dense_matrix_sparsity_known_answers(case):
    # test/coloring_tests.jl:171-219 with a user f! written in torch (the TorchF launcher)
    fn, theta, M, sp, expected = DENSE_CASES[case]
    N = len(theta)
    J = np.full((M, N), np.nan)
    f = fd.TorchF(fn, M, N)
    cache = fd.JacobianCache(_dev(np.array(theta)), _dev(np.zeros(M)), _dev(np.zeros(M)), "forward",
                             sparsity=np.array(sp))
    fd.finite_difference_jacobian_b(J, f, np.array(theta), cache)
    E = np.array(expected, float)
    assert np.linalg.norm(J - E) <= 1.5e-8 * np.linalg.norm(E) * 10
    assert f.fcalls == N + 1


def test_poisoned_cache_and_x_untouched():
    # test/cache_reuse_tests.jl:64-83 : garbage in the cache arrays must not matter; x stays bitwise intact
    J_REF = np.array([[2.0, 0.0], [0.0, 3.0], [4.0, 0.0]])
    colptr, rowval = P.csc_from_dense(J_REF)
    sp = fd.SparseMatrixCSC(3, 2, colptr, rowval)

    def foo(y, x):
        y.copy_(torch.stack([2 * x[0], 3 * x[1], 4 * x[0]]))

    for fdtype in FDTYPES:
        cache = fd.JacobianCache(_dev(np.full(2, 1e10)), _dev(np.full(3, 1e10)),
                                 None if fdtype == "complex" else _dev(np.full(3, 1e10)), fdtype)
        x = _dev(np.array([1.0, 2.0]))
        x_orig = x.clone()
        J = np.zeros((3, 2))
        fd.finite_difference_jacobian_b(J, fd.TorchF(foo, 3, 2), x, cache, sparsity=sp, colorvec=np.array([1, 2]))
        assert np.allclose(J, J_REF, rtol=0, atol=1e-6), fdtype
        assert torch.equal(x, x_orig)


def test_colors_chunked_and_many(oracle):
    # scratch cap forces several colour chunks; C > 8 exercises the segmented epsilon reduction; C > 254 int32 colours
    N = 3000
    x = np.random.default_rng(21).random(N)
    colptr, rowval = P.tridiag_csc(N)
    for C, cap in ((3, 1), (12, 200_000), (300, 0)):
        colors = P.cyclic_colors(N, C)
        for fdtype in FDTYPES:
            J = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.full(rowval.size, np.nan)))
            plan = fd.make_plan(J, J, colors, fdtype, scratch_bytes=cap)
            f = fd.BuiltinF("tridiag_nl", N)
            plan.jacobian(f, _dev(x), [J.nzval])
            if cap:
                assert plan.info(fd.lib.INFO_NCHUNKS) > 1
            ref = oracle.jacobian(fdtype, oracle.Fixture("tridiag_nl", N), x, colors, kind=oracle.PAT_CSC_COMMON,
                                  colptr=colptr, rowval=rowval)
            assert f.fcalls == ref["fcalls"]
            eps = plan.epsilons()
            assert np.allclose(eps, _oracle_eps(x, colors, fdtype), rtol=1e-12, atol=0)
            _tol_ok(J.nzval.cpu().numpy(), ref["out"], np.min(np.abs(eps)), 5.0, "C=%d %s" % (C, fdtype))


def test_uncoloured_columns_are_zero_filled(oracle):
    # colour < 1 => column never perturbed; fill_matrix! leaves its stored entries at 0
    N = 50
    x = np.random.default_rng(3).random(N)
    colors = P.cyclic_colors(N, 3)
    colors[[4, 17]] = 0
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.full(rowval.size, np.nan)))
    fd.finite_difference_jacobian_b(J, fd.BuiltinF("tridiag", N), _dev(x), "forward", colorvec=colors)
    ref = oracle.jacobian("forward", oracle.Fixture("tridiag", N), x, colors, kind=oracle.PAT_CSC_COMMON,
                          colptr=colptr, rowval=rowval)
    _tol_ok(J.nzval.cpu().numpy(), ref["out"], 1.5e-8, 4.0, "uncoloured")
    cols = P.csc_cols(colptr)
    assert np.all(J.nzval.cpu().numpy()[(cols == 5) | (cols == 18)] == 0.0)


def test_dir_and_steps(oracle):
    # dir = -1 and explicit relstep/absstep (src/epsilons.jl:26-29)
    N = 300
    x = np.random.default_rng(8).random(N) + 0.5
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.zeros(rowval.size)))
    cache = fd.JacobianCache(_dev(x), "forward", colorvec=colors, sparsity=J)
    fd.finite_difference_jacobian_b(J, fd.BuiltinF("tridiag_nl", N), _dev(x), cache, relstep=1e-6, absstep=1e-9, dir=-1)
    eps = cache.last_plan.epsilons()
    assert np.all(eps < 0)
    ref = oracle.jacobian("forward", oracle.Fixture("tridiag_nl", N), x, colors, kind=oracle.PAT_CSC_COMMON,
                          colptr=colptr, rowval=rowval, relstep=1e-6, absstep=1e-9, dir=-1.0)
    _tol_ok(J.nzval.cpu().numpy(), ref["out"], np.min(np.abs(eps)), 6.0, "dir=-1")


def test_config3_shape_lap5_central(oracle):
    # BASELINE config 3 shape at a size the oracle finishes quickly: 5-point Laplacian, 5 colours, central
    nx, ny = 400, 250
    N = nx * ny
    x = np.random.default_rng(3).random(N)
    colptr, rowval = P.lap5_csc(nx, ny)
    colors = P.lap5_colors(nx, ny)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.zeros(rowval.size)))
    f = fd.BuiltinF("lap5", nx, ny)
    fd.finite_difference_jacobian_b(J, f, _dev(x), "central", colorvec=colors)
    assert f.fcalls == 10
    ref = oracle.jacobian("central", oracle.Fixture("lap5", nx, ny), x, colors, kind=oracle.PAT_CSC_COMMON,
                          colptr=colptr, rowval=rowval)
    _tol_ok(J.nzval.cpu().numpy(), ref["out"], np.min(np.abs(_oracle_eps(x, colors, "central"))), 8.0, "lap5")
    got = J.nzval.cpu().numpy()
    isdiag = rowval == P.csc_cols(colptr)
    assert np.allclose(got[isdiag], -4.0, rtol=0, atol=1e-8) and np.allclose(got[~isdiag], 1.0, rtol=0, atol=1e-8)


def test_config5_shape_blockbanded_complex(oracle):
    # BASELINE config 5 shape (reduced): block-tridiagonal dense 32x32 blocks, complex step, 96 colours
    nb, bs = 60, 32
    N = nb * bs
    x = np.random.default_rng(5).random(N)
    lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
    colors = lay.colors()
    assert colors.max() == 96
    Jb = fd.BlockBandedMatrix(_dev(np.full(lay.data_len, np.nan)), lay)
    f = fd.BuiltinF("blockcoupled", nb, bs)
    fd.finite_difference_jacobian_b(Jb, f, _dev(x), "complex", colorvec=colors)
    assert f.fcalls == 96
    ref = oracle.jacobian("complex", oracle.Fixture("blockcoupled", nb, bs), x, colors, kind=oracle.PAT_BLOCKBANDED,
                          blk_sizes=lay.blk_sizes, bl=1, bu=1, block_starts=lay.block_starts,
                          block_strides=lay.block_strides, out_len=lay.data_len)
    got = Jb.data.cpu().numpy()
    # complex step has no subtractive cancellation: a few ulp of |J| (sum order of sigma differs)
    assert np.max(np.abs(got - ref["out"])) <= 1e-12 * max(1.0, np.max(np.abs(ref["out"])))


def test_column_window_matches_full(oracle):
    # multi-GPU sharding primitive: windows of columns reproduce the slices of the full result bit for bit
    N = 10007
    x = np.random.default_rng(9).random(N)
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    Jfull = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.zeros(rowval.size)))
    full = fd.make_plan(Jfull, Jfull, colors, "forward")
    full.jacobian(fd.BuiltinF("tridiag_nl", N), _dev(x), [Jfull.nzval])
    want = Jfull.nzval.cpu().numpy()
    cuts = [0, 2500, 2501, 7000, N]
    for a, b in zip(cuts[:-1], cuts[1:]):
        plan = fd.make_plan(Jfull, Jfull, colors, "forward", col_window=(a, b), x_window=(max(a - 2, 0), min(b + 2, N)))
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(fd.BuiltinF("tridiag_nl", N), _dev(x), [out])
        e0 = plan.info(fd.lib.INFO_ENTRY_BEGIN)
        assert e0 == colptr[a] - 1 and plan.out_len(0) == colptr[b] - colptr[a]
        assert np.array_equal(out.cpu().numpy(), want[e0:e0 + plan.out_len(0)])


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("case", ["csc_window", "csc_list", "csc_sorted", "tridiagonal", "banded", "blockbanded", "densej"])
def test_colour_ownership_sums_to_full(monkeypatch, fdtype, case):
    # multi-GPU split by colour (fd_plan_opts.color_begin/end): each owner writes only its columns' stored values,
    # leaves the rest untouched, and the owners' outputs add up to the full result bit for bit -- with an f! that
    # is evaluated on full vectors (no row window), and fcalls = owned colours (+1 base evaluation for forward)
    from finitediff_jl_amd import sharded as S
    monkeypatch.delenv("FDJAC_WINDOW", raising=False)
    monkeypatch.delenv("FDJAC_SORTED", raising=False)
    N = 4097
    fam, prm, sp = "tridiag_nl", (N,), None
    colors = P.cyclic_colors(N, 3)
    colors[[7, 2048]] = 0                      # columns without colour: written (as 0) by the owner of colour 1
    if case.startswith("csc"):
        if case == "csc_sorted":
            nx, ny = 96, 64
            N = nx * ny
            colptr, rowval = P.lap5_csc(nx, ny)
            colors = P.lap5_colors(nx, ny)
            fam, prm = "lap5", (nx, ny)
            monkeypatch.setenv("FDJAC_WINDOW", "0")
            monkeypatch.setenv("FDJAC_SORTED", "1")
        else:
            colptr, rowval = P.tridiag_csc(N)
            if case == "csc_list":
                monkeypatch.setenv("FDJAC_WINDOW", "0")
        J = fd.SparseMatrixCSC(N, N, colptr, rowval)
        sp = J
    elif case == "tridiagonal":
        J = fd.Tridiagonal(None, torch.empty(N, dtype=torch.float64, device="cuda"), None)
    elif case == "banded":
        J = fd.BandedMatrix(torch.empty((N, 3), dtype=torch.float64, device="cuda").t(), N, 1, 1)
    elif case == "blockbanded":
        nb, bs = 40, 8
        N = nb * bs
        lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
        colors = lay.colors()
        J = fd.BlockBandedMatrix(None, lay)
        sp = J
        fam, prm = "blockcoupled", (nb, bs)
    else:
        N = 60
        colors = P.cyclic_colors(N, 3)
        colptr, rowval = P.tridiag_csc(N)
        J = torch.empty((N, N), dtype=torch.float64, device="cuda").t()
        sp = fd.SparseMatrixCSC(N, N, colptr, rowval)
        prm = (N,)
    C = int(colors.max())
    x = _dev(np.random.default_rng(51).random(N))
    full_plan = fd.make_plan(J, sp, colors, fdtype)
    nouts = full_plan.nouts
    full = [_dev(np.full(full_plan.out_len(k), np.nan)) for k in range(nouts)]
    full_plan.jacobian(fd.BuiltinF(fam, *prm), x, full)
    world = 3 if C >= 3 else 2
    cuts = S.partition_colors(colors, world)
    assert cuts[0] == 0 and cuts[-1] == C and np.all(np.diff(cuts) >= 1)
    acc = [torch.zeros_like(t) for t in full]
    for r in range(world):
        plan = fd.make_plan(J, sp, colors, fdtype, color_range=(cuts[r], cuts[r + 1]))
        out = [_dev(np.zeros(plan.out_len(k))) for k in range(nouts)]
        f = fd.BuiltinF(fam, *prm)
        plan.jacobian(f, x, out)
        owned = int(cuts[r + 1] - cuts[r])
        assert f.fcalls == owned * (2 if fdtype == "central" else 1) + (1 if fdtype == "forward" else 0)
        for a, o in zip(acc, out):
            a += o
    for a, w in zip(acc, full):
        assert not torch.isnan(w).any()
        assert torch.equal(a, w)


def test_large_properties_headline_size():
    # BASELINE config 2/4 sizes: N = 10^6 (exact), property checks that do not need the oracle:
    # linear fixture => J is the constant stencil regardless of x; call count 1 + C; x untouched.
    N = 10 ** 6
    x = _dev(np.random.default_rng(2).random(N))
    xc = x.clone()
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, torch.full((rowval.size,), float("nan"), dtype=torch.float64,
                                                            device="cuda"))
    f = fd.BuiltinF("tridiag", N)
    fd.finite_difference_jacobian_b(J, f, x, "forward", colorvec=colors)
    assert f.fcalls == 4 and torch.equal(x, xc)
    got = J.nzval.cpu().numpy()
    isdiag = rowval == P.csc_cols(colptr)
    assert np.max(np.abs(got[isdiag] + 2.0)) < 5e-8 and np.max(np.abs(got[~isdiag] - 1.0)) < 5e-8


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("N,C,cap", [(1, 1, 0), (2, 2, 0), (1001, 3, 0), (4096, 3, 0), (3000, 12, 100_000), (5000, 300, 0)])
def test_lazy_points_bit_identical(fdtype, N, C, cap):
    # fd_f_launch_lazy: perturbing while loading must reproduce the materialised path bit for bit,
    # including the fused base evaluation, colour chunking and int32 colours
    x = _dev(np.random.default_rng(N + C).random(N))
    colors = P.cyclic_colors(N, C)
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    outs = []
    calls = []
    for lazy in (False, True, "pairs"):   # materialised | lazy (complex step: imaginary parts only) | lazy (re,im) pairs
        plan = fd.make_plan(J, J, colors, fdtype, scratch_bytes=cap)
        f = fd.BuiltinF("tridiag_nl", N)
        if lazy:
            plan.set_lazy(f, imag_only=(lazy is True))
        out = _dev(np.full(rowval.size, np.nan))
        plan.jacobian(f, x, [out])
        outs.append(out.cpu().numpy())
        calls.append((f.fcalls, plan.fcalls_last))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert calls[0] == calls[1] == calls[2]


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("family", ["lap5", "clamp5"])
def test_lazy_points_stencil_bit_identical(oracle, fdtype, family):
    nx, ny = 64, 37
    N = nx * ny
    xh = np.random.default_rng(77).random(N)
    x = _dev(xh)
    colors = P.lap5_colors(nx, ny)
    colptr, rowval = P.lap5_csc(nx, ny)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    outs = []
    for lazy in (False, True, "pairs"):   # materialised | lazy (complex step: imaginary parts only) | lazy (re,im) pairs
        plan = fd.make_plan(J, J, colors, fdtype)
        f = fd.BuiltinF(family, nx, ny)
        if lazy:
            plan.set_lazy(f, imag_only=(lazy is True))
        out = _dev(np.full(rowval.size, np.nan))
        plan.jacobian(f, x, [out])
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    ref = oracle.jacobian(fdtype, oracle.Fixture(family, nx, ny), xh, colors, kind=oracle.PAT_CSC_COMMON,
                          colptr=colptr, rowval=rowval)
    _tol_ok(outs[1], ref["out"], np.min(np.abs(_oracle_eps(xh, colors, fdtype))), 8.0, "lazy " + family)


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("lazy", [False, True, "pairs"])
@pytest.mark.parametrize("case", ["bs32", "bs8_none", "bs6_window", "mixed_even", "mixed_odd"])
def test_blockbanded_row_pair_kernel_bit_identical(monkeypatch, oracle, fdtype, lazy, case):
    # k_decompress_colrange_wg<VEC>: columns whose first row, row count and destination are all even are processed as
    # row pairs (16-B accesses); the scalar instantiation (unaligned output, odd layouts) must give the same bits
    sizes = {"bs32": np.full(12, 32), "bs8_none": np.full(30, 8), "bs6_window": np.full(20, 6),
             "mixed_even": np.array([4, 8, 2, 6, 10, 4, 4, 12]), "mixed_odd": np.array([4, 7, 2, 6, 9, 4, 5, 12])}[case]
    lay = P.BlockBandedLayout(sizes, 1, 1)
    N = int(sizes.sum())
    colors = lay.colors()
    if case == "bs8_none":
        colors = colors.copy()
        colors[[1, 40, N - 2]] = 0
    win = (12, N - 18) if case == "bs6_window" else None
    uniform = len(set(sizes.tolist())) == 1
    x = _dev(np.random.default_rng(62).random(N) - 0.3)
    Jb = fd.BlockBandedMatrix(None, lay)
    if uniform:
        f = fd.BuiltinF("blockcoupled", len(sizes), int(sizes[0]))
    else:
        if lazy:
            pytest.skip("user f! in torch has no lazy launcher")
        blk = torch.as_tensor(np.repeat(np.arange(len(sizes)), sizes), device="cuda")

        def fn(fx, xv):   # block sums couple neighbouring blocks: block-tridiagonal with dense blocks
            sig = torch.zeros(len(sizes), dtype=xv.dtype, device=xv.device).index_add_(0, blk, xv)
            S = sig.clone()
            S[1:] += sig[:-1]
            S[:-1] += sig[1:]
            fx.copy_(xv * S[blk] + torch.sin(xv))
        f = fd.TorchF(fn, N, N)
    outs = []
    for unaligned in (False, True):
        # the row-pair instantiation needs a 16-byte aligned output: an output that starts one element later takes the scalar
        # instantiation of the same kernel (what layouts with odd rows / destinations always take)
        monkeypatch.setenv("FDJAC_LAZY_STORE", "0")
        plan = fd.make_plan(Jb, Jb, colors, fdtype, col_window=win)
        if lazy:
            plan.set_lazy(f, imag_only=(lazy is True))
        buf = _dev(np.full(plan.out_len(0) + 1, np.nan))
        out = buf[1:] if unaligned else buf[:-1]
        plan.jacobian(f, x, [out])
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1], equal_nan=True)
    if win is None:
        assert not np.isnan(outs[0]).any()
        if uniform:
            ref = oracle.jacobian(fdtype, oracle.Fixture("blockcoupled", len(sizes), int(sizes[0])), x.cpu().numpy(), colors,
                                  kind=oracle.PAT_BLOCKBANDED, blk_sizes=lay.blk_sizes, bl=1, bu=1, block_starts=lay.block_starts,
                                  block_strides=lay.block_strides, out_len=lay.data_len)
            if fdtype == "complex":
                assert np.max(np.abs(outs[0] - ref["out"])) <= 1e-12 * max(1.0, np.max(np.abs(ref["out"])))
            else:
                _tol_ok(outs[0], ref["out"], np.min(np.abs(_oracle_eps(x.cpu().numpy(), colors, fdtype))), 40.0, "bb pairs " + case)


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("case", ["c5_shape", "bs5_cyclic", "bs64", "bs1", "none", "chunked", "declined", "window"])
def test_lazy_points_blockcoupled_bit_identical(fdtype, case):
    # the block-coupled family's lazy launcher (sigma of every point formed on chip) vs perturb + sigma + apply kernels
    nb, bs = {"c5_shape": (40, 32), "bs5_cyclic": (25, 5), "bs64": (9, 64), "bs1": (50, 1), "declined": (700, 1)}.get(case, (30, 8))
    N = nb * bs
    lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
    colors = lay.colors()
    if case == "bs5_cyclic":
        colors = P.cyclic_colors(N, 15)          # a colouring with a different structure (still valid: 3 blocks x 5)
    if case == "declined":
        colors = np.arange(1, N + 1)              # 700 colours: one batch does not fit the launcher's LDS -> it declines
    if case == "none":
        colors = colors.copy()
        colors[[0, 9, N - 1]] = 0
    cap = 40_000 if case == "chunked" else 0
    win = (2 * bs + 0, N - bs) if case == "window" else None
    x = _dev(np.random.default_rng(61).random(N) - 0.3)
    Jb = fd.BlockBandedMatrix(None, lay)
    outs, calls = [], []
    for lazy in (False, True, "pairs"):   # materialised | lazy (complex step: imaginary parts only) | lazy (re,im) pairs
        plan = fd.make_plan(Jb, Jb, colors, fdtype, scratch_bytes=cap, col_window=win)
        f = fd.BuiltinF("blockcoupled", nb, bs)
        if lazy:
            plan.set_lazy(f, imag_only=(lazy is True))
        if case == "chunked":
            assert plan.info(fd.lib.INFO_NCHUNKS) > 1
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(f, x, [out])
        outs.append(out.cpu().numpy())
        calls.append(f.fcalls)
    assert not np.isnan(outs[0]).any()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert calls[0] == calls[1] == calls[2]


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("case", ["tridiag", "tridiag_chunked", "tridiag_none", "lap5"])
def test_sorted_gather_kernel_bit_identical(monkeypatch, fdtype, case):
    # the LDS-transposed (colour-sorted) decompression must equal the storage-order kernel bit for bit
    if case == "lap5":
        nx, ny = 128, 96
        N = nx * ny
        colptr, rowval = P.lap5_csc(nx, ny)
        colors = P.lap5_colors(nx, ny)
        fam, prm, cap = "lap5", (nx, ny), 0
    else:
        N = 9001
        colptr, rowval = P.tridiag_csc(N)
        colors = P.cyclic_colors(N, 3)
        fam, prm = "tridiag_nl", (N,)
        cap = 300_000 if case == "tridiag_chunked" else 0
        if case == "tridiag_none":
            colors[[0, 77, 4096, N - 1]] = 0
    x = _dev(np.random.default_rng(31).random(N))
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    outs = []
    monkeypatch.setenv("FDJAC_WINDOW", "0")
    for forced in ("0", "1"):
        monkeypatch.setenv("FDJAC_SORTED", forced)
        plan = fd.make_plan(J, J, colors, fdtype, scratch_bytes=cap)
        assert plan.info(fd.lib.INFO_SORTED_GATHER) == int(forced)
        assert plan.info(fd.lib.INFO_WINDOW) == 0
        out = _dev(np.full(rowval.size, np.nan))
        plan.jacobian(fd.BuiltinF(fam, *prm), x, [out])
        outs.append(out.cpu().numpy())
    assert not np.isnan(outs[0]).any()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("case", ["tridiag", "tridiag_chunked", "tridiag_none", "tridiag_f_in", "band5", "bidiag_window",
                                  "lap5", "lap5_chunked", "lap5_small_tile", "lap5_2d", "lap5_2d_chunked", "lap5_2d_window",
                                  "lap5_2d_odd"])
def test_row_window_kernel_bit_identical(monkeypatch, fdtype, case):
    # the row-window kernel (dense f! loads -> LDS -> entries) must equal the storage-order gather kernel bit for bit
    monkeypatch.delenv("FDJAC_SORTED", raising=False)
    monkeypatch.delenv("FDJAC_WINDOW", raising=False)
    N = 9001
    cap, c0, c1 = 0, None, None
    fam, prm = "tridiag_nl", (N,)
    if case.startswith("lap5"):        # three row windows per tile (rows k-nx, k, k+nx), or 2-D tiles (R grid rows x L columns)
        nx, ny = (128, 96) if case != "lap5_small_tile" else (1100, 12)   # 3 x 5 x (2048/5+2) rows > LDS budget -> 1024-entry tiles
        if case == "lap5_2d_odd":
            nx, ny = 150, 77               # nx not a multiple of the tile width, ny not a multiple of the run count
        monkeypatch.setenv("FDJAC_WINDOW2D", "1" if "_2d" in case else "0")
        N = nx * ny
        colptr, rowval = P.lap5_csc(nx, ny)
        colors = P.lap5_colors(nx, ny)
        fam, prm = "lap5", (nx, ny)
        cap = 700_000 if case.endswith("_chunked") else 0
        if case in ("lap5", "lap5_2d"):
            colors[[5, 777, N - 3]] = 0
        if case == "lap5_2d_window":
            c0, c1 = 3 * nx + 17, N - 2 * nx - 5    # a column window that starts / ends inside grid rows
    elif case == "band5":
        colptr, rowval = P.banded_csc(N, N, 2, 2)
        colors = P.cyclic_colors(N, 5)
    elif case == "bidiag_window":      # odd column window start: the local rows begin at an odd row
        colptr, rowval = P.banded_csc(N, N, 1, 0)
        colors = P.cyclic_colors(N, 2)
        c0, c1 = 1235, 8000
    else:
        colptr, rowval = P.tridiag_csc(N)
        colors = P.cyclic_colors(N, 3)
        cap = 300_000 if case == "tridiag_chunked" else 0
        if case == "tridiag_none":
            colors[[0, 77, 4096, N - 1]] = 0
    xh = np.random.default_rng(37).random(N)
    x = _dev(xh)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    f_in = None
    if case == "tridiag_f_in":         # caller's f_in: unpadded, deliberately only 8-B aligned
        f_in = _dev(np.random.default_rng(38).random(N + 1))[1:]   # any values: both kernels see the same f_in
    outs = []
    for forced in ("0", "1"):
        monkeypatch.setenv("FDJAC_WINDOW", forced)
        plan = fd.make_plan(J, J, colors, fdtype, scratch_bytes=cap, col_window=(c0, c1) if c0 is not None else None)
        assert plan.info(fd.lib.INFO_WINDOW) == int(forced)
        if forced == "1":
            assert 100 <= plan.info(fd.lib.INFO_WIN_OVERREAD_X100) <= (125 if fam != "lap5" else 400)
            assert plan.info(fd.lib.INFO_WINDOW2D) == int("_2d" in case)
        out = _dev(np.full(plan.out_len(0), np.nan))
        f = fd.BuiltinF(fam, *prm)
        plan.jacobian(f, x, [out], f_in=f_in)
        outs.append(out.cpu().numpy())
    assert not np.isnan(outs[0]).any()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("case", ["n1", "n2", "n3", "n1023", "n1024", "n1025", "n5003", "none", "chunked", "f_in",
                                  "win_even", "win_odd", "win_tail", "c4", "c5_fallback"])
def test_tridiagonal_window_kernel_bit_identical(monkeypatch, fdtype, case):
    # Tridiagonal J: the row-window kernel (dense loads -> LDS -> three dense diagonals) vs the gather kernel
    N = {"n1": 1, "n2": 2, "n3": 3, "n1023": 1023, "n1024": 1024, "n1025": 1025}.get(case, 5003)
    C = 4 if case == "c4" else 5 if case == "c5_fallback" else min(3, N)
    colors = P.cyclic_colors(N, C)
    if case == "none":
        colors[[0, 1, 1024, 2047, 2048, N - 1]] = 0
    cap = 120_000 if case == "chunked" else 0
    win = {"win_even": (1024, 4000), "win_odd": (1025, 4001), "win_tail": (2048, N)}.get(case)
    x = _dev(np.random.default_rng(41).random(N))
    f_in = _dev(np.random.default_rng(42).random(N + 1))[1:] if case == "f_in" else None
    outs = []
    for forced in ("0", "1"):
        monkeypatch.setenv("FDJAC_WINDOW", forced) if forced == "0" else monkeypatch.delenv("FDJAC_WINDOW")
        J = fd.Tridiagonal(None, torch.empty(N, dtype=torch.float64, device="cuda"), None)
        plan = fd.make_plan(J, None, colors, fdtype, scratch_bytes=cap, col_window=win,
                            x_window=(max(win[0] - 2, 0), min(win[1] + 2, N)) if win else None)
        if case == "chunked":
            assert plan.info(fd.lib.INFO_NCHUNKS) > 1
        # the variant is fixed when the plan is created and reported by the plan: row windows need <= 4 colours and an
        # even first column, so the two runs really compare k_decompress_tridiag with k_decompress_tridiag_window
        want_window = forced == "1" and C <= 4 and (win is None or win[0] % 2 == 0)
        assert plan.info(fd.lib.INFO_WINDOW) == int(want_window)
        o = [_dev(np.full(plan.out_len(k), np.nan)) for k in range(3)]
        plan.jacobian(fd.BuiltinF("tridiag_nl", N), x, o, f_in=f_in)
        outs.append(np.concatenate([t.cpu().numpy() for t in o]))
    assert not np.isnan(outs[0]).any()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("l,u,M,N", [(1, 1, 5003, 5003), (2, 1, 4100, 4097), (0, 3, 2500, 3100), (3, 0, 3300, 2000), (4, 4, 3000, 3000)])
def test_banded_window_kernel_bit_identical(monkeypatch, fdtype, l, u, M, N):
    # BandedMatrix J through the row-window kernel (implicit indices) vs the per-slot gather kernel
    colors = P.cyclic_colors(N, min(l + u + 1, 8)) if l + u + 1 <= 8 else P.cyclic_colors(N, l + u + 1)
    colors[[3, N // 2]] = 0
    x = _dev(np.random.default_rng(43).random(N))
    A = torch.as_tensor(np.random.default_rng(44).random((M, 3)), device="cuda")

    def fn(fx, xx):   # any f! with the right band: f_i = sum_k A[i,k] * x[clamp(i-1+k)]^2
        idx = torch.arange(M, device="cuda")
        acc = torch.zeros(M, dtype=xx.dtype, device="cuda")
        for k in range(3):
            acc = acc + A[:, k].to(xx.dtype) * xx[torch.clamp(idx - 1 + k, 0, N - 1)] ** 2
        fx.copy_(acc)

    outs = []
    for forced in ("0", "1"):
        monkeypatch.setenv("FDJAC_WINDOW", forced) if forced == "0" else monkeypatch.delenv("FDJAC_WINDOW")
        data = torch.full((N, l + u + 1), float("nan"), dtype=torch.float64, device="cuda").t()
        J = fd.BandedMatrix(data, M, l, u)
        plan = fd.make_plan(J, None, colors, fdtype)
        assert plan.info(fd.lib.INFO_WINDOW) == (int(forced) if l + u + 1 <= 8 else 0)
        plan.jacobian(fd.TorchF(fn, M, N), x, [data])
        outs.append(data.cpu().numpy().copy())
    assert not np.isnan(outs[0]).any()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("case", ["csc", "lap5_1d", "banded"])
def test_window_tile_sizes_bit_identical(monkeypatch, fdtype, case):
    # the 1-D row-window kernel has three tile sizes (2048 / 1024 / 512 stored entries = 4 / 2 / 1 entry pairs per
    # thread); the plan picks by launch size, FDJAC_WIN_TILE forces one: same bits from all of them
    monkeypatch.delenv("FDJAC_WINDOW", raising=False)
    monkeypatch.setenv("FDJAC_WINDOW2D", "0")
    N = 30011
    colors = P.cyclic_colors(N, 3)
    fam, prm = "tridiag_nl", (N,)
    if case == "csc":
        cp, rv = P.tridiag_csc(N)
        J = fd.SparseMatrixCSC(N, N, cp, rv, None)
        sp = J
    elif case == "lap5_1d":
        nx, ny = 150, 110
        N = nx * ny
        cp, rv = P.lap5_csc(nx, ny)
        J = fd.SparseMatrixCSC(N, N, cp, rv, None)
        sp = J
        colors = P.lap5_colors(nx, ny)
        fam, prm = "lap5", (nx, ny)
        monkeypatch.setenv("FDJAC_WINDOW", "1")
    else:
        J = fd.BandedMatrix(torch.zeros((N, 3), dtype=torch.float64, device="cuda").t(), N, 1, 1)
        sp = None
    x = _dev(np.random.default_rng(78).random(N) + 0.2)
    f = fd.BuiltinF(fam, *prm)
    res = []
    for tile in ("", "2048", "1024", "512"):
        if tile:
            monkeypatch.setenv("FDJAC_WIN_TILE", tile)
        else:
            monkeypatch.delenv("FDJAC_WIN_TILE", raising=False)
        plan = fd.make_plan(J, sp, colors, fdtype)
        assert plan.info(fd.lib.INFO_WINDOW) == 1 and plan.info(fd.lib.INFO_WINDOW2D) == 0
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(f, x, [out])
        res.append(out.cpu().numpy())
    assert not np.isnan(res[0]).any()
    for r in res[1:]:
        assert np.array_equal(res[0], r)


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("case", ["tridiag", "tridiag_none", "tridiag_window_chunked", "banded21", "banded33", "greedy", "f32"])
def test_periodic_entry_codes_bit_identical(monkeypatch, fdtype, case):
    # regular tiles (code[q + P] == code[q] + S) read only the head of their entry codes and compute the rest;
    # FDJAC_WIN_PERIODIC=0 keeps every code explicit: same bits, and the detected period is the pattern's
    monkeypatch.delenv("FDJAC_WINDOW", raising=False)
    dtype = np.float32 if case == "f32" else np.float64
    tdt = torch.float32 if case == "f32" else torch.float64
    N = 50021
    colors = P.cyclic_colors(N, 3)
    win, cap, want_period, fam = None, 0, 9, "tridiag_nl"
    if case in ("tridiag", "tridiag_none", "tridiag_window_chunked", "greedy", "f32"):
        cp, rv = P.tridiag_csc(N)
        J = fd.SparseMatrixCSC(N, N, cp, rv, None)
        sp = J
        if case == "tridiag_none":
            colors = colors.copy()
            colors[[5, 20000, N - 3]] = 0          # tiles with a column without colour stay explicit
        if case == "tridiag_window_chunked":
            win, cap = (1001, N - 2002), 1_000_000   # a column window, colours in chunks
        if case == "greedy":
            rng = np.random.default_rng(3)           # a valid but irregular colouring: 6 colours picked at random
            colors = np.empty(N, np.int64)
            for j in range(N):
                used = {colors[j - 1] if j > 0 else 0, colors[j - 2] if j > 1 else 0}
                colors[j] = rng.choice([c for c in range(1, 7) if c not in used])
            want_period = 0
    else:
        l, u = (2, 1) if case == "banded21" else (3, 3)
        C = l + u + 1
        colors = P.cyclic_colors(N, C)
        J = fd.BandedMatrix(torch.zeros((N, C), dtype=tdt, device="cuda").t(), N, l, u)
        sp = None
        want_period = C * C
        fam = "tridiag_nl"        # any f! works for a bit-identity check of the decompression
    x = torch.as_tensor(np.random.default_rng(79).random(N) + 0.2, dtype=tdt, device="cuda")
    f = fd.BuiltinF(fam, N, dtype=dtype)
    res = []
    for per in ("1", "0"):
        monkeypatch.setenv("FDJAC_WIN_PERIODIC", per)
        plan = fd.make_plan(J, sp, colors, fdtype, col_window=win, scratch_bytes=cap, dtype=dtype)
        assert plan.info(fd.lib.INFO_WINDOW) == 1
        assert plan.info(fd.lib.INFO_WIN_PERIOD) == (want_period if per == "1" else 0)
        if case == "tridiag_window_chunked":
            assert plan.info(fd.lib.INFO_NCHUNKS) > 1
        out = torch.full((plan.out_len(0),), float("nan"), dtype=tdt, device="cuda")
        plan.set_lazy(f)
        plan.jacobian(f, x, [out])
        res.append(out.cpu().numpy())
    assert not np.isnan(res[0]).any()
    assert np.array_equal(res[0], res[1])


def test_row_window_heuristic(monkeypatch):
    monkeypatch.delenv("FDJAC_SORTED", raising=False)
    monkeypatch.delenv("FDJAC_WINDOW", raising=False)
    monkeypatch.delenv("FDJAC_WINDOW2D", raising=False)
    N = 20000
    colptr, rowval = P.tridiag_csc(N)
    Jt = fd.SparseMatrixCSC(N, N, colptr, rowval)
    assert fd.make_plan(Jt, Jt, P.cyclic_colors(N, 3), "forward").info(fd.lib.INFO_WINDOW) == 1
    # 9 colours on a tridiagonal pattern: 3x more f! values than stored entries -> gathers are cheaper
    assert fd.make_plan(Jt, Jt, P.cyclic_colors(N, 9), "forward").info(fd.lib.INFO_WINDOW) == 0
    nx, ny = 400, 100
    colptr, rowval = P.lap5_csc(nx, ny)
    Jl = fd.SparseMatrixCSC(nx * ny, nx * ny, colptr, rowval)
    pl = fd.make_plan(Jl, Jl, P.lap5_colors(nx, ny), "central")
    # 5-point stencil: scattered gathers -> 2-D tiles (6 grid rows x 64 columns): every f! value loaded ~1.4 times
    assert pl.info(fd.lib.INFO_WINDOW) == 1 and pl.info(fd.lib.INFO_WINDOW2D) == 1
    assert 120 <= pl.info(fd.lib.INFO_WIN_OVERREAD_X100) <= 220
    monkeypatch.setenv("FDJAC_WINDOW2D", "0")     # storage-order tiles: three row windows per tile, ~3 loads per value
    pl1 = fd.make_plan(Jl, Jl, P.lap5_colors(nx, ny), "central")
    assert pl1.info(fd.lib.INFO_WINDOW) == 1 and pl1.info(fd.lib.INFO_WINDOW2D) == 0
    assert 250 <= pl1.info(fd.lib.INFO_WIN_OVERREAD_X100) <= 400
    monkeypatch.delenv("FDJAC_WINDOW2D")
    # a random pattern has no row locality at all: neither windows nor ... (stays on the gather kernels)
    rng = np.random.default_rng(3)
    A = np.zeros((3000, 3000))
    A[rng.integers(0, 3000, 20000), rng.integers(0, 3000, 20000)] = 1
    cpr, rvr = P.csc_from_dense(A)
    Jr = fd.SparseMatrixCSC(3000, 3000, cpr, rvr)
    cr = fd.matrix_colors(Jr)
    assert fd.make_plan(Jr, Jr, cr, "forward").info(fd.lib.INFO_WINDOW) == 0


def test_sorted_gather_heuristic(monkeypatch):
    monkeypatch.delenv("FDJAC_SORTED", raising=False)
    monkeypatch.setenv("FDJAC_WINDOW", "0")     # the choice between the two GATHER kernels
    N = 20000
    colptr, rowval = P.tridiag_csc(N)
    Jt = fd.SparseMatrixCSC(N, N, colptr, rowval)
    pt = fd.make_plan(Jt, Jt, P.cyclic_colors(N, 3), "forward")
    assert pt.info(fd.lib.INFO_SORTED_GATHER) == 0          # banded: storage order is already coherent
    nx, ny = 400, 100
    colptr, rowval = P.lap5_csc(nx, ny)
    Jl = fd.SparseMatrixCSC(nx * ny, nx * ny, colptr, rowval)
    pl = fd.make_plan(Jl, Jl, P.lap5_colors(nx, ny), "central")
    assert pl.info(fd.lib.INFO_SORTED_GATHER) == 1          # 5-point stencil: scattered
    assert pl.info(fd.lib.INFO_LINES_DIRECT_X100) > 1.5 * pl.info(fd.lib.INFO_LINES_SORTED_X100)


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("N", [1, 2, 1000, 100003])
def test_jvp_parity(oracle, fdtype, N):
    # finite_difference_jvp! (src/jvp.jl:238-274) on the device vs the oracle, same x, v
    rng = np.random.default_rng(N)
    x, v = rng.random(N), rng.random(N) - 0.5
    out = _dev(np.full(N, np.nan))
    cache = fd.JVPCache(_dev(x), fdtype)
    fd.finite_difference_jvp_b(out, fd.BuiltinF("tridiag_nl", N), _dev(x), _dev(v), cache)
    ref = oracle.jvp(fdtype, oracle.Fixture("tridiag_nl", N), x, v)
    assert abs(cache.last_epsilon - ref["eps"]) <= 1e-12 * abs(ref["eps"])
    _tol_ok(out.cpu().numpy(), ref["jvp"], ref["eps"], 5.0, "jvp %s N=%d" % (fdtype, N))
    # fd_jvp_async (enqueue only) gives the same bits; 8-B-aligned x / v take the scalar kernels
    out2 = _dev(np.full(N, np.nan))
    fd.finite_difference_jvp_b(out2, fd.BuiltinF("tridiag_nl", N), _dev(x), _dev(v), cache, sync=False)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    xo, vo, oo = _dev(np.concatenate([[0.0], x]))[1:], _dev(np.concatenate([[0.0], v]))[1:], _dev(np.full(N + 1, np.nan))[1:]
    fd.finite_difference_jvp_b(oo, fd.BuiltinF("tridiag_nl", N), xo, vo, cache)
    _tol_ok(oo.cpu().numpy(), ref["jvp"], ref["eps"], 5.0, "jvp unaligned %s N=%d" % (fdtype, N))
    # host arrays + f_in through the ABI's staging path
    if fdtype == "forward":
        xm, xp = np.concatenate([[0.0], x[:-1]]), np.concatenate([x[1:], [0.0]])
        fin = (xm - 2 * x) + xp + (x * x) * xp
        outh = np.zeros(N)
        f2 = fd.BuiltinF("tridiag_nl", N)
        fd.finite_difference_jvp_b(outh, f2, x, v, "forward", fin)
        assert f2.fcalls == 1
        refh = oracle.jvp("forward", oracle.Fixture("tridiag_nl", N), x, v, f_in=fin)
        _tol_ok(outh, refh["jvp"], refh["eps"], 5.0, "jvp host f_in")


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("family", ["tridiag", "tridiag_nl", "lap5", "clamp5"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_jvp_lazy_points_bit_identical(oracle, fdtype, family, dtype):
    # fd_f_launch_lazy_jvp: f! perturbs x + eps*v while loading; same bits as the materialised points, same call count
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    if family in ("lap5", "clamp5"):
        nx, ny = 260, 131
        N, prm = nx * ny, (nx, ny)
    else:
        N, prm = 40001, (40001,)
    rng = np.random.default_rng(5)
    x = torch.as_tensor(rng.random(N), dtype=tdt, device="cuda")
    v = torch.as_tensor(rng.random(N) - 0.5, dtype=tdt, device="cuda")
    res, calls, eps = [], [], []
    # (lazy launcher writing the finished quotient -- FD_LAZY_JVP_CAP_QUOTIENT, the default --, lazy launcher writing the
    #  values, materialised points)
    for lazy, quotient in ((True, True), (True, False), (False, False)):
        f = fd.BuiltinF(family, *prm, dtype=dtype)
        assert f.lazy_jvp_fn is not None and f.lazy_jvp_caps == fd.lib.LAZY_JVP_CAP_QUOTIENT
        out = torch.full((N,), float("nan"), dtype=tdt, device="cuda")
        cache = fd.JVPCache(x, fdtype, lazy=lazy, quotient=quotient)
        fd.finite_difference_jvp_b(out, f, x, v, cache)
        res.append(out.clone())
        calls.append(f.fcalls)
        eps.append(cache.last_epsilon)
        launches = f.counts()[0]
        assert launches == (1 if lazy else (2 if fdtype == "forward" else 1))
    assert not torch.isnan(res[0]).any()
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])
    assert calls[0] == calls[1] == calls[2] == 2 and eps[0] == eps[1] == eps[2]
    if dtype == np.float64 and family != "clamp5":
        ref = oracle.jvp(fdtype, oracle.Fixture(family, *prm), x.cpu().numpy(), v.cpu().numpy())
        _tol_ok(res[0].cpu().numpy(), ref["jvp"], ref["eps"], 8.0, "lazy jvp %s %s" % (family, fdtype))
    if fdtype == "forward":     # f_in given: no base evaluation, one point
        outs = []
        for lazy in (True, False):
            g = fd.BuiltinF(family, *prm, dtype=dtype)
            o = torch.full((N,), float("nan"), dtype=tdt, device="cuda")
            fd.finite_difference_jvp_b(o, g, x, v, fd.JVPCache(x, "forward", lazy=lazy), torch.ones_like(x))   # an arbitrary f_in
            outs.append(o)
            assert g.fcalls == 1
        assert torch.equal(outs[0], outs[1])
    # families without a lazy JVP launcher fall back silently
    assert fd.BuiltinF("blockcoupled", 10, 8).lazy_jvp_fn is None and fd.BuiltinF("lap5", 7, 9).lazy_jvp_fn is None


def test_jvp_reference_fixture_and_errors():
    # test/finitedifftests.jl:440-448 with a user f! in torch; :complex is rejected (src/jvp.jl:248-250)
    rng = np.random.default_rng(17)
    x, vdir = rng.random(2), rng.random(2)
    e = np.exp(x[0])
    J_ref = np.array([[-7 + x[1] ** 3, 3 * (3 + x[0]) * x[1] ** 2],
                      [e * x[1] * np.cos(1 - e * x[1]), e * np.cos(1 - e * x[1])]])

    def iipf(fv, xx):
        fv.copy_(torch.stack([(xx[0] + 3) * (xx[1] ** 3 - 7) + 18, torch.sin(xx[1] * torch.exp(xx[0]) - 1)]))

    for fdtype, tol in (("forward", 1e-6), ("central", 1e-8)):
        out = np.zeros(2)
        fd.finite_difference_jvp_b(out, fd.TorchF(iipf, 2, 2), x, vdir, fdtype)
        assert np.max(np.abs(out - J_ref @ vdir)) < tol
    with pytest.raises(ValueError):
        fd.finite_difference_jvp_b(np.zeros(2), fd.TorchF(iipf, 2, 2), x, vdir, "complex")


@pytest.mark.parametrize("fdtype", FDTYPES)
def test_dense_uncoloured_arm(oracle, fdtype):
    # sparsity === nothing (src/jacobians.jl:548-557, 590-598, 626-631): per-element eps, J[:, i] column by column
    N = 300
    x = np.random.default_rng(41).random(N) + 0.1
    J = torch.full((N, N), float("nan"), dtype=torch.float64, device="cuda").t()   # column-major
    f = fd.BuiltinF("tridiag_nl", N)
    fd.finite_difference_jacobian_b(J, f, _dev(x), fdtype)          # cache-less, colorvec = 1:N, no sparsity
    ref = oracle.jacobian(fdtype, oracle.Fixture("tridiag_nl", N), x, np.arange(1, N + 1))
    assert f.fcalls == ref["fcalls"] == {"forward": N + 1, "central": 2 * N, "complex": N}[fdtype]
    rel = fd.default_relstep(fdtype)
    eps_min = EPS64 if fdtype == "complex" else rel * 0.1
    _tol_ok(J.cpu().numpy(), ref["out"], eps_min, 5.0, "dense arm " + fdtype)


def test_dense_arm_reference_tolerances_and_config1():
    # test/finitedifftests.jl:455-462 with a torch f!; BASELINE config 1 (f = sin.(x), N = 1000) on the device
    x = np.random.default_rng(7).random(2)
    e = np.exp(x[0])
    J_ref = np.array([[-7 + x[1] ** 3, 3 * (3 + x[0]) * x[1] ** 2],
                      [e * x[1] * np.cos(1 - e * x[1]), e * np.cos(1 - e * x[1])]])

    def iipf(fv, xx):
        fv.copy_(torch.stack([(xx[0] + 3) * (xx[1] ** 3 - 7) + 18, torch.sin(xx[1] * torch.exp(xx[0]) - 1)]))

    for fdtype, tol in (("forward", 1e-6), ("central", 1e-8), ("complex", 1e-14)):
        J = np.zeros((2, 2))
        fd.finite_difference_jacobian_b(J, fd.TorchF(iipf, 2, 2), x, fdtype)
        assert np.max(np.abs(J - J_ref)) < tol, fdtype
    J = np.zeros((2, 2))
    fd.finite_difference_jacobian_b(J, fd.TorchF(iipf, 2, 2), x, "forward", dir=-1)
    assert np.max(np.abs(J - J_ref)) < 1e-6
    N = 1000
    xs = np.random.default_rng(1).random(N)
    Js = np.zeros((N, N), order="F")
    fs = fd.TorchF(lambda fv, xx: torch.sin(xx, out=fv), N, N)
    fd.finite_difference_jacobian_b(Js, fs, xs, "forward")
    assert fs.fcalls == N + 1
    assert np.max(np.abs(Js - np.diag(np.cos(xs)))) < 1e-6


def test_out_of_place_api(oracle):
    # finite_difference_jacobian (src/jacobians.jl:240-259, 277-429): J allocated like jac_prototype / sparsity
    N = 30
    x = np.random.default_rng(2).random(N)
    colors = np.tile([1, 2, 3], 10)
    colptr, rowval = P.tridiag_csc(N)
    sp = fd.SparseMatrixCSC(N, N, colptr, rowval)
    exact = np.diag(np.full(N, -2.0)) + np.diag(np.ones(N - 1), 1) + np.diag(np.ones(N - 1), -1)
    # dense, no sparsity: test/coloring_tests.jl:28-31 (31 f calls)
    f = fd.BuiltinF("tridiag", N)
    J = fd.finite_difference_jacobian(f, _dev(x))
    assert f.fcalls == 31 and np.linalg.norm(J.cpu().numpy() - exact) <= 1.5e-8 * np.linalg.norm(exact) * 4
    # sparse prototype: result type preserved (test/out_of_place_tests.jl:28-35), 4 f calls
    f = fd.BuiltinF("tridiag", N)
    Js = fd.finite_difference_jacobian(f, _dev(x), colorvec=colors, sparsity=sp, jac_prototype=sp)
    assert isinstance(Js, fd.SparseMatrixCSC) and f.fcalls == 4
    assert np.linalg.norm(P.csc_to_dense(N, N, colptr, rowval, Js.nzval.cpu().numpy()) - exact) <= 1e-7
    # sparsity without prototype -> dense zeros(size(sparsity)) (src/jacobians.jl:305-307), host arrays
    Jd = fd.finite_difference_jacobian(fd.BuiltinF("tridiag", N), x, "central", colorvec=colors, sparsity=sp)
    assert isinstance(Jd, np.ndarray) and np.linalg.norm(Jd - exact) <= 1e-8
    # cache reuse at a new x (test/cache_reuse_tests.jl:30-39)
    cache = fd.JacobianCache(_dev(np.zeros(N)), "forward", colorvec=colors, sparsity=sp)
    fd.finite_difference_jacobian(fd.BuiltinF("tridiag", N), _dev(np.zeros(N)), cache, jac_prototype=sp)
    J2 = fd.finite_difference_jacobian(fd.BuiltinF("tridiag", N), _dev(x), cache, jac_prototype=sp)
    assert np.linalg.norm(P.csc_to_dense(N, N, colptr, rowval, J2.nzval.cpu().numpy()) - exact) <= 1e-6


@pytest.mark.parametrize("fdtype", ["forward", "central"])
def test_small_fused_launch_is_a_plan_property(monkeypatch, oracle, fdtype):
    # FDJAC_SMALL is read at plan creation; the fused single-workgroup launch and the wide path agree with the oracle
    N = 3000
    colptr, rowval = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    xh = np.random.default_rng(72).random(N)
    ref = oracle.jacobian(fdtype, oracle.Fixture("tridiag_nl", N), xh, colors, kind=oracle.PAT_CSC_COMMON,
                          colptr=colptr, rowval=rowval)["out"]
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    for small in ("1", "0"):
        monkeypatch.setenv("FDJAC_SMALL", small)
        plan = fd.make_plan(J, J, colors, fdtype)
        assert plan.info(fd.lib.INFO_SMALL_FUSED) == int(small)
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(fd.BuiltinF("tridiag_nl", N), _dev(xh), [out])
        got = out.cpu().numpy()
        assert np.all(np.abs(got - ref) <= 1e-6 * np.abs(ref) + 1e-7)


@pytest.mark.parametrize("C,shift,N", [(3, 0, 100003), (3, 2, 40000), (5, 1, 70001), (8, 3, 33333), (2, 1, 4099 * 5), (4, 0, 2048 * 9 + 1)])
def test_eps_reduction_variants_bit_identical(monkeypatch, C, shift, N):
    # cyclic colourings: the step-size reduction computes the colours (FD_INFO_EPS_CYCLIC) instead of reading them;
    # (x is read with non-temporal loads when the hand-over path follows, plain loads otherwise).  Same values, same order: same bits,
    # and they equal the masked-norm rule restated in numpy (src/jacobians.jl:559-561) to 1e-13
    colors = ((np.arange(N) + shift) % C + 1).astype(np.int64)
    xh = np.random.default_rng(90 + C).random(N) * 3 - 1
    x = _dev(xh)
    colptr, rowval = P.banded_csc(N, N, C // 2, C - 1 - C // 2)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    got = {}
    for cyc in ("1", "0"):
        monkeypatch.setenv("FDJAC_EPS_CYCLIC", cyc)
        monkeypatch.setenv("FDJAC_SMALL", "0")
        plan = fd.make_plan(J, J, colors, "forward")
        assert plan.info(fd.lib.INFO_EPS_CYCLIC) == (C if cyc == "1" else 0)
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(fd.TorchF(lambda fx, xx: fx.copy_(xx * xx), N, N), x, [out])
        got[cyc] = plan.epsilons()
    ref = got["0"]
    for k, v in got.items():
        assert np.array_equal(v, ref), k
    assert np.allclose(ref, _oracle_eps(xh, colors, "forward"), rtol=1e-13, atol=0)
    # a colouring that is NOT cyclic falls back to reading the colours
    monkeypatch.setenv("FDJAC_EPS_CYCLIC", "1")
    c2 = colors.copy()
    c2[N // 2] = c2[N // 2] % C + 1
    assert fd.make_plan(J, J, c2, "forward").info(fd.lib.INFO_EPS_CYCLIC) == 0


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("coloring", ["valid", "invalid_two_colours"])
def test_device_matches_broadcast_accumulate_arm(oracle, fdtype, coloring):
    # the reference's arm for GPU arrays (`fast_jacobian_setindex!`, src/jacobians.jl:665-671: += over all
    # listed entries into a zeroed J).  The device path stores by assignment, which gives the accumulate arm's values for
    # any colouring (tests/test_oracle_golden.py::test_broadcast_accumulate_arm_equals_assignment); here the HIP result
    # is compared with the ACCUMULATE oracle directly, dense J + dense-matrix sparsity, valid and invalid colourings.
    M, N = 45, 40
    rng = np.random.default_rng(321)
    A = (rng.random((M, N)) < 0.12).astype(np.float64)
    A[np.arange(N), np.arange(N)] = 1.0
    W = rng.random((M, N)) * A
    Wd = _dev(W)
    rows, cols = oracle.findstructralnz_dense(A)
    colors = np.arange(1, N + 1, dtype=np.int64) if coloring == "valid" else (np.arange(N) % 2 + 1).astype(np.int64)
    xh = rng.random(N) + 0.1

    def fn_np(fx, x):
        fx[:] = W @ (x * x)

    def fn_t(fx, x):
        fx.copy_(Wd.to(x.dtype) @ (x * x))

    ref = oracle.jacobian(fdtype, oracle.PyF(fn_np, M, N), xh, colors, M, kind=oracle.PAT_COO_DENSEJ_ACCUM,
                          rows_index=rows, cols_index=cols)
    J = torch.full((N, M), float("nan"), dtype=torch.float64, device="cuda").t()      # column-major M x N
    f = fd.TorchF(fn_t, M, N)
    fd.finite_difference_jacobian_b(J, f, _dev(xh), fdtype, colorvec=colors, sparsity=A)
    assert f.fcalls == ref["fcalls"]
    eps_min = np.min(np.abs(_oracle_eps(xh, colors, fdtype)))
    _tol_ok(J.cpu().numpy(), ref["out"], eps_min, float(np.abs(W).sum(axis=1).max()) * 1.5, "accumulate arm " + fdtype + " " + coloring)


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("case", ["tridiag_nl", "tridiag_nl_small", "tridiag_nl_chunked", "tridiag_nl_none", "tridiag_nl_window", "tridiag_nl_owned",
                                  "tridiag_nl_gather", "tridiag_native", "banded", "lap5_nl", "lap5_nl_gather", "lap5_nl_chunked", "clamp5", "lap5"])
def test_lazy_differences_bit_identical(monkeypatch, fdtype, case):
    # FD_LAZY_CAP_DIFF: the lazy launcher hands over f(x + d) - f(x) / f(x + d) - f(x - d) (the subtraction of
    # src/jacobians.jl:565,607 moved into f!'s launch), the decompression divides by eps / 2 eps.  Same bits as storing the
    # values and subtracting in the decompression, for every kernel variant; same f! evaluation count.
    win = own = None
    cap = 0
    if case.startswith("tridiag") or case == "banded":
        N = 2001 if case == "tridiag_nl_small" else 70_001
        colptr, rowval = P.tridiag_csc(N)
        colors = P.cyclic_colors(N, 3)
        fam, prm = "tridiag_nl", (N,)
        if case == "tridiag_nl_chunked":
            cap = 8 * 2 * ((N + 31) // 32 * 32) * (2 if fdtype == "central" else 1) * 2      # two colours per chunk
        if case == "tridiag_nl_none":
            colors[[0, 77, 4096, N - 1]] = 0
        if case == "tridiag_nl_window":
            win = (N // 5 + 1, 4 * N // 5)
        if case == "tridiag_nl_owned":
            own = (1, 3)
        if case == "tridiag_nl_gather":
            monkeypatch.setenv("FDJAC_WINDOW", "0")
    else:
        nx, ny = 128, 96
        N = nx * ny
        colptr, rowval = P.lap5_csc(nx, ny)
        colors = P.lap5_colors(nx, ny)
        fam, prm = ("clamp5" if case == "clamp5" else "lap5" if case == "lap5" else "lap5_nl"), (nx, ny)
        if case == "lap5_nl_gather":
            monkeypatch.setenv("FDJAC_WINDOW", "0")
        if case == "lap5_nl_chunked":
            cap = 8 * 2 * N * (2 if fdtype == "central" else 1) * 2 + 4096
    x = _dev(np.random.default_rng(61).random(N))
    outs, calls = [], []
    for diff in (True, False, "env"):
        monkeypatch.setenv("FDJAC_LAZY_DIFF", "0") if diff == "env" else monkeypatch.delenv("FDJAC_LAZY_DIFF", raising=False)
        if case == "tridiag_native":
            plan = fd.make_plan(fd.Tridiagonal(None, np.empty(N), None), None, colors, fdtype)
        elif case == "banded":
            plan = fd.make_plan(fd.BandedMatrix(None, N, 1, 1), None, colors, fdtype)
        else:
            J = fd.SparseMatrixCSC(N, N, colptr, rowval)
            plan = fd.make_plan(J, J, colors, fdtype, scratch_bytes=cap, col_window=win, color_range=own,
                                x_window=(max(win[0] - 3, 1), min(win[1] + 3, N)) if win else None)
        if "chunked" in case:
            assert plan.info(fd.lib.INFO_NCHUNKS) > 1
        f = fd.BuiltinF(fam, *prm)
        assert f.lazy_caps & fd.lib.LAZY_CAP_DIFF
        plan.set_lazy(f, diff=(diff is not False))
        assert plan.info(fd.lib.INFO_LAZY_DIFF) == (1 if diff is True else 0)
        o = [_dev(np.full(plan.out_len(k), 0.0 if own else np.nan)) for k in range(plan.info(fd.lib.INFO_NOUTS))]
        plan.jacobian(f, x, o)
        outs.append([t.cpu().numpy() for t in o])
        calls.append((f.fcalls, plan.fcalls_last))
    for a, b, c in zip(*outs):
        assert not np.isnan(a).any()
        assert np.array_equal(a, b) and np.array_equal(a, c)
    assert calls[0] == calls[1] == calls[2]
    C = int(colors.max())
    ncol = (own[1] - own[0]) if own else C
    assert calls[0][0] == ncol * (2 if fdtype == "central" else 1) + (1 if fdtype == "forward" else 0)


def test_lazy_differences_respect_f_in_and_oracle(oracle):
    # a caller-supplied f_in is the subtrahend the reference uses (src/jacobians.jl:540-545): no differences from the
    # launcher then; without it the differences path must still match the oracle
    N = 50_001
    colptr, rowval = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    xh = np.random.default_rng(62).random(N)
    x = _dev(xh)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    f = fd.BuiltinF("tridiag_nl", N)
    plan = fd.make_plan(J, J, colors, "forward")
    plan.set_lazy(f)
    xp = np.concatenate([[0.0], xh, [0.0]])
    bogus = _dev(xp[:-2] - 2 * xp[1:-1] + xp[2:] + xp[1:-1] ** 2 * xp[2:] + 1.0)     # f(x) + 1, NOT f(x): the result must show it
    a, b = _dev(np.full(rowval.size, np.nan)), _dev(np.full(rowval.size, np.nan))
    plan.jacobian(f, x, [a])
    plan.jacobian(f, x, [b], f_in=bogus)
    ref = oracle.jacobian("forward", oracle.Fixture("tridiag_nl", N), xh, colors, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    em = np.min(np.abs(_oracle_eps(xh, colors, "forward")))
    _tol_ok(a.cpu().numpy(), ref["out"], em, 8.0, "lazy differences")
    assert not torch.equal(a, b)
    eps = np.abs(_oracle_eps(xh, colors, "forward"))
    assert np.all(np.abs((a - b).cpu().numpy()) > 0.5 / eps.max())     # every entry moved by ~ 1/eps


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("case", ["tridiag", "tridiag_window", "tridiag_devplan", "tridiag_chunked", "band5", "bidiag", "banded11", "banded23",
                                  "banded_rect", "tridiag_shifted", "tridiag_none", "tridiag_t2048", "tridiag_t512",
                                  "diag_csc", "diag_csc_oddwin", "diag_banded_oddwin", "bidiag_oddwin", "band5_oddwin", "banded23_evenwin"])
def test_band_descriptors_computed_bit_identical(monkeypatch, fdtype, case):
    # Uniform bands: the row-window kernel computes the descriptors of its regular tiles from the band parameters instead of
    # loading them (FD_INFO_BAND_DESC tiles; the plan verified that the stored descriptors are what the kernel computes).
    # FDJAC_BAND_DESC=0 loads every descriptor: same bits.
    N = M = 150_011
    l = u = 1
    win = None
    cap = 0
    banded = case.startswith("banded")
    if case == "band5":
        l = u = 2
    if case == "bidiag":
        l, u = 1, 0
    if case == "banded23":
        l, u = 2, 3
    if case == "banded_rect":
        l, u, M = 3, 1, N + 40
    if case.startswith("diag_"):                     # a diagonal band (w = 1): the one width whose "w - 2" is negative
        l = u = 0
        banded = "banded" in case
    if case == "bidiag_oddwin":
        l, u = 1, 0
    if case == "band5_oddwin":
        l = u = 2
    if case == "banded23_evenwin":
        l, u = 2, 3
    w = l + u + 1
    colors = P.cyclic_colors(N, w)
    if case == "tridiag_shifted":
        colors = ((np.arange(N) + 2) % 3 + 1).astype(np.int64)
    if case == "tridiag_none":
        colors[[5, N // 2]] = 0                      # not cyclic any more: every descriptor is loaded
    if case == "tridiag_chunked":
        cap = 8 * 2 * ((N + 31) // 32 * 32) * (2 if fdtype != "forward" else 1) * 2
    if case == "tridiag_window":
        win = (N // 5 + 1, 4 * N // 5)
    if case.endswith("_oddwin"):                     # first column of the window odd / even: the tiles' first entries move
        win = (17_793, 126_540)
    if case.endswith("_evenwin"):
        win = (17_794, 126_541)
    if case in ("tridiag_t2048", "tridiag_t512"):
        monkeypatch.setenv("FDJAC_WIN_TILE", case[-4:] if case.endswith("2048") else "512")
    x = _dev(np.random.default_rng(81).random(N))
    A = torch.as_tensor(np.random.default_rng(83).random((M, w)), device="cuda")

    def fn(fx, xx):
        idx = torch.arange(M, device="cuda")
        acc = torch.zeros(M, dtype=xx.dtype, device="cuda")
        for k in range(w):
            acc = acc + A[:, k].to(xx.dtype) * xx[torch.clamp(idx - l + k, 0, N - 1)] ** 2
        fx.copy_(acc)

    outs, infos = [], []
    for comp in ("1", "0"):
        monkeypatch.setenv("FDJAC_BAND_DESC", comp)
        monkeypatch.setenv("FDJAC_PLAN_DEVICE", "1" if case == "tridiag_devplan" else "0")
        if banded:
            plan = fd.make_plan(fd.BandedMatrix(None, M, l, u), None, colors, fdtype, col_window=win)
        else:
            colptr, rowval = P.banded_csc(M, N, l, u)
            J = fd.SparseMatrixCSC(M, N, colptr, rowval)
            plan = fd.make_plan(J, J, colors, fdtype, scratch_bytes=cap, col_window=win)
        assert plan.info(fd.lib.INFO_WINDOW) == 1
        infos.append(plan.info(fd.lib.INFO_BAND_DESC))
        out = _dev(np.full(plan.out_len(0), np.nan))
        if l == u == 1 and not banded:
            f = fd.BuiltinF("tridiag_nl", N)
            plan.set_lazy(f)
        else:
            f = fd.TorchF(fn, M, N)
        plan.jacobian(f, x, [out])
        outs.append(out.cpu().numpy())
    if case == "tridiag_none":
        assert infos == [0, 0]
    else:
        assert infos[0] > 10 and infos[1] == 0      # (all but the corner tiles)
    assert not np.isnan(outs[0]).any()
    assert np.array_equal(outs[0], outs[1])


def test_band_index_arithmetic_at_large_entry_counts(monkeypatch):
    # the computed descriptors divide entry numbers up to 2^31 by the band width with a multiply-shift
    # (fd_div31): a pattern whose entry numbers exceed 2^26 must give the bits of the loaded-descriptor path
    N = 23_000_003                                      # tridiagonal: 6.9e7 stored entries
    colptr, rowval = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    x = _dev(np.random.default_rng(91).random(N))
    f = fd.BuiltinF("tridiag_nl", N)
    outs = []
    for comp in ("1", "0"):
        monkeypatch.setenv("FDJAC_BAND_DESC", comp)
        plan = fd.make_plan(J, J, colors, "forward")
        assert (plan.info(fd.lib.INFO_BAND_DESC) > 0) == (comp == "1")
        out = _dev(np.full(rowval.size, np.nan))
        plan.set_lazy(f)
        plan.jacobian(f, x, [out])
        outs.append(out)
        del plan
    assert not torch.isnan(outs[0]).any()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_switch_combinations_bit_identical(monkeypatch, seed):
    # Every kernel variant / plan builder / hand-over is a different route to the same IEEE operations on the same operands.
    # Reference: the plain gather kernel on a host-built plan with the plain hand-over of the f! values.  Variant: a random
    # combination of the switches of DESIGN section 5 (tile size, periodic codes, computed descriptors, 2-D tiles, differences
    # hand-over, the storing launch and its variants, device plan builder, colour chunks, column window).  Same bits, same
    # number of f! evaluations.
    import os
    rng = np.random.default_rng(int(os.environ.get("FDJAC_TEST_SEED_BASE", "7000")) + seed)
    fdtype = FDTYPES[int(rng.integers(0, 3))]
    scale = float(os.environ.get("FDJAC_TEST_SIZE_SCALE", "1"))       # (exploratory runs also shrink the problems)
    if seed % 2 == 0:
        N = max(3, int(rng.integers(60_000, 330_000) * scale))
        colptr, rowval = P.tridiag_csc(N)
        C = 3
        colors = ((np.arange(N) + int(rng.integers(0, 3))) % 3 + 1).astype(np.int64)
        fam, prm, halo = "tridiag_nl", (N,), 3
    else:
        nx, ny = 2 * max(1, int(rng.integers(40, 260) * scale ** 0.5)), max(2, int(rng.integers(60, 400) * scale ** 0.5))
        N = nx * ny
        colptr, rowval = P.lap5_csc(nx, ny)
        C = 5
        colors = P.lap5_colors(nx, ny)
        fam, prm, halo = ("lap5_nl" if rng.random() < 0.6 else "clamp5"), (nx, ny), nx + 2
    if rng.random() < 0.25:
        colors = colors.copy()
        colors[rng.integers(0, N, size=4)] = 0
    win = None
    if rng.random() < 0.35 and N >= 12:
        a = int(rng.integers(0, N // 3))
        win = (a + 1, int(rng.integers(a + N // 3, N)))
    cap = 0
    if rng.random() < 0.3:      # two colours per chunk
        cap = 8 * 2 * ((N + 31) // 32 * 32) * (2 if fdtype != "forward" else 1) * 2 + 8192
    dtype = np.float32 if rng.random() < 0.3 else np.float64          # (fd32_*: the second instantiation of every kernel)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    x = torch.as_tensor(rng.random(N), dtype=tdt, device="cuda")
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    xwin = (max(win[0] - halo, 1), min(win[1] + halo, N)) if win else None
    if dtype == np.float32 and cap:
        cap //= 2

    def run(env, diff):
        for k in ("FDJAC_WINDOW", "FDJAC_SORTED", "FDJAC_WIN_TILE", "FDJAC_WIN_PERIODIC", "FDJAC_BAND_DESC",
                  "FDJAC_WINDOW2D", "FDJAC_PLAN_DEVICE", "FDJAC_LAZY_DIFF", "FDJAC_LAZY_STORE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        plan = fd.make_plan(J, J, colors, fdtype, scratch_bytes=cap, col_window=win, x_window=xwin, dtype=dtype)
        f = fd.BuiltinF(fam, *prm, dtype=dtype)
        plan.set_lazy(f, diff=diff)
        out = torch.full((plan.out_len(0),), float("nan"), dtype=tdt, device="cuda")
        plan.jacobian(f, x, [out])
        return out, f.fcalls

    ref, calls_ref = run({"FDJAC_WINDOW": "0", "FDJAC_SORTED": "0", "FDJAC_PLAN_DEVICE": "0"}, False)
    env = {}
    pick = lambda name, choices: env.__setitem__(name, str(choices[int(rng.integers(0, len(choices)))])) if rng.random() < 0.6 else None
    pick("FDJAC_WIN_TILE", [512, 1024, 2048])
    pick("FDJAC_WIN_PERIODIC", [0, 1])
    pick("FDJAC_BAND_DESC", [0, 1])
    pick("FDJAC_WINDOW2D", [0, 1])
    pick("FDJAC_PLAN_DEVICE", [0, 1])
    pick("FDJAC_SORTED", [0, 1])
    pick("FDJAC_LAZY_STORE", [0, 1])
    diff = bool(rng.random() < 0.7)
    got, calls = run(env, diff)
    assert not torch.isnan(ref).any()
    bad = torch.nonzero(got != ref).flatten()
    assert bad.numel() == 0, (fam, prm, fdtype, np.dtype(dtype).name, win, cap, env, diff, int(bad.numel()), bad[:8].tolist(),
                              got[bad[:8]].tolist(), ref[bad[:8]].tolist())
    assert calls == calls_ref


@pytest.mark.parametrize("seed", list(range(36)))
def test_random_storage_kinds_recover_a_linear_map(seed):
    # f(x) = A x with a random A inside a random pattern: the coloured finite-difference Jacobian must return A's stored
    # entries in every storage type the reference's plugins cover -- SparseMatrixCSC (common pattern), dense J with a sparse or
    # a dense-matrix pattern, Tridiagonal, BandedMatrix (rectangular, asymmetric), BlockBandedMatrix (ragged blocks) -- for
    # every fdtype, with a greedy colouring of the pattern (fd_color_columns_greedy).  Forward / central differences of a linear
    # map are exact up to rounding / eps; the complex step to a few ulp.
    import os
    rng = np.random.default_rng(int(os.environ.get("FDJAC_TEST_SEED_BASE", "5000")) + seed)
    kind = ["csc", "dense_sparse", "dense_dense", "tridiagonal", "banded", "blockbanded"][seed % 6]
    fdtype = FDTYPES[int(rng.integers(0, 3))]
    tol = {"forward": 2e-6, "central": 2e-9, "complex": 1e-13}[fdtype]
    lay = None
    if kind == "blockbanded":
        nb = int(rng.integers(3, 12))
        sizes = rng.integers(1, 7, size=nb)
        bl, bu = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        lay = P.BlockBandedLayout(sizes, bl, bu)
        M = N = lay.N
        mask = np.zeros((M, N), dtype=bool)
        off = np.concatenate([[0], np.cumsum(sizes)])
        for bi in range(nb):
            for bj in range(nb):
                if -bu <= bi - bj <= bl:
                    mask[off[bi]:off[bi + 1], off[bj]:off[bj + 1]] = True
    elif kind == "tridiagonal":
        M = N = int(rng.integers(4, 400))
        i, j = np.meshgrid(np.arange(M), np.arange(N), indexing="ij")
        mask = np.abs(i - j) <= 1
    elif kind == "banded":
        N = int(rng.integers(5, 300))
        M = max(2, N + int(rng.integers(-4, 5)))
        l, u = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        i, j = np.meshgrid(np.arange(M), np.arange(N), indexing="ij")
        mask = (i - j <= l) & (j - i <= u)
    else:
        M, N = int(rng.integers(3, 220)), int(rng.integers(3, 220))
        mask = rng.random((M, N)) < min(0.5, 4.0 / N)
        mask[rng.integers(0, M, size=N), np.arange(N)] = True          # no empty column
    A = np.where(mask, rng.random((M, N)) + 0.25, 0.0)
    x = rng.random(N)
    colptr, rowval = P.csc_from_dense(mask.astype(float))
    colors = fd.matrix_colors(fd.SparseMatrixCSC(M, N, colptr, rowval)) if kind != "blockbanded" else lay.colors()
    At = _dev(A)
    f = fd.TorchF(lambda fx, xx: fx.copy_(At.to(xx.dtype) @ xx), M, N)
    if kind == "csc":
        Jm = fd.SparseMatrixCSC(M, N, colptr, rowval, _dev(np.full(rowval.size, np.nan)))
        fd.finite_difference_jacobian_b(Jm, f, _dev(x), fdtype, colorvec=colors, sparsity=Jm)
        got = P.csc_to_dense(M, N, colptr, rowval, Jm.nzval.cpu().numpy())
    elif kind in ("dense_sparse", "dense_dense"):
        Jd = torch.full((N, M), float("nan"), dtype=torch.float64, device="cuda").t()       # column-major M x N
        sp = fd.SparseMatrixCSC(M, N, colptr, rowval) if kind == "dense_sparse" else mask.astype(float)
        fd.finite_difference_jacobian_b(Jd, f, _dev(x), fdtype, colorvec=colors, sparsity=sp)
        got = Jd.cpu().numpy()
    elif kind == "tridiagonal":
        Jt = fd.Tridiagonal(_dev(np.full(N - 1, np.nan)), _dev(np.full(N, np.nan)), _dev(np.full(N - 1, np.nan)))
        fd.finite_difference_jacobian_b(Jt, f, _dev(x), fdtype, colorvec=colors)
        got = np.diag(Jt.d.cpu().numpy()) + np.diag(Jt.dl.cpu().numpy(), -1) + np.diag(Jt.du.cpu().numpy(), 1)
    elif kind == "banded":
        data = torch.full((N, l + u + 1), float("nan"), dtype=torch.float64, device="cuda").t()
        fd.finite_difference_jacobian_b(fd.BandedMatrix(data, M, l, u), f, _dev(x), fdtype, colorvec=colors)
        got = P.banded_to_dense(data.cpu().numpy(), M, N, l, u)
    else:
        data = _dev(np.full(lay.data_len, np.nan))
        Jb = fd.BlockBandedMatrix(data, lay)
        fd.finite_difference_jacobian_b(Jb, f, _dev(x), fdtype, colorvec=colors, sparsity=Jb)
        got = lay.to_dense(data.cpu().numpy())
    assert np.all(np.isfinite(got)), (kind, fdtype, M, N)
    err = np.abs(got - A).max()
    assert err <= tol * max(1.0, float(np.abs(A @ x).max())), (kind, fdtype, M, N, err)


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_jvp_sizes_three_forms_agree(oracle, seed):
    # finite_difference_jvp! (src/jvp.jl:238-274) at random sizes (odd, tiny, around the small-problem threshold): the lazy
    # launcher that writes the finished quotient, the lazy launcher that writes values, and the materialised points give the same
    # bits; the result matches the oracle
    import os
    rng = np.random.default_rng(int(os.environ.get("FDJAC_TEST_SEED_BASE", "9000")) + seed)
    fdtype = ["forward", "central"][int(rng.integers(0, 2))]
    if seed % 3 == 2:
        nx, ny = 2 * int(rng.integers(2, 150)), int(rng.integers(2, 200))
        family, prm, N = ["lap5", "lap5_nl", "clamp5"][int(rng.integers(0, 3))], (nx, ny), nx * ny
    else:
        N = int([1, 2, 3, 5, 64, 16383, 16384, 16385][seed // 3]) if seed % 3 == 0 else int(rng.integers(4, 120_000))
        family, prm = ["tridiag", "tridiag_nl"][int(rng.integers(0, 2))], (N,)
    x = _dev(rng.random(N))
    v = _dev(rng.random(N) - 0.5)
    res = []
    for lazy, quotient in ((True, True), (True, False), (False, False)):
        f = fd.BuiltinF(family, *prm)
        out = _dev(np.full(N, np.nan))
        fd.finite_difference_jvp_b(out, f, x, v, fd.JVPCache(x, fdtype, lazy=lazy, quotient=quotient))
        assert f.fcalls == 2
        res.append(out)
    assert not torch.isnan(res[0]).any()
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2]), (family, prm, fdtype)
    if family != "clamp5":
        ref = oracle.jvp(fdtype, oracle.Fixture(family, *prm), x.cpu().numpy(), v.cpu().numpy())
        _tol_ok(res[0].cpu().numpy(), ref["jvp"], ref["eps"], 8.0, "jvp %s %s N=%d" % (family, fdtype, N))


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("case", ["plain", "linear_f", "shifted", "four_colours", "window", "window_odd", "window_odd_even", "chunked", "devplan",
                                  "none", "small_n", "tiny_n", "dir_minus", "chunked_window_odd", "chunked_four_colours", "chunked_shifted"])
def test_lazy_store_bit_identical(monkeypatch, fdtype, case):
    # include/fdjac_device.h, the default since round 3: a FD_LAZY_CAP_STORE launcher stores the finished quotients into nzval
    # itself (exact band verified at plan time, destination of (row, colour) by arithmetic) and the library launches no
    # decompression.  Same operations on the same operands as the hand-over path (FDJAC_LAZY_STORE=0): same bits, same f!
    # evaluation count.  Two kernels behind the capability: the column-centric wave kernel (fd_band_emit_wave: wave-private LDS
    # window, dense non-temporal stores) and, when the colours arrive in chunks ("chunked*": a scratch cap), round 2's
    # workgroup-owned rows.
    chunked = case.startswith("chunked")
    case = case[8:] if case.startswith("chunked_") else case
    N = {"small_n": 20_001, "tiny_n": 130}.get(case, 150_017)
    colptr, rowval = P.tridiag_csc(N)
    C = 4 if case == "four_colours" else 3
    colors = ((np.arange(N) + (2 if case == "shifted" else 0)) % C + 1).astype(np.int64)
    if case == "none":
        colors[[7, N // 2]] = 0
    win = {"window": (30_001, 120_000), "window_odd": (30_002, 119_999), "window_odd_even": (30_002, 120_000)}.get(case)
    cap = 8 * 2 * ((N + 31) // 32 * 32) * (2 if fdtype == "central" else 1) * 2 if chunked else 0
    x = _dev(np.random.default_rng(97).random(N))
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    outs, calls = [], []
    for store in ("1", "0"):
        monkeypatch.setenv("FDJAC_LAZY_STORE", store)
        monkeypatch.setenv("FDJAC_PLAN_DEVICE", "1" if case == "devplan" else "0")
        plan = fd.make_plan(J, J, colors, fdtype, scratch_bytes=cap, col_window=win,
                            x_window=(max(win[0] - 3, 1), min(win[1] + 3, N)) if win else None)
        f = fd.BuiltinF("tridiag" if case == "linear_f" else "tridiag_nl", N)
        assert f.lazy_caps & fd.lib.LAZY_CAP_STORE
        plan.set_lazy(f)
        assert plan.info(fd.lib.INFO_LAZY_STORE) == (1 if (store == "1" and case != "none") else 0)
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(f, x, [out], dir=-1.0 if case == "dir_minus" else True)
        outs.append(out)
        calls.append((f.fcalls, plan.fcalls_last))
    assert not torch.isnan(outs[0]).any()
    bad = torch.nonzero(outs[0].view(torch.int64) != outs[1].view(torch.int64)).flatten()
    assert bad.numel() == 0, (int(bad.numel()), bad[:8].tolist(), outs[0][bad[:8]].tolist(), outs[1][bad[:8]].tolist())
    assert calls[0] == calls[1]


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["csc", "banded", "tridiagonal"])
@pytest.mark.parametrize("case", ["plain", "shifted5", "window", "window_odd", "tiny"])
def test_lazy_store_storage_types_bit_identical(monkeypatch, oracle, fdtype, dtype, kind, case):
    # the same capability for every storage type whose band is arithmetic -- SparseMatrixCSC nzval, BandedMatrix data
    # (ext/FiniteDiffBandedMatricesExt.jl:13-27; corner slots hold 0) and Tridiagonal dl / d / du (src/iteration_utils.jl:25-32)
    # -- in both element types: bits of the hand-over path, and the oracle within tolerance
    N = 131 if case == "tiny" else 70_003
    C, shift = (5, 3) if case == "shifted5" else (3, 0)
    colors = ((np.arange(N) + shift) % C + 1).astype(np.int64)
    win = {"window": (10_001, 50_000), "window_odd": (10_002, 49_999)}.get(case)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    x = torch.as_tensor(np.random.default_rng(5).random(N), dtype=tdt, device="cuda")
    colptr, rowval = P.tridiag_csc(N)
    if kind == "csc":
        J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    elif kind == "banded":
        J = fd.BandedMatrix(torch.zeros((N, 3), dtype=tdt, device="cuda").t(), N, 1, 1)
    else:
        J = fd.Tridiagonal(torch.zeros(N - 1, dtype=tdt, device="cuda"), torch.zeros(N, dtype=tdt, device="cuda"),
                           torch.zeros(N - 1, dtype=tdt, device="cuda"))
    res = []
    for store in ("1", "0"):
        monkeypatch.setenv("FDJAC_LAZY_STORE", store)
        plan = fd.make_plan(J, J, colors, fdtype, col_window=win, dtype=dtype)
        f = fd.BuiltinF("tridiag_nl", N, dtype=dtype)
        plan.set_lazy(f)
        assert plan.info(fd.lib.INFO_LAZY_STORE) == int(store)
        outs = [torch.full((plan.out_len(k),), float("nan"), dtype=tdt, device="cuda") for k in range(plan.nouts)]
        plan.jacobian(f, x, outs)
        res.append((outs, f.fcalls, plan.fcalls_last))
    ity = torch.int32 if dtype == np.float32 else torch.int64
    for a, b in zip(res[0][0], res[1][0]):
        assert not torch.isnan(a).any()
        assert torch.equal(a.view(ity), b.view(ity)), (kind, case, int((a.view(ity) != b.view(ity)).sum()))
    assert res[0][1:] == res[1][1:]
    if kind == "csc" and win is None:
        ref = oracle.jacobian(fdtype, oracle.Fixture("tridiag_nl", N, dtype=dtype), x.cpu().numpy(), colors, kind=oracle.PAT_CSC_COMMON,
                              colptr=colptr, rowval=rowval)
        got = res[0][0][0].cpu().numpy().astype(np.float64)
        if dtype == np.float64:
            _tol_ok(got, ref["out"], float(np.min(np.abs(_oracle_eps(x.cpu().numpy(), colors, fdtype)))), 16.0, "store %s %s" % (kind, fdtype))
        else:
            assert np.max(np.abs(got - ref["out"]) / (2e-2 * np.abs(ref["out"]) + 5e-2)) <= 1.0


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("case", ["small", "small_nl", "device_plan", "device_plan_nl", "window", "wide_colours", "invalid_colouring", "clamp5",
                                  "ragged_width", "float32"])
def test_lazy_store_stencil5_bit_identical(monkeypatch, oracle, fdtype, case):
    # fd_stencil5_store (include/fdjac_device.h): for the exact 5-point stencil on an nx x ny grid with a VALID colouring the
    # Laplacian fixtures' launch stores the CSC Jacobian itself (k_f_stencil5_store_wave, column-centric) -- bits of the hand-over
    # path (FDJAC_LAZY_STORE=0), same f! evaluation count, oracle parity; an invalid colouring or the clamped fixture keep the
    # hand-over path
    nx, ny = {"device_plan": (640, 300), "device_plan_nl": (640, 300), "window": (256, 200), "ragged_width": (70, 50)}.get(case, (64, 40))
    N = nx * ny
    fam = "clamp5" if case == "clamp5" else ("lap5_nl" if case.endswith("_nl") else "lap5")
    dtype = np.float32 if case == "float32" else np.float64
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    colptr, rowval = P.lap5_csc(nx, ny)
    ii, jj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    colors = P.lap5_colors(nx, ny)
    if case == "wide_colours":
        colors = ((ii + 3 * jj) % 7 + 1).T.reshape(-1).astype(np.int64)          # 7 colours, still distance-2 valid
    if case == "invalid_colouring":
        colors = ((ii + jj) % 5 + 1).T.reshape(-1).astype(np.int64)              # (i+1, j) and (i, j+1) share a colour and the row (i+1, j+1)... and (i, j)
    win = (nx * 37 + 11, nx * 150 + 200) if case == "window" else None
    x = torch.as_tensor(np.random.default_rng(12).random(N), dtype=tdt, device="cuda")
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    res = []
    for store in ("1", "0"):
        monkeypatch.setenv("FDJAC_LAZY_STORE", store)
        plan = fd.make_plan(J, J, colors, fdtype, col_window=win, dtype=dtype)
        f = fd.BuiltinF(fam, nx, ny, dtype=dtype)
        plan.set_lazy(f)
        want_store = store == "1" and case not in ("invalid_colouring", "clamp5")     # (clamp5 has no FD_LAZY_CAP_STORE)
        assert plan.info(fd.lib.INFO_LAZY_STORE) == (1 if want_store else 0), case
        out = torch.full((plan.out_len(0),), float("nan"), dtype=tdt, device="cuda")
        plan.jacobian(f, x, [out])
        res.append((out, f.fcalls, plan.fcalls_last, plan.timings()))
    ity = torch.int32 if dtype == np.float32 else torch.int64
    assert not torch.isnan(res[0][0]).any()
    bad = torch.nonzero(res[0][0].view(ity) != res[1][0].view(ity)).flatten()
    assert bad.numel() == 0, (case, int(bad.numel()), bad[:8].tolist())
    assert res[0][1:3] == res[1][1:3]
    if case in ("small", "small_nl", "wide_colours") and dtype == np.float64:
        ref = oracle.jacobian(fdtype, oracle.Fixture(fam, nx, ny), x.cpu().numpy(), colors, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
        _tol_ok(res[0][0].cpu().numpy(), ref["out"], float(np.min(np.abs(_oracle_eps(x.cpu().numpy(), colors, fdtype)))), 8.0, "stencil store " + case)
        assert res[0][1] == ref["fcalls"]


@pytest.mark.parametrize("case", ["c5_shape", "bs8", "bs64", "bs2", "window", "window_mid_block", "chunked", "pairs_off", "invalid_colouring",
                                  "odd_block", "none", "cyclic", "float32", "float32_bs64"])
def test_lazy_store_blockbanded_complex_bit_identical(monkeypatch, oracle, case):
    # fd_colrange_store (include/fdjac_device.h), BASELINE config 5's shape: BlockBandedMatrix of equal dense blocks, complex step,
    # valid colouring -> the block-coupled fixture's launch forms sigma of every point on chip, evaluates every (row, column)
    # entry at its own point and stores imag / eps into the block-banded data itself (k_f_blockcoupled_store); bits of the
    # hand-over path (FDJAC_LAZY_STORE=0), same f! evaluation count, oracle parity.  Shapes the kernel does not take (odd
    # blocks, an invalid colouring, colorvec 0) keep the hand-over path.
    dtype = np.float32 if case.startswith("float32") else np.float64
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    nb, bs = {"c5_shape": (40, 32), "float32": (40, 32), "float32_bs64": (9, 64), "bs64": (9, 64), "bs2": (70, 2), "odd_block": (25, 5), "window": (40, 32), "window_mid_block": (40, 32),
              "chunked": (40, 32)}.get(case, (30, 8))
    sizes = np.full(nb, bs)
    N = int(sizes.sum())
    lay = P.BlockBandedLayout(sizes, 1, 1)
    colors = lay.colors()
    if case == "cyclic":
        colors = P.cyclic_colors(N, 3 * bs)
    if case == "invalid_colouring":
        colors = colors.copy()
        colors[bs] = colors[0]                    # first columns of blocks 0 and 1 share rows 0 .. 2 bs - 1
    if case == "none":
        colors = colors.copy()
        colors[[3, N - 2]] = 0
    win = {"window": (2 * bs + 1, N - bs), "window_mid_block": (2 * bs + 6, N - bs - 3)}.get(case)
    cap = 60_000 if case == "chunked" else 0
    x = torch.as_tensor(np.random.default_rng(77).random(N) - 0.3, dtype=tdt, device="cuda")
    Jb = fd.BlockBandedMatrix(None, lay)
    res = []
    for store in ("1", "0"):
        monkeypatch.setenv("FDJAC_LAZY_STORE", store)
        plan = fd.make_plan(Jb, Jb, colors, "complex", scratch_bytes=cap, col_window=win, dtype=dtype)
        f = fd.BuiltinF("blockcoupled", nb, int(sizes[0]), dtype=dtype)
        assert f.lazy_caps & fd.lib.LAZY_CAP_STORE
        plan.set_lazy(f, store=(case != "pairs_off"))
        want = store == "1" and case not in ("pairs_off", "invalid_colouring", "none")
        assert plan.info(fd.lib.INFO_LAZY_STORE) == (1 if want else 0), case
        if case == "chunked":
            assert plan.info(fd.lib.INFO_NCHUNKS) > 1
        out = torch.full((plan.out_len(0) + 2,), float("nan"), dtype=tdt, device="cuda")
        plan.jacobian(f, x, [out[:-2]])
        assert torch.isnan(out[-2:]).all()
        res.append((out[:-2], f.fcalls, plan.fcalls_last, plan.timings()))
    if win is None and case != "none":
        assert not torch.isnan(res[0][0]).any()
    ity = torch.int32 if dtype == np.float32 else torch.int64
    a, b = res[0][0].view(ity), res[1][0].view(ity)
    bad = torch.nonzero(a != b).flatten()
    assert bad.numel() == 0, (case, int(bad.numel()), bad[:8].tolist(), res[0][0][bad[:8]].tolist(), res[1][0][bad[:8]].tolist())
    assert res[0][1:3] == res[1][1:3]
    if case in ("c5_shape", "bs8", "bs2", "cyclic"):
        ref = oracle.jacobian("complex", oracle.Fixture("blockcoupled", nb, bs), x.cpu().numpy(), colors, kind=oracle.PAT_BLOCKBANDED,
                              blk_sizes=lay.blk_sizes, bl=1, bu=1, block_starts=lay.block_starts,
                              block_strides=lay.block_strides, out_len=lay.data_len)
        got = res[0][0].cpu().numpy()
        assert np.max(np.abs(got - ref["out"])) <= 1e-12 * max(1.0, np.max(np.abs(ref["out"])))
        assert res[0][1] == ref["fcalls"]


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("grid", [(1, 9), (9, 1), (2, 7), (7, 2), (3, 3), (4, 4), (5, 5), (130, 3), (3, 130), (128, 5), (129, 4), (256, 2), (2, 256),
                                  (127, 6), (131, 131)])
def test_lazy_store_stencil5_degenerate_grids(monkeypatch, oracle, fdtype, grid):
    # grids thinner than the stencil, one column / one row wide, a tile wide plus / minus one: whatever the plan decides (store the
    # Jacobian from f!'s launch, as a band, as a stencil, or hand over), the bits are those of the hand-over path and the oracle agrees
    nx, ny = grid
    N = nx * ny
    colptr, rowval = P.lap5_csc(nx, ny)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    colors = fd.matrix_colors(J)
    x = _dev(np.random.default_rng(nx * 1000 + ny).random(N))
    res = []
    for fam in ("lap5", "lap5_nl"):
        outs = []
        for store in ("1", "0"):
            monkeypatch.setenv("FDJAC_LAZY_STORE", store)
            plan = fd.make_plan(J, J, colors, fdtype)
            f = fd.BuiltinF(fam, nx, ny)
            if getattr(f, "lazy_fn", None) is not None:      # (the fixtures' lazy launchers need a grid of at least 2 x 2 tiles' worth)
                plan.set_lazy(f)
            out = _dev(np.full(plan.out_len(0) + 2, np.nan))
            plan.jacobian(f, x, [out[:-2]])
            assert torch.isnan(out[-2:]).all() and not torch.isnan(out[:-2]).any()
            outs.append((out[:-2], f.fcalls, int(plan.info(fd.lib.INFO_LAZY_STORE))))
        assert torch.equal(outs[0][0].view(torch.int64), outs[1][0].view(torch.int64)), (grid, fam, outs[0][2])
        assert outs[0][1] == outs[1][1]
        ref = oracle.jacobian(fdtype, oracle.Fixture(fam, nx, ny), x.cpu().numpy(), colors, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
        _tol_ok(outs[0][0].cpu().numpy(), ref["out"], float(np.min(np.abs(_oracle_eps(x.cpu().numpy(), colors, fdtype)))), 8.0, "degenerate grid %dx%d %s" % (nx, ny, fam))
        assert outs[0][1] == ref["fcalls"]


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("kind", ["csc", "banded", "tridiagonal"])
def test_lazy_store_tridiagonal_small_sizes(monkeypatch, oracle, fdtype, kind):
    # every small N around the wave / tile sizes of k_f_tridiag_store_wave (128 columns per wavefront, 512 per workgroup), each
    # storage type: bits of the hand-over path, the oracle within tolerance, nothing written past the end
    for N in (1, 2, 3, 4, 5, 7, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1025):
        colors = (np.arange(N) % 3 + 1).astype(np.int64)
        colptr, rowval = P.tridiag_csc(N)
        x = _dev(np.random.default_rng(N).random(N))
        res = []
        for store in ("1", "0"):
            monkeypatch.setenv("FDJAC_LAZY_STORE", store)
            if kind == "csc":
                J = fd.SparseMatrixCSC(N, N, colptr, rowval)
            elif kind == "banded":
                J = fd.BandedMatrix(torch.zeros((N, 3), dtype=torch.float64, device="cuda").t(), N, 1, 1)
            else:
                J = fd.Tridiagonal(torch.zeros(max(N - 1, 0), dtype=torch.float64, device="cuda"), torch.zeros(N, dtype=torch.float64, device="cuda"),
                                   torch.zeros(max(N - 1, 0), dtype=torch.float64, device="cuda"))
            plan = fd.make_plan(J, J, colors, fdtype)
            f = fd.BuiltinF("tridiag_nl", N)
            if getattr(f, "lazy_fn", None) is not None:
                plan.set_lazy(f)
            outs = [torch.full((plan.out_len(k) + 2,), float("nan"), dtype=torch.float64, device="cuda") for k in range(plan.nouts)]
            plan.jacobian(f, x, [o[:o.numel() - 2] for o in outs])
            for o in outs:
                assert torch.isnan(o[-2:]).all() and not torch.isnan(o[:-2]).any(), (N, kind)
            res.append(([o[:-2] for o in outs], f.fcalls))
        for a, b in zip(res[0][0], res[1][0]):
            assert torch.equal(a.view(torch.int64), b.view(torch.int64)), (N, kind)
        assert res[0][1] == res[1][1]
        if kind == "csc":
            ref = oracle.jacobian(fdtype, oracle.Fixture("tridiag_nl", N), x.cpu().numpy(), colors, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
            _tol_ok(res[0][0][0].cpu().numpy(), ref["out"], float(np.min(np.abs(_oracle_eps(x.cpu().numpy(), colors, fdtype)))), 16.0, "small N %d" % N)


@pytest.mark.parametrize("shape", [(1, 2), (1, 32), (2, 32), (3, 32), (4, 32), (5, 32), (9, 32), (2, 1), (7, 3), (3, 64), (13, 16), (4, 6), (6, 4)])
def test_lazy_store_blockbanded_small_shapes(monkeypatch, oracle, shape):
    # one to a few block-columns (the groups of k_f_blockcoupled_store hold 4, with 2 halo blocks on either side), block sizes
    # 1 .. 64: bits of the hand-over path, oracle parity
    nb, bs = shape
    N = nb * bs
    lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
    colors = lay.colors()
    x = _dev(np.random.default_rng(nb * 100 + bs).random(N) - 0.4)
    Jb = fd.BlockBandedMatrix(None, lay)
    res = []
    for store in ("1", "0"):
        monkeypatch.setenv("FDJAC_LAZY_STORE", store)
        plan = fd.make_plan(Jb, Jb, colors, "complex")
        f = fd.BuiltinF("blockcoupled", nb, bs)
        plan.set_lazy(f)
        out = _dev(np.full(plan.out_len(0) + 2, np.nan))
        plan.jacobian(f, x, [out[:-2]])
        assert torch.isnan(out[-2:]).all() and not torch.isnan(out[:-2]).any()
        res.append((out[:-2], f.fcalls))
    assert torch.equal(res[0][0].view(torch.int64), res[1][0].view(torch.int64)), shape
    assert res[0][1] == res[1][1]
    ref = oracle.jacobian("complex", oracle.Fixture("blockcoupled", nb, bs), x.cpu().numpy(), colors, kind=oracle.PAT_BLOCKBANDED,
                          blk_sizes=lay.blk_sizes, bl=1, bu=1, block_starts=lay.block_starts, block_strides=lay.block_strides,
                          out_len=lay.data_len)
    assert np.max(np.abs(res[0][0].cpu().numpy() - ref["out"])) <= 1e-12 * max(1.0, np.max(np.abs(ref["out"])))


@pytest.mark.parametrize("fdtype", ["forward", "central"])
def test_far_band_tile_order_is_pure_scheduling(monkeypatch, oracle, fdtype):
    # sorted-gather plans of patterns with a far band (3-D 7-point stencil: offsets +-1, +-nx, +-nx*ny) walk their tiles region by
    # region across the planes instead of in storage order (fdjac_api.hip, "Tile ORDER"): the same values in the same places
    n = 36
    N = n ** 3
    k = np.arange(N, dtype=np.int64)
    i, j, l = k % n, (k // n) % n, k // (n * n)
    has = np.stack([l > 0, j > 0, i > 0, np.ones(N, bool), i < n - 1, j < n - 1, l < n - 1], axis=1)
    rows = np.stack([k - n * n, k - n, k - 1, k, k + 1, k + n, k + n * n], axis=1)
    colptr = np.empty(N + 1, np.int64)
    colptr[0] = 1
    np.cumsum(has.sum(axis=1), out=colptr[1:])
    colptr[1:] += 1
    rowval = (rows[has] + 1).astype(np.int64)
    colors = ((i + 2 * j + 3 * l) % 7 + 1).astype(np.int64)
    x = _dev(np.random.default_rng(9).random(N))

    def f_t(fv, xx):        # a 3-D stencil residual with a different weight per neighbour (no coupling across grid lines / planes)
        X = xx.view(n, n, n)                        # [l, j, i]
        F = X * X
        F[:, :, 1:] += 1.0 * X[:, :, :-1]
        F[:, :, :-1] += 0.5 * X[:, :, 1:]
        F[:, 1:, :] += 0.25 * X[:, :-1, :]
        F[:, :-1, :] += 0.125 * X[:, 1:, :]
        F[1:] += 2.0 * X[:-1]
        F[:-1] += 3.0 * X[1:]
        fv.copy_(F.reshape(-1))

    outs = []
    for order in ("2", "0"):
        monkeypatch.setenv("FDJAC_TILE_ORDER", order)
        monkeypatch.setenv("FDJAC_SORTED", "1")
        J = fd.SparseMatrixCSC(N, N, colptr, rowval)
        plan = fd.make_plan(J, J, colors, fdtype)
        assert plan.info(fd.lib.INFO_SORTED_GATHER) == 1
        f = fd.TorchF(f_t, N, N)
        out = _dev(np.full(rowval.size, np.nan))
        plan.jacobian(f, x, [out])
        outs.append(out)
    assert not torch.isnan(outs[0]).any()
    assert torch.equal(outs[0].view(torch.int64), outs[1].view(torch.int64))
    # against the analytic Jacobian of f_t
    got = outs[0].cpu().numpy()
    col = P.csc_cols(colptr) - 1
    row = rowval - 1
    xh = x.cpu().numpy()
    want = np.select([row == col, row == col + 1, row == col - 1, row == col + n, row == col - n, row == col + n * n, row == col - n * n],
                     [2 * xh[col], 1.0, 0.5, 0.25, 0.125, 2.0, 3.0])
    assert np.max(np.abs(got - want)) < (2e-6 if fdtype == "forward" else 2e-8)


def test_far_band_tile_order_with_a_padding_tile_in_a_column_window(monkeypatch):
    # the index lists are padded to 4096 entries = TWO tiles of the sorted-gather kernel, which walks ceil(nnz_local / 2048) tiles:
    # when nnz_local mod 4096 is in (0, 2048] the padded lists hold one all-padding tile that must not take a real tile's place in
    # the far-band order (round-3 advisor finding: a column window of a 3-D stencil dropped 2048 stored values)
    n = 36
    N = n ** 3
    k = np.arange(N, dtype=np.int64)
    i, j, l = k % n, (k // n) % n, k // (n * n)
    has = np.stack([l > 0, j > 0, i > 0, np.ones(N, bool), i < n - 1, j < n - 1, l < n - 1], axis=1)
    rows = np.stack([k - n * n, k - n, k - 1, k, k + 1, k + n, k + n * n], axis=1)
    colptr = np.empty(N + 1, np.int64)
    colptr[0] = 1
    np.cumsum(has.sum(axis=1), out=colptr[1:])
    colptr[1:] += 1
    rowval = (rows[has] + 1).astype(np.int64)
    colors = ((i + 2 * j + 3 * l) % 7 + 1).astype(np.int64)
    x = _dev(np.random.default_rng(19).random(N))

    def f_t(fv, xx):
        X = xx.view(n, n, n)
        F = X * X
        F[:, :, 1:] += 1.0 * X[:, :, :-1]
        F[:, :, :-1] += 0.5 * X[:, :, 1:]
        F[:, 1:, :] += 0.25 * X[:, :-1, :]
        F[:, :-1, :] += 0.125 * X[:, 1:, :]
        F[1:] += 2.0 * X[:-1]
        F[:-1] += 3.0 * X[1:]
        fv.copy_(F.reshape(-1))

    a = 5000
    checked = 0
    for b in range(N - 9000, N - 2000, 97):          # windows whose last column is NOT the largest key of the order
        nloc = int(colptr[b] - colptr[a])
        if not (0 < nloc % 4096 <= 2048):
            continue
        outs = []
        for order in ("2", "0"):
            monkeypatch.setenv("FDJAC_TILE_ORDER", order)
            monkeypatch.setenv("FDJAC_SORTED", "1")
            J = fd.SparseMatrixCSC(N, N, colptr, rowval)
            plan = fd.make_plan(J, J, colors, "forward", col_window=(a, b))
            assert plan.info(fd.lib.INFO_SORTED_GATHER) == 1 and plan.out_len() == nloc
            out = _dev(np.full(nloc, np.nan))
            plan.jacobian(fd.TorchF(f_t, N, N), x, [out])
            outs.append(out)
        assert not torch.isnan(outs[0]).any(), (b, int(torch.isnan(outs[0]).sum()))
        assert torch.equal(outs[0].view(torch.int64), outs[1].view(torch.int64))
        checked += 1
        if checked == 3:
            break
    assert checked == 3


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_blockbanded_switch_combinations_bit_identical(monkeypatch, seed):
    # the block-banded routes -- materialised points, the lazy launcher (imaginary parts or (re, im) pairs), the storing launch,
    # one wave / one workgroup per column range in the decompression -- are different schedules of the same operations: random
    # shapes, block sizes, column windows, colour chunks, element types; same bits, same number of f! evaluations
    rng = np.random.default_rng(9100 + seed)
    fdtype = FDTYPES[int(rng.integers(0, 3))] if seed % 3 else "complex"
    bs = int(rng.choice([1, 2, 3, 5, 8, 16, 31, 32, 33, 48, 64]))
    nb = int(rng.integers(1, max(2, min(400, 20000 // bs))))
    N = nb * bs
    lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
    colors = lay.colors()
    if rng.random() < 0.3:
        perm = rng.permutation(int(colors.max())) + 1          # another valid colouring: the colours renamed
        colors = perm[colors - 1].astype(np.int64)
    if rng.random() < 0.15:
        colors = colors.copy()
        colors[rng.integers(0, N, size=2)] = 0
    win = None
    if rng.random() < 0.35 and N >= 12:
        a = int(rng.integers(0, N // 3))
        win = (a + 1, int(rng.integers(a + N // 3, N)))
    dtype = np.float32 if rng.random() < 0.3 else np.float64
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    cap = 0
    if rng.random() < 0.3:      # a few colours per chunk
        per = (N + 31) // 32 * 32 * (2 if fdtype != "forward" else 1) * (2 if fdtype == "complex" else 1) * dtype().itemsize * 2
        cap = per * int(rng.integers(2, 9)) + 8192
    x = torch.as_tensor(rng.random(N) - 0.25, dtype=tdt, device="cuda")
    Jb = fd.BlockBandedMatrix(None, lay)

    def run(env, lazy, imag_only, store):
        for k in ("FDJAC_LAZY_STORE",):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        plan = fd.make_plan(Jb, Jb, colors, fdtype, scratch_bytes=cap, col_window=win, dtype=dtype)
        f = fd.BuiltinF("blockcoupled", nb, bs, dtype=dtype)
        if lazy:
            plan.set_lazy(f, imag_only=imag_only, store=store)
        out = torch.full((plan.out_len(0) + 2,), float("nan"), dtype=tdt, device="cuda")
        plan.jacobian(f, x, [out[:-2]])
        assert torch.isnan(out[-2:]).all()
        return out[:-2], f.fcalls

    ref, calls_ref = run({"FDJAC_LAZY_STORE": "0"}, False, True, False)
    env = {}
    if rng.random() < 0.3:
        env["FDJAC_LAZY_STORE"] = "0"
    got, calls = run(env, True, bool(rng.random() < 0.6), bool(rng.random() < 0.8))
    ity = torch.int32 if dtype == np.float32 else torch.int64
    both_nan = torch.isnan(ref) & torch.isnan(got)            # (outside a column window nothing is written by either)
    bad = torch.nonzero((got.view(ity) != ref.view(ity)) & ~both_nan).flatten()
    assert bad.numel() == 0, (nb, bs, fdtype, np.dtype(dtype).name, win, cap, env, int(bad.numel()), bad[:8].tolist(), got[bad[:8]].tolist(),
                              ref[bad[:8]].tolist())
    assert calls == calls_ref


@pytest.mark.parametrize("case", ["stencil3d", "random_band", "lap5_forced", "stencil3d_none", "random_band_window", "random_band_chunked"])
def test_sorted_gather_fx_through_lds_bit_identical(monkeypatch, oracle, case):
    # forward differences on the colour-sorted gather kernel: f(x) of the rows a tile touches is staged in LDS (at most 8 runs of
    # rows per tile, found at plan time) instead of gathered per entry as the list kernel does; same bits, oracle parity
    rng = np.random.default_rng(31)
    if case.startswith("stencil3d"):
        n = 30
        N = n ** 3
        k = np.arange(N, dtype=np.int64)
        i, j, l = k % n, (k // n) % n, k // (n * n)
        has = np.stack([l > 0, j > 0, i > 0, np.ones(N, bool), i < n - 1, j < n - 1, l < n - 1], axis=1)
        rws = np.stack([k - n * n, k - n, k - 1, k, k + 1, k + n, k + n * n], axis=1)
        colptr = np.empty(N + 1, np.int64); colptr[0] = 1
        np.cumsum(has.sum(axis=1), out=colptr[1:]); colptr[1:] += 1
        rowval = (rws[has] + 1).astype(np.int64)
        colors = ((i + 2 * j + 3 * l) % 7 + 1).astype(np.int64)
    elif case.startswith("random_band"):
        N = 60_000
        offs = np.sort(rng.integers(-300, 301, size=(N, 6)), axis=1)
        rws = np.sort(np.clip(np.arange(N)[:, None] + offs, 0, N - 1), axis=1)
        keep = np.ones_like(rws, bool); keep[:, 1:] = rws[:, 1:] != rws[:, :-1]
        colptr = np.empty(N + 1, np.int64); colptr[0] = 1
        np.cumsum(keep.sum(axis=1), out=colptr[1:]); colptr[1:] += 1
        rowval = (rws[keep] + 1).astype(np.int64)
        colors = fd.matrix_colors(fd.SparseMatrixCSC(N, N, colptr, rowval))
    else:
        nx, ny = 150, 90
        N = nx * ny
        colptr, rowval = P.lap5_csc(nx, ny)
        colors = P.lap5_colors(nx, ny)
    if case.endswith("_none"):
        colors = colors.copy(); colors[rng.integers(0, N, 5)] = 0
    win = (N // 5, N - N // 7) if case.endswith("_window") else None
    C = int(colors.max())
    cap = 8 * 2 * ((N + 31) // 32 * 32) * max(2, C // 5) + 8192 if case.endswith("_chunked") else 0
    xh = rng.random(N)
    x = _dev(xh)
    col = P.csc_cols(colptr) - 1
    wts = (0.3 + ((rowval * 7 + col * 3) % 11) / 11.0)               # one weight per stored entry: f = A_w * (x + x^2 / 2)
    # (row-wise in a fixed order -- ELL layout -- so that f! is deterministic: atomics would not be)
    order = np.lexsort((col, rowval - 1))
    r_s, c_s, w_s = (rowval - 1)[order], col[order], wts[order]
    first = np.searchsorted(r_s, np.arange(N))
    slot = np.arange(r_s.size) - first[r_s]
    K = int(slot.max()) + 1
    ell_c, ell_w = np.zeros((K, N), np.int64), np.zeros((K, N))
    ell_c[slot, r_s], ell_w[slot, r_s] = c_s, w_s
    ell_c_t, ell_w_t = _dev(ell_c).long(), _dev(ell_w)

    def f_t(fv, xx):
        g = xx + 0.5 * xx * xx
        acc = ell_w_t[0] * g[ell_c_t[0]]
        for kk in range(1, K):
            acc = acc + ell_w_t[kk] * g[ell_c_t[kk]]
        fv.copy_(acc)

    outs = []
    for srt in ("1", "0"):            # the sorted gather (f(x) through LDS) against the list kernel (f(x) gathered per entry)
        monkeypatch.setenv("FDJAC_SORTED", srt)
        monkeypatch.setenv("FDJAC_WINDOW", "0")
        J = fd.SparseMatrixCSC(N, N, colptr, rowval)
        plan = fd.make_plan(J, J, colors, "forward", scratch_bytes=cap, col_window=win)
        assert plan.info(fd.lib.INFO_SORTED_GATHER) == int(srt)
        if cap:
            assert plan.info(fd.lib.INFO_NCHUNKS) > 1
        f = fd.TorchF(f_t, N, N)
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(f, x, [out])
        outs.append(out)
    assert torch.equal(outs[0].view(torch.int64), outs[1].view(torch.int64))
    got = outs[0].cpu().numpy()
    want = wts * (1.0 + xh[col])
    if win is not None:
        e0, e1 = colptr[win[0]] - 1, colptr[win[1]] - 1            # (column window [a, b), 0-based)
        want = want[e0:e1]
    keep_e = (colors[col] != 0)
    if win is not None:
        keep_e = keep_e[e0:e1]
    assert not np.isnan(got[keep_e]).any()
    assert np.max(np.abs(got[keep_e] - want[keep_e])) < 5e-6
