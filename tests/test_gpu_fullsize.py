"""BASELINE.json's configurations at their FULL sizes.

Two kinds of check:

1. HIP path vs the CPU oracle, EVERY stored value (`test_full_size_vs_oracle`): configs 2, 3, 4 and 5 at BASELINE's
   exact sizes with the NONLINEAR fixtures (J depends on x, so a stale buffer or a wrong step size cannot pass), with
   SURVEY 8(c)'s tolerance  |J_gpu - J_cpu| <= 1e-6*|J_cpu| + 16*eps(Float64)*max|f|/|eps_c|  and the step sizes to
   1e-12 relative.  The oracle needs 0.2 s (N = 10^7 tridiagonal forward) to ~1.5 s (5-point central, block-banded
   complex step) per Jacobian on one host core.
2. Size-independent properties computed with plain torch on the device (independent of the oracle):

* the analytic Jacobian of the seeded nonlinear fixtures (stored entry by stored entry, in storage order);
* linear fixtures: J is the constant stencil whatever x is, and J*v == f(v) exactly up to rounding;
* J*v (from the stored values) == finite_difference_jvp! of the same f at the same x;
* column shards concatenate to the bits of the unsharded call; a repeated call returns the same bits;
* x is left untouched and the number of f! calls is the reference's.

Finite-difference tolerances (|f''| <= 2, |f| <= 8 for these fixtures, x in (0,1)):
forward  eps/2*|f''| + 4*ulp(f)/eps  -> 2e-6;  central  eps^2/6*|f'''| + 2*ulp(f)/eps -> 2e-8;  complex 1e-13.
"""
import numpy as np
import pytest

import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ATOL = {"forward": 2e-6, "central": 2e-8, "complex": 1e-13}
CALLS = {"forward": lambda C: C + 1, "central": lambda C: 2 * C, "complex": lambda C: C}


def _rand(N, seed):
    return torch.as_tensor(np.random.default_rng(seed).random(N), dtype=torch.float64, device="cuda")


def _nan(n):
    return torch.full((int(n),), float("nan"), dtype=torch.float64, device="cuda")


def _spmv(colptr, rowval, nzval, v):
    """J*v from CSC storage (1-based colptr / rowval as the reference holds them)."""
    cols = torch.as_tensor(P.csc_cols(colptr) - 1, device="cuda")
    rows = torch.as_tensor(np.asarray(rowval, np.int64) - 1, device="cuda")
    return torch.zeros(v.numel(), dtype=nzval.dtype, device="cuda").index_add_(0, rows, nzval * v[cols])


def _tridiag_nl_analytic(x):
    """Stored values (CSC order: per column j the rows j-1, j, j+1) of f_i = x[i-1] - 2x[i] + x[i+1] + x[i]^2 x[i+1]."""
    N = x.numel()
    xp = torch.cat([x[1:], x.new_zeros(1)])
    diag = -2.0 + 2.0 * x * xp                     # df_j/dx_j
    upper = 1.0 + x[:-1] ** 2                      # df_{j-1}/dx_j, j = 1..N-1
    lower = torch.ones(N - 1, dtype=x.dtype, device=x.device)   # df_{j+1}/dx_j
    return lower, diag, upper


def _csc_from_diagonals(lower, diag, upper):
    N = diag.numel()
    out = torch.empty(3 * N - 2, dtype=diag.dtype, device=diag.device)
    # column 0 holds rows (0, 1) at entries 0, 1; column j >= 1 rows (j-1, j, j+1) at 3j-1, 3j, 3j+1
    j = torch.arange(N, device=diag.device)
    out[torch.clamp(3 * j - 1, min=0) + (j > 0)] = diag
    out[3 * j[1:] - 1] = upper
    out[3 * j[:-1] + 1] = lower
    return out


@pytest.mark.parametrize("fdtype", ["forward", "central", "complex"])
@pytest.mark.parametrize("N", [10 ** 6, 10 ** 7])
def test_tridiagonal_full_size_against_analytic_jacobian(fdtype, N):
    # BASELINE configs 2 (N = 10^6) and 4 (N = 10^7): tridiagonal CSC, colorvec = repeat(1:3)
    x = _rand(N, 2 if N == 10 ** 6 else 4)
    xc = x.clone()
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, _nan(rowval.size))
    f = fd.BuiltinF("tridiag_nl", N)
    cache = fd.JacobianCache(x, fdtype, colorvec=P.cyclic_colors(N, 3), sparsity=J)
    fd.finite_difference_jacobian_b(J, f, x, cache)
    assert f.fcalls == CALLS[fdtype](3) and torch.equal(x, xc)
    want = _csc_from_diagonals(*_tridiag_nl_analytic(x))
    err = (J.nzval - want).abs().max().item()
    assert err <= ATOL[fdtype], "worst entry error %.3e" % err
    # idempotence: the same call again returns the same bits
    first = J.nzval.clone()
    J.nzval.fill_(float("nan"))
    fd.finite_difference_jacobian_b(J, f, x, cache)
    assert torch.equal(J.nzval, first)
    # J*v from the stored values == finite_difference_jvp! (central) of the same f
    if fdtype == "central":
        v = _rand(N, 9) - 0.5
        Jv = _spmv(colptr, rowval, J.nzval, v)
        out = _nan(N)
        fd.finite_difference_jvp_b(out, fd.BuiltinF("tridiag_nl", N), x, v, fd.JVPCache(x, "central"))
        assert (out - Jv).abs().max().item() <= 1e-6


def test_tridiagonal_full_size_column_shards_concatenate():
    # BASELINE config 4: the 8 column ranges of the multi-GPU line, run one after the other on one device, give the bits
    # of the unsharded call (x windows with the halo the pattern needs)
    from finitediff_jl_amd import sharded as S
    N, world = 10 ** 7, 8
    x = _rand(N, 4)
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    pat = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    f = fd.BuiltinF("tridiag", N)
    full = _nan(rowval.size)
    fd.make_plan(pat, pat, colors, "forward").jacobian(f, x, [full])
    cuts = S.partition_columns(colptr, world)
    ranges = S.entry_ranges(colptr, cuts)
    got = _nan(rowval.size)
    for r in range(world):
        c0, c1 = int(cuts[r]), int(cuts[r + 1])
        plan = fd.make_plan(pat, pat, colors, "forward", col_window=(c0, c1), x_window=S.x_window(cuts, r, N, 1, 1, 1))
        a, b = ranges[r]
        plan.jacobian(f, x, [got[a:b]])
    assert torch.equal(got, full)
    # linear fixture: the constant second-difference stencil (test/coloring_tests.jl:19-26) at every one of 3N-2 entries
    isdiag = torch.as_tensor(rowval == P.csc_cols(colptr), device="cuda")
    assert (full[isdiag] + 2.0).abs().max().item() < 5e-8 and (full[~isdiag] - 1.0).abs().max().item() < 5e-8


@pytest.mark.parametrize("fdtype", ["central", "forward"])
def test_laplacian_full_size(fdtype):
    # BASELINE config 3: N = 10^7 (4000 x 2500) 5-point Laplacian, 5 colours
    nx, ny = 4000, 2500
    N = nx * ny
    x = _rand(N, 3)
    colptr, rowval = P.lap5_csc(nx, ny)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, _nan(rowval.size))
    f = fd.BuiltinF("lap5", nx, ny)
    fd.finite_difference_jacobian_b(J, f, x, fdtype, colorvec=P.lap5_colors(nx, ny))
    assert f.fcalls == CALLS[fdtype](5)
    isdiag = torch.as_tensor(rowval == P.csc_cols(colptr), device="cuda")
    tol = 1e-9 if fdtype == "central" else 1e-7
    assert (J.nzval[isdiag] + 4.0).abs().max().item() < tol and (J.nzval[~isdiag] - 1.0).abs().max().item() < tol
    # linear and homogeneous: J*v == f(v)
    v = _rand(N, 8) - 0.5
    Jv = _spmv(colptr, rowval, J.nzval, v)
    V = v.view(ny, nx)                          # zero-Dirichlet 5-point Laplacian, nx the fast index
    fv = -4.0 * V
    fv[:, 1:] += V[:, :-1]
    fv[:, :-1] += V[:, 1:]
    fv[1:, :] += V[:-1, :]
    fv[:-1, :] += V[1:, :]
    assert (Jv - fv.reshape(-1)).abs().max().item() < 1e-6


def test_blockbanded_full_size_against_analytic_jacobian():
    # BASELINE config 5: 10^4 dense 32 x 32 blocks, block-tridiagonal, complex step, 96 colours.
    # f_b[k] = x_b[k]*(sig_{b-1} + sig_b + sig_{b+1}) + sin(x_b[k]), sig_b = sum_j w_j x_b[j], w_j = (j+1)/bs
    nb, bs = 10 ** 4, 32
    N = nb * bs
    x = _rand(N, 5)
    lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
    Jb = fd.BlockBandedMatrix(_nan(lay.data_len), lay)
    f = fd.BuiltinF("blockcoupled", nb, bs)
    fd.finite_difference_jacobian_b(Jb, f, x, "complex", colorvec=lay.colors())
    assert f.fcalls == 96
    w = torch.arange(1, bs + 1, dtype=torch.float64, device="cuda") / bs
    xb = x.view(nb, bs)
    sig = xb @ w
    z = sig.new_zeros(1)
    Ssum = torch.cat([z, sig[:-1]]) + sig + torch.cat([sig[1:], z])
    dself = Ssum[:, None] + torch.cos(xb)                                   # extra diagonal term of block (b, b)
    # interior block-columns J = 1..nb-2: panel of 3*bs rows (blocks J-1, J, J+1) x bs columns, column-major
    first = 2 * bs * bs
    inner = Jb.data[first:first + (nb - 2) * 3 * bs * bs].view(nb - 2, bs, 3 * bs)   # [J, column j, row r]
    xr = x.unfold(0, 3 * bs, bs)                                              # rows' x: blocks J-1..J+1
    want = xr[:, None, :] * w[None, :, None]
    jj = torch.arange(bs, device="cuda")
    want[:, jj, bs + jj] += dself[1:-1]
    scale = want.abs().max().item()
    assert (inner - want).abs().max().item() <= 1e-13 * max(1.0, scale)
    # the two edge block-columns (2*bs rows each)
    for Jc, rows_x, doff, sl in ((0, x[:2 * bs], 0, slice(0, first)),
                                 (nb - 1, x[N - 2 * bs:], bs, slice(lay.data_len - first, lay.data_len))):
        pan = Jb.data[sl].view(bs, 2 * bs)
        wt = rows_x[None, :] * w[:, None]
        wt[jj, doff + jj] += dself[Jc]
        assert (pan - wt).abs().max().item() <= 1e-13 * max(1.0, scale)


def test_float32_full_size_headline():
    # the Float32 instantiation on BASELINE config 4's shape: linear fixture, constant stencil
    N = 10 ** 7
    x = _rand(N, 4).float()
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, torch.full((rowval.size,), float("nan"), dtype=torch.float32, device="cuda"))
    f = fd.BuiltinF("tridiag", N, dtype=np.float32)
    fd.finite_difference_jacobian_b(J, f, x, "forward", colorvec=P.cyclic_colors(N, 3))
    assert f.fcalls == 4
    isdiag = torch.as_tensor(rowval == P.csc_cols(colptr), device="cuda")
    # eps ~ 3.5e-4*sqrt(|x_c|) ~ 1e-2; ulp(f)/eps ~ 2.4e-7/1e-2
    assert (J.nzval[isdiag] + 2.0).abs().max().item() < 2e-3 and (J.nzval[~isdiag] - 1.0).abs().max().item() < 2e-3


# ---- HIP path vs the CPU oracle at BASELINE's exact sizes ------------------------------------------------------------
EPS64 = float(np.finfo(np.float64).eps)


def _oracle_eps(x, colors, fdtype):
    """The masked-norm step sizes (src/jacobians.jl:559-561 / 600-602 / 624) restated in numpy."""
    C = int(colors.max())
    if fdtype == "complex":
        return np.full(C, EPS64)
    rel = fd.default_relstep(fdtype)
    ss = np.bincount(colors - 1, weights=x * x, minlength=C)      # sum of squares per colour
    return np.maximum(rel * np.sqrt(np.sqrt(ss)), rel)


_CASES = {}


def _fullsize_case(name):
    """Pattern / colours of one BASELINE configuration (built once per session: the host-side pattern generators need
    several seconds at N = 10^7)."""
    if name not in _CASES:
        _CASES[name] = _build_fullsize_case(name)
    return _CASES[name]


def _build_fullsize_case(name):
    if name in ("c2", "c4"):                      # BASELINE configs 2 / 4: tridiagonal CSC, forward
        N = 10 ** 6 if name == "c2" else 10 ** 7
        colptr, rowval = P.tridiag_csc(N)
        return dict(N=N, seed=2 if name == "c2" else 4, fdtype="forward", colors=P.cyclic_colors(N, 3), fam="tridiag_nl",
                    prm=(N,), kind="csc", colptr=colptr, rowval=rowval, fscale=5.0)
    if name == "c3":                              # config 3: 4000 x 2500 5-point stencil, central
        nx, ny = 4000, 2500
        colptr, rowval = P.lap5_csc(nx, ny)
        return dict(N=nx * ny, seed=3, fdtype="central", colors=P.lap5_colors(nx, ny), fam="lap5_nl", prm=(nx, ny),
                    kind="csc", colptr=colptr, rowval=rowval, fscale=9.0)
    nb, bs = 10 ** 4, 32                          # config 5: 10^4 dense 32x32 blocks, complex step
    lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
    return dict(N=nb * bs, seed=5, fdtype="complex", colors=lay.colors(), fam="blockcoupled", prm=(nb, bs), kind="bb",
                lay=lay, fscale=60.0)


@pytest.mark.parametrize("lazy", [True, False], ids=["lazy_f", "materialized_f"])
@pytest.mark.parametrize("name", ["c2", "c3", "c4", "c5"])
def test_full_size_vs_oracle(oracle, name, lazy):
    # test/coloring_tests.jl:33-49 (tridiagonal), :99-119 (5-point stencil, block-banded) at BASELINE.json's sizes
    cs = _fullsize_case(name)
    N, fdtype, colors = cs["N"], cs["fdtype"], cs["colors"]
    xh = np.random.default_rng(cs["seed"]).random(N)
    x = torch.as_tensor(xh, device="cuda")
    xc = x.clone()
    f = fd.BuiltinF(cs["fam"], *cs["prm"])
    if cs["kind"] == "csc":
        J = fd.SparseMatrixCSC(N, N, cs["colptr"], cs["rowval"], None)
        okw = dict(kind=oracle.PAT_CSC_COMMON, colptr=cs["colptr"], rowval=cs["rowval"])
    else:
        lay = cs["lay"]
        J = fd.BlockBandedMatrix(None, lay)
        okw = dict(kind=oracle.PAT_BLOCKBANDED, blk_sizes=lay.blk_sizes, bl=lay.bl, bu=lay.bu,
                   block_starts=lay.block_starts, block_strides=lay.block_strides, out_len=lay.data_len)
    plan = fd.make_plan(J, J, colors, fdtype)
    if lazy:
        plan.set_lazy(f)
    out = _nan(plan.out_len(0))
    plan.jacobian(f, x, [out])
    C = int(colors.max())
    assert f.fcalls == CALLS[fdtype](C) and torch.equal(x, xc) and plan.fcalls_last == CALLS[fdtype](C)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    ref = oracle.jacobian(fdtype, oracle.Fixture(cs["fam"], *cs["prm"]), xh, colors, **okw)
    assert ref["fcalls"] == f.fcalls
    eps = plan.epsilons()
    assert np.allclose(eps, _oracle_eps(xh, colors, fdtype), rtol=1e-12, atol=0)
    want = ref["out"]
    atol = 16 * EPS64 * cs["fscale"] / float(np.min(np.abs(eps)))
    if fdtype == "complex":
        atol = 1e-12 * float(np.max(np.abs(want)))
    dev = np.abs(got - want)
    bad = dev > 1e-6 * np.abs(want) + atol
    assert not bad.any(), "%s: %d of %d stored values off, worst %.3e (atol %.1e)" % (name, int(bad.sum()), got.size,
                                                                                       float(dev.max()), atol)


@pytest.mark.parametrize("kernel", ["auto", "sorted"])
@pytest.mark.parametrize("fdtype", ["forward", "central"])
def test_large_rectangular_random_pattern_vs_oracle(monkeypatch, oracle, fdtype, kernel):
    # a general pattern at N = 10^6 (1.2e6 x 1.0e6, ~6 entries per column around a slanted "diagonal" plus a few far ones, greedy
    # colouring): the colour-sorted gather path with its plan-time extras (f(x) runs staged in LDS for forward differences, the
    # far-band tile order where it applies, host plan loops on threads) against the CPU oracle, every stored value
    if kernel == "sorted":
        monkeypatch.setenv("FDJAC_SORTED", "1")
    M, N = 1_200_000, 1_000_000
    rng = np.random.default_rng(1234)
    centre = (np.arange(N) * (M / N)).astype(np.int64)
    rws = centre[:, None] + rng.integers(-3000, 3001, size=(N, 6))
    far = rng.random(N) < 0.03
    rws[far, 0] = rng.integers(0, M, size=int(far.sum()))
    rws = np.abs(rws)                                              # (reflected at the matrix edges: clipping would pile thousands
    rws = np.sort(np.where(rws > M - 1, 2 * (M - 1) - rws, rws), axis=1)   #  of entries onto the first and last row)
    keep = np.ones_like(rws, bool)
    keep[:, 1:] = rws[:, 1:] != rws[:, :-1]
    colptr = np.empty(N + 1, np.int64)
    colptr[0] = 1
    np.cumsum(keep.sum(axis=1), out=colptr[1:])
    colptr[1:] += 1
    rowval = (rws[keep] + 1).astype(np.int64)
    col = np.repeat(np.arange(N), keep.sum(axis=1))
    J = fd.SparseMatrixCSC(M, N, colptr, rowval)
    colors = fd.matrix_colors(J)
    C = int(colors.max())
    assert C < 64
    wts = 0.3 + ((rowval * 7 + col * 3) % 11) / 11.0
    # f = A_w (x + x^2 / 2), row-wise in a fixed order (ELL) so that it is deterministic on the device and the same sum on the host
    order = np.lexsort((col, rowval - 1))
    r_s, c_s, w_s = (rowval - 1)[order], col[order], wts[order]
    first = np.searchsorted(r_s, np.arange(M))
    slot = np.arange(r_s.size) - first[r_s]
    K = int(slot.max()) + 1
    ell_c, ell_w = np.zeros((K, M), np.int64), np.zeros((K, M))
    ell_c[slot, r_s], ell_w[slot, r_s] = c_s, w_s
    ell_c_t, ell_w_t = torch.as_tensor(ell_c, device="cuda"), torch.as_tensor(ell_w, device="cuda")

    def f_t(fv, xx):
        g = xx + 0.5 * xx * xx
        acc = ell_w_t[0] * g[ell_c_t[0]]
        for kk in range(1, K):
            acc = acc + ell_w_t[kk] * g[ell_c_t[kk]]
        fv.copy_(acc)

    def f_n(fv, xx):
        g = xx + 0.5 * xx * xx
        acc = ell_w[0] * g[ell_c[0]]
        for kk in range(1, K):
            acc = acc + ell_w[kk] * g[ell_c[kk]]
        fv[:] = acc

    xh = rng.random(N)
    x = torch.as_tensor(xh, device="cuda")
    plan = fd.make_plan(J, J, colors, fdtype)
    kern = ("window2d" if plan.info(fd.lib.INFO_WINDOW2D) else "window" if plan.info(fd.lib.INFO_WINDOW)
            else "sorted" if plan.info(fd.lib.INFO_SORTED_GATHER) else "list")
    print("decompression kernel:", kern, "colours:", C)
    assert kernel == "auto" or kern == "sorted"
    f = fd.TorchF(f_t, M, N)
    out = _nan(plan.out_len(0))
    plan.jacobian(f, x, [out])
    assert f.fcalls == CALLS[fdtype](C)
    got = out.cpu().numpy()
    assert not np.isnan(got).any()
    want = wts * (1.0 + xh[col])                                   # the analytic Jacobian
    assert np.max(np.abs(got - want)) < (2e-5 if fdtype == "forward" else 2e-8)
    ref = oracle.jacobian(fdtype, oracle.PyF(f_n, M, N), xh, colors, M=M, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    assert ref["fcalls"] == f.fcalls
    eps = plan.epsilons()
    assert np.allclose(eps, _oracle_eps(xh, colors, fdtype), rtol=1e-12, atol=0)
    atol = 16 * EPS64 * 8.0 / float(np.min(np.abs(eps)))
    dev = np.abs(got - ref["out"])
    bad = dev > 1e-6 * np.abs(ref["out"]) + atol
    assert not bad.any(), "%d of %d stored values off, worst %.3e (atol %.1e)" % (int(bad.sum()), got.size, float(dev.max()), atol)
