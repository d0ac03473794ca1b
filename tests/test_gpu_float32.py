"""The Float32 instantiation (fd32_*): x::Vector{Float32} problems of the reference, on the device vs the Float32
oracle (the same C restatement compiled with element type float).

Tolerance: |J_gpu - J_cpu| <= 1e-3*|J_cpu| + 16*eps(Float32)*scale/|eps_c|  (forward differences in Float32 carry
~3e-4 of relative truncation + rounding error themselves); complex step 1e-5; step sizes agree to 1e-5 relative
(the masked norm is accumulated in Float64 on the device, in Float32 by the reference).
"""
import numpy as np
import pytest

import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
EPS32 = float(np.finfo(np.float32).eps)
FDTYPES = ["forward", "central", "complex"]
F32 = np.float32


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=F32), device="cuda")


def _eps_ref(x, colors, fdtype):
    if fdtype == "complex":
        return np.full(int(colors.max()), EPS32)
    rel = fd.default_relstep(fdtype, F32)
    return np.array([max(rel * np.sqrt(np.linalg.norm((x * (colors == c)).astype(np.float64))), rel)
                     for c in range(1, int(colors.max()) + 1)])


def _tol_ok(g, c, eps_min, fscale, what, rtol=1e-3):
    g, c = np.asarray(g, np.float64), np.asarray(c, np.float64)
    atol = 16 * EPS32 * fscale / abs(eps_min)
    bad = np.abs(g - c) > rtol * np.abs(c) + atol
    assert not bad.any(), "%s: %d entries off, worst %.3e (atol %.1e)" % (what, int(bad.sum()), float(np.max(np.abs(g - c))), atol)


@pytest.mark.parametrize("fdtype,ncalls", [("forward", 4), ("central", 6), ("complex", 3)])
def test_reference_fixture_float32(oracle, fdtype, ncalls):
    # test/coloring_tests.jl:33-49 with x = rand(Float32, 30)
    N = 30
    x = np.random.default_rng(11).random(N).astype(F32)
    colors = np.tile([1, 2, 3], 10)
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.full(rowval.size, np.nan)))
    f = fd.BuiltinF("tridiag", N, dtype=F32)
    cache = fd.JacobianCache(_dev(x), fdtype, colorvec=colors, sparsity=J)
    assert cache.dtype == np.float32
    fd.finite_difference_jacobian_b(J, f, _dev(x), cache)
    assert f.fcalls == ncalls and J.nzval.dtype == torch.float32
    ref = oracle.jacobian(fdtype, oracle.Fixture("tridiag", N, dtype=F32), x, colors, kind=oracle.PAT_CSC_COMMON,
                          colptr=colptr, rowval=rowval)
    assert ref["out"].dtype == np.float32
    eps = cache.last_plan.epsilons()
    assert np.allclose(eps, _eps_ref(x, colors, fdtype), rtol=1e-5, atol=0)
    _tol_ok(J.nzval.cpu().numpy(), ref["out"], np.min(np.abs(eps)), 4.0, "f32 tridiag csc " + fdtype,
            rtol=1e-5 if fdtype == "complex" else 1e-3)
    dense = P.csc_to_dense(N, N, colptr, rowval, J.nzval.cpu().numpy().astype(np.float64))
    exact = np.diag(np.full(N, -2.0)) + np.diag(np.ones(N - 1), 1) + np.diag(np.ones(N - 1), -1)
    assert np.abs(dense - exact).max() <= {"forward": 2e-3, "central": 1e-4, "complex": 1e-6}[fdtype]


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("kind", ["csc_window", "csc_list", "lap5", "tridiagonal", "banded", "blockbanded", "dense_arm", "densej"])
def test_storage_types_float32(monkeypatch, oracle, fdtype, kind):
    monkeypatch.delenv("FDJAC_WINDOW", raising=False)
    N = 4099
    fam, prm = "tridiag_nl", (N,)
    colors = P.cyclic_colors(N, 3)
    okw = {}
    if kind in ("csc_window", "csc_list", "densej"):
        if kind == "densej":
            N, prm = 60, (60,)
            colors = P.cyclic_colors(N, 3)
        colptr, rowval = P.tridiag_csc(N)
        if kind == "csc_list":
            monkeypatch.setenv("FDJAC_WINDOW", "0")
        okw = dict(kind=oracle.PAT_CSC_COMMON if kind != "densej" else oracle.PAT_CSC_DENSEJ, colptr=colptr, rowval=rowval)
        if kind == "densej":
            J = torch.full((N, N), float("nan"), dtype=torch.float32, device="cuda").t()
            sp = fd.SparseMatrixCSC(N, N, colptr, rowval)
        else:
            J = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.full(rowval.size, np.nan)))
            sp = J
    elif kind == "lap5":
        nx, ny = 96, 70
        N = nx * ny
        colptr, rowval = P.lap5_csc(nx, ny)
        colors = P.lap5_colors(nx, ny)
        fam, prm = "lap5", (nx, ny)
        J = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.full(rowval.size, np.nan)))
        sp = J
        okw = dict(kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    elif kind == "tridiagonal":
        J = fd.Tridiagonal(_dev(np.full(N - 1, np.nan)), _dev(np.full(N, np.nan)), _dev(np.full(N - 1, np.nan)))
        sp = None
        cp, rv = P.tridiag_csc(N)
        okw = dict(kind=oracle.PAT_COO_TRIDIAG, rows_index=rv, cols_index=P.csc_cols(cp))
    elif kind == "banded":
        data = torch.full((N, 3), float("nan"), dtype=torch.float32, device="cuda").t()
        J = fd.BandedMatrix(data, N, 1, 1)
        sp = None
        okw = dict(kind=oracle.PAT_BANDED, l=1, u=1)
    elif kind == "blockbanded":
        nb, bs = 30, 8
        N = nb * bs
        lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
        colors = lay.colors()
        J = fd.BlockBandedMatrix(_dev(np.full(lay.data_len, np.nan)), lay)
        sp = J
        fam, prm = "blockcoupled", (nb, bs)
        okw = dict(kind=oracle.PAT_BLOCKBANDED, blk_sizes=lay.blk_sizes, bl=1, bu=1, block_starts=lay.block_starts,
                   block_strides=lay.block_strides, out_len=lay.data_len)
    else:   # dense uncoloured arm
        N, prm = 120, (120,)
        colors = np.arange(1, N + 1)
        J = torch.full((N, N), float("nan"), dtype=torch.float32, device="cuda").t()
        sp = None
        okw = dict(kind=oracle.PAT_NONE)
    x = (np.random.default_rng(21).random(N) + 0.1).astype(F32)
    f = fd.BuiltinF(fam, *prm, dtype=F32)
    if kind == "dense_arm":
        fd.finite_difference_jacobian_b(J, f, _dev(x), fdtype)
    else:
        fd.finite_difference_jacobian_b(J, f, _dev(x), fdtype, colorvec=colors, sparsity=sp if sp is not None else "default")
    ref = oracle.jacobian(fdtype, oracle.Fixture(fam, *prm, dtype=F32), x, colors, **okw)
    assert f.fcalls == ref["fcalls"]
    if kind == "tridiagonal":
        got = np.concatenate([J.dl.cpu().numpy(), J.d.cpu().numpy(), J.du.cpu().numpy()])
        want = np.concatenate([np.asarray(o).ravel() for o in ref["out"]])
    elif kind in ("banded",):
        got, want = J.data.cpu().numpy().ravel(order="F"), np.asarray(ref["out"]).ravel(order="F")
    elif kind == "blockbanded":
        got, want = J.data.cpu().numpy(), ref["out"]
    elif kind in ("dense_arm", "densej"):
        got, want = J.cpu().numpy().ravel(order="F"), np.asarray(ref["out"]).ravel(order="F")
    else:
        got, want = J.nzval.cpu().numpy(), ref["out"]
    assert got.dtype == np.float32
    fin = np.isfinite(want)
    if kind == "dense_arm":
        rel = fd.default_relstep(fdtype, F32)
        eps_min = EPS32 if fdtype == "complex" else rel * 0.1
    else:
        eps_min = float(np.min(np.abs(_eps_ref(x, colors, fdtype))))
    _tol_ok(got[fin], want[fin], eps_min, 8.0, "f32 %s %s" % (kind, fdtype), rtol=1e-5 if fdtype == "complex" else 1e-3)
    assert np.all(got[~fin] == 0)


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("pattern", ["tridiag", "lap5", "blockcoupled"])
def test_kernel_variants_bit_identical_float32(monkeypatch, fdtype, pattern):
    # lazy vs materialised points, window vs gather kernels: the same bits in Float32 too
    if pattern == "tridiag":
        N = 9001
        cp, rv = P.tridiag_csc(N)
        colors, fam, prm = P.cyclic_colors(N, 3), "tridiag_nl", (N,)
        J = fd.SparseMatrixCSC(N, N, cp, rv)
    elif pattern == "lap5":
        nx, ny = 150, 77
        N = nx * ny
        cp, rv = P.lap5_csc(nx, ny)
        colors, fam, prm = P.lap5_colors(nx, ny), "lap5", (nx, ny)
        J = fd.SparseMatrixCSC(N, N, cp, rv)
    else:
        nb, bs = 40, 32
        N = nb * bs
        lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
        colors, fam, prm = lay.colors(), "blockcoupled", (nb, bs)
        J = fd.BlockBandedMatrix(None, lay)
    x = _dev(np.random.default_rng(31).random(N) + 0.1)
    outs = []
    for variant in ("gather_materialised", "window_materialised", "window_lazy", "window_lazy_pairs"):
        monkeypatch.setenv("FDJAC_WINDOW", "0" if variant.startswith("gather") else "1")
        plan = fd.make_plan(J, J, colors, fdtype, dtype=F32)
        f = fd.BuiltinF(fam, *prm, dtype=F32)
        if "lazy" in variant:
            plan.set_lazy(f, imag_only=not variant.endswith("pairs"))
        out = torch.full((plan.out_len(0),), float("nan"), dtype=torch.float32, device="cuda")
        plan.jacobian(f, x, [out])
        outs.append(out.cpu().numpy())
    assert not np.isnan(outs[0]).any()
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)


def test_user_f_and_host_arrays_float32():
    # a user f! in torch on Float32 views; numpy float32 in/out through the host path; dense-matrix sparsity
    def fn(dx, x):
        dx.copy_(torch.stack([x[0] ** 2 + x[1] ** 2 - x[0], x[0] * x[1], x[0] * x[2], x[0]]))

    theta = np.array([-3.0, 2.0, 13.3], F32)
    J = np.full((4, 3), np.nan, F32)
    f = fd.TorchF(fn, 4, 3, dtype=F32)
    cache = fd.JacobianCache(theta.copy(), np.zeros(4, F32), np.zeros(4, F32), "forward",
                             sparsity=np.array([[1, 1, 0], [1, 1, 0], [1, 0, 1], [1, 0, 0]]))
    fd.finite_difference_jacobian_b(J, f, theta, cache)
    E = np.array([[-7.0, 4.0, 0], [2.0, -3.0, 0.0], [13.3, 0.0, -3.0], [1.0, 0.0, 0.0]])
    assert J.dtype == np.float32 and f.fcalls == 4
    assert np.linalg.norm(J - E) <= 5e-3 * np.linalg.norm(E)
    with pytest.raises(TypeError):      # mixing element types is an error, not a silent conversion
        fd.finite_difference_jacobian_b(np.zeros((4, 3)), f, theta, cache)
    with pytest.raises(TypeError):      # ... also a Float64 launcher on a Float32 problem
        fd.finite_difference_jacobian_b(J, fd.TorchF(fn, 4, 3), theta, cache)


@pytest.mark.parametrize("fdtype", ["forward", "central"])
def test_jvp_float32(oracle, fdtype):
    N = 20011
    rng = np.random.default_rng(5)
    x, v = rng.random(N).astype(F32), (rng.random(N) - 0.5).astype(F32)
    out = _dev(np.full(N, np.nan))
    cache = fd.JVPCache(_dev(x), fdtype)
    fd.finite_difference_jvp_b(out, fd.BuiltinF("tridiag_nl", N, dtype=F32), _dev(x), _dev(v), cache)
    ref = oracle.jvp(fdtype, oracle.Fixture("tridiag_nl", N, dtype=F32), x, v)
    assert abs(cache.last_epsilon - ref["eps"]) <= 1e-5 * abs(ref["eps"])
    _tol_ok(out.cpu().numpy(), ref["jvp"], ref["eps"], 5.0, "f32 jvp " + fdtype)


def test_headline_shape_float32_properties():
    # N = 10^6 tridiagonal in Float32: linear fixture => the stencil, call count, x untouched
    N = 10 ** 6
    x = _dev(np.random.default_rng(2).random(N))
    xc = x.clone()
    cp, rv = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, cp, rv, torch.full((rv.size,), float("nan"), dtype=torch.float32, device="cuda"))
    f = fd.BuiltinF("tridiag", N, dtype=F32)
    fd.finite_difference_jacobian_b(J, f, x, "forward", colorvec=P.cyclic_colors(N, 3))
    assert f.fcalls == 4 and torch.equal(x, xc)
    got = J.nzval.cpu().numpy()
    isdiag = rv == P.csc_cols(cp)
    assert np.max(np.abs(got[isdiag] + 2.0)) < 0.2 and np.max(np.abs(got[~isdiag] - 1.0)) < 0.2   # eps ~ 0.01 at this norm


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("layout", ["csc", "csc_window0", "csc_window1", "csc_window2", "csc_window3", "banded", "banded_window", "tridiagonal",
                                    "tridiagonal_unaligned"])
def test_tridiagonal_store_four_columns_per_lane_float32(monkeypatch, fdtype, layout):
    # Float32: the storing launch of the tridiagonal fixture takes four columns per lane and writes 16-byte quads
    # (k_f_tridiag_store_wave4 / fd_band_emit_wave4).  Every position of a wavefront's first value relative to a quad boundary
    # (column windows starting at 4k, 4k+1, 4k+2, 4k+3), the three storage layouts, an output that is not 16-byte aligned:
    # the same bits as the hand-over path (f! values -> decompression kernel).
    N = 70_003
    colors = P.cyclic_colors(N, 3)
    x = _dev(np.random.default_rng(12).random(N) + 0.1)
    win = None
    if layout.startswith("csc"):
        cp, rv = P.tridiag_csc(N)
        J = fd.SparseMatrixCSC(N, N, cp, rv)
        if "window" in layout:
            a = 1000 + int(layout[-1])
            win = (a + 1, N - 777)
    elif layout.startswith("banded"):
        J = fd.BandedMatrix(None, N, 1, 1)
        if "window" in layout:
            win = (1002 + 1, N - 5)
    else:
        J = fd.Tridiagonal(None, np.empty(N, np.float32), None)
    outs = {}
    for store in ("1", "0"):
        monkeypatch.setenv("FDJAC_LAZY_STORE", store)
        sp = J if not layout.startswith("tridiagonal") and not layout.startswith("banded") else None
        plan = fd.make_plan(J, sp, colors, fdtype, dtype=F32, col_window=win)
        f = fd.BuiltinF("tridiag_nl", N, dtype=F32)
        plan.set_lazy(f)
        assert plan.info(fd.lib.INFO_LAZY_STORE) == int(store)
        bufs = []
        for k in range(plan.nouts):
            n = plan.out_len(k)
            if layout == "tridiagonal_unaligned":
                bufs.append(torch.full((n + 1,), float("nan"), dtype=torch.float32, device="cuda")[1:])
            else:
                bufs.append(torch.full((n,), float("nan"), dtype=torch.float32, device="cuda"))
        plan.jacobian(f, x, bufs)
        outs[store] = [b.cpu().numpy() for b in bufs]
    for a, b in zip(outs["1"], outs["0"]):
        assert not np.isnan(b).any()
        assert np.array_equal(a, b), layout


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("fam", ["lap5", "lap5_nl"])
@pytest.mark.parametrize("shape", [(256, 9), (260, 8), (1028, 7), (1536, 6), (64, 12), (516, 5, "window"), (1028, 7, "special")])
def test_stencil5_store_four_columns_per_lane_float32(monkeypatch, fdtype, fam, shape):
    # Float32: the storing launch of the 5-point fixtures takes four columns per lane, writes 16-byte quads and divides through
    # Float64 (k_f_stencil5_store_wave4 / fd_stencil5_emit_wave4).  Tiles at the two ends of a grid row, partial last tiles, interior
    # tiles, the first / last grid rows, a column window that cuts grid rows, and operands that send whole columns to the true division
    # (huge, zero and overflowing differences): the bits of the two-columns-per-lane kernel and of the hand-over path.
    nx, ny = shape[0], shape[1]
    what = shape[2] if len(shape) > 2 else ""
    N = nx * ny
    cp, rv = P.lap5_csc(nx, ny)
    colors = P.lap5_colors(nx, ny)
    J = fd.SparseMatrixCSC(N, N, cp, rv)
    xh = (np.random.default_rng(5).random(N) + 0.1).astype(F32)
    if what == "special":
        xh[3 * nx + 300] = 1e30
        xh[3 * nx + 700] = 3e38
        xh[2 * nx + 900: 2 * nx + 910] = 0.0
        xh[4 * nx + 20] = 1e-30
    x = _dev(xh)
    win = (nx + 37 + 1, N - nx - 5) if what == "window" else None
    outs = {}
    for form in ("wave4", "wave2", "handover"):
        monkeypatch.setenv("FDJAC_LAZY_STORE", "0" if form == "handover" else "1")
        monkeypatch.setenv("FDJAC_S5_WAVE4", "0" if form == "wave2" else "1")
        plan = fd.make_plan(J, J, colors, fdtype, dtype=F32, col_window=win)
        f = fd.BuiltinF(fam, nx, ny, dtype=F32)
        plan.set_lazy(f)
        assert plan.info(fd.lib.INFO_LAZY_STORE) == (0 if form == "handover" else 1)
        out = torch.full((plan.out_len(0),), float("nan"), dtype=torch.float32, device="cuda")
        plan.jacobian(f, x, [out])
        outs[form] = out.cpu().numpy()
    if what != "special":
        assert not np.isnan(outs["handover"]).any()
    for form in ("wave4", "wave2"):
        assert np.array_equal(outs[form].view(np.uint32), outs["handover"].view(np.uint32)), form


@pytest.mark.parametrize("fdtype", ["forward", "central"])
def test_tridiagonal_store_float32_special_operands(monkeypatch, fdtype):
    # zero differences next to huge coordinates, overflow, tiny values in the four-columns-per-lane storing launch: the bits of the
    # hand-over path everywhere (the division through Float64 of the 5-point kernel measured no gain here -- 30.6 against 30.6 us,
    # the kernel is bound by its stores -- and is not used)
    N = 50_000
    colors = P.cyclic_colors(N, 3)
    xh = (np.random.default_rng(8).random(N) + 0.1).astype(F32)
    xh[1000] = 1e30; xh[2000] = 3e38; xh[3000:3010] = 0.0; xh[4000] = 1e-30; xh[5000] = -1e20; xh[6000:6004] = 1e-38
    x = _dev(xh)
    cp, rv = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, cp, rv)
    outs = {}
    for store in ("1", "0"):
        monkeypatch.setenv("FDJAC_LAZY_STORE", store)
        plan = fd.make_plan(J, J, colors, fdtype, dtype=F32)
        f = fd.BuiltinF("tridiag_nl", N, dtype=F32)
        plan.set_lazy(f)
        assert plan.info(fd.lib.INFO_LAZY_STORE) == int(store)
        out = torch.full((plan.out_len(0),), float("nan"), dtype=torch.float32, device="cuda")
        plan.jacobian(f, x, [out])
        outs[store] = out.cpu().numpy()
    assert np.array_equal(outs["1"].view(np.uint32), outs["0"].view(np.uint32))
