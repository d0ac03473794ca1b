"""Pins the CPU oracle (oracle/fd_oracle.c) against every known-answer the reference's own
tests hold for the coloured-Jacobian path.  CPU only.

Each test cites the reference test lines it restates (relative to /root/reference).
"""
import numpy as np
import pytest

import finitediff_jl_amd  # noqa: F401
from finitediff_jl_amd import patterns as P


def second_derivative_stencil(N):  # test/coloring_tests.jl:19-26
    A = np.zeros((N, N))
    for i in range(N):
        A[i, i] = -2
        if i > 0:
            A[i, i - 1] = 1
        if i < N - 1:
            A[i, i + 1] = 1
    return A


def isapprox(a, b, rtol=np.sqrt(np.finfo(float).eps), atol=0.0):  # Julia isapprox on Frobenius norm
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.linalg.norm(a - b) <= max(atol, rtol * max(np.linalg.norm(a), np.linalg.norm(b)))


RNG = np.random.default_rng(12345)


@pytest.mark.parametrize("fdtype,ncalls", [("forward", 4), ("central", 6), ("complex", 3)])
def test_tridiag_csc_common(oracle, fdtype, ncalls):
    # test/coloring_tests.jl:33-49
    N = 30
    colptr, rowval = P.tridiag_csc(N)
    f = oracle.Fixture("tridiag", N)
    r = oracle.jacobian(fdtype, f, RNG.random(N), np.tile([1, 2, 3], 10), kind=oracle.PAT_CSC_COMMON,
                        colptr=colptr, rowval=rowval)
    assert r["fcalls"] == ncalls
    assert isapprox(P.csc_to_dense(N, N, colptr, rowval, r["out"]), second_derivative_stencil(N))


@pytest.mark.parametrize("fdtype,ncalls", [("forward", 4), ("central", 6), ("complex", 3)])
def test_tridiag_densej_sparse_pattern(oracle, fdtype, ncalls):
    # test/coloring_tests.jl:51-70
    N = 30
    colptr, rowval = P.tridiag_csc(N)
    f = oracle.Fixture("tridiag", N)
    r = oracle.jacobian(fdtype, f, RNG.random(N), np.tile([1, 2, 3], 10), kind=oracle.PAT_CSC_DENSEJ,
                        colptr=colptr, rowval=rowval)
    assert r["fcalls"] == ncalls
    assert isapprox(r["out"], second_derivative_stencil(N))


@pytest.mark.parametrize("fdtype,ncalls", [("forward", 4), ("central", 6), ("complex", 3)])
def test_tridiagonal_type_via_findstructralnz(oracle, fdtype, ncalls):
    # test/coloring_tests.jl:72-88,94-96 : Tridiagonal J, generic COO loop (src/iteration_utils.jl:25-32)
    N = 30
    colptr, rowval = P.tridiag_csc(N)
    rows, cols = rowval, P.csc_cols(colptr)
    f = oracle.Fixture("tridiag", N)
    r = oracle.jacobian(fdtype, f, RNG.random(N), np.tile([1, 2, 3], 10), kind=oracle.PAT_COO_TRIDIAG,
                        rows_index=rows, cols_index=cols)
    dl, d, du = r["out"]
    assert r["fcalls"] == ncalls
    J = np.diag(d) + np.diag(dl, -1) + np.diag(du, 1)
    assert isapprox(J, second_derivative_stencil(N))


def test_banded(oracle):
    # test/coloring_tests.jl:90-92
    N = 30
    f = oracle.Fixture("tridiag", N)
    r = oracle.jacobian("forward", f, RNG.random(N), np.tile([1, 2, 3], 10), kind=oracle.PAT_BANDED, l=1, u=1)
    assert isapprox(P.banded_to_dense(r["out"], N, N, 1, 1), second_derivative_stencil(N))


def test_stencil_blockbanded_vs_sparse(oracle):
    # test/coloring_tests.jl:99-119 : 100x100 clamped stencil; block-banded and sparse J agree.
    nx = ny = 100
    N = nx * ny
    x = RNG.random(N)
    f = oracle.Fixture("clamp5", nx, ny)
    colptr, rowval = P.lap5_csc(nx, ny)
    # 9-colour scheme of the (1,1)/(1,1) banded-block-banded structure
    k = np.arange(N)
    colors9 = 3 * ((k // nx) % 3) + (k % nx) % 3 + 1
    rs = oracle.jacobian("forward", f, x, colors9, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    lay = P.BlockBandedLayout(np.full(ny, nx), 1, 1)
    rb = oracle.jacobian("forward", f, x, lay.colors(), kind=oracle.PAT_BLOCKBANDED, blk_sizes=lay.blk_sizes,
                         bl=1, bu=1, block_starts=lay.block_starts, block_strides=lay.block_strides,
                         out_len=lay.data_len)
    assert rb["fcalls"] == 301
    # Jbb ~ Jsparse (compared entry-wise without densifying 10^4 x 10^4)
    at = lay.index_of(rowval - 1, P.csc_cols(colptr) - 1)
    assert isapprox(rb["out"][at], rs["out"])
    rest = rb["out"].copy()
    rest[at] = 0
    assert np.max(np.abs(rest)) < 1e-6  # dense in-band blocks are zero off the stencil
    # exact answer of the clamped sum stencil: each neighbour contributes 1, clamped ones add to self
    i, j = k % nx, k // nx
    diag = 1.0 + (i == 0) + (i == nx - 1) + (j == 0) + (j == ny - 1)
    isdiag = (rowval == P.csc_cols(colptr))
    assert np.allclose(rs["out"][isdiag], diag, rtol=0, atol=5e-7)
    assert np.allclose(rs["out"][~isdiag], 1.0, rtol=0, atol=5e-7)


def test_stencil_bandedblockbanded_vs_sparse(oracle):
    # test/coloring_tests.jl:109-115 : Jbbb ~ Jsparse with the BandedBlockBandedMatrix store path
    # (ext/FiniteDiffBlockBandedMatricesExt.jl:16-42), (1,1) block bandwidths, (1,1) sub-block bandwidths, 9 colours
    nx = ny = 100
    N = nx * ny
    x = RNG.random(N)
    f = oracle.Fixture("clamp5", nx, ny)
    colptr, rowval = P.lap5_csc(nx, ny)
    lay = P.BandedBlockBandedLayout(np.full(ny, nx), 1, 1, 1, 1)
    colors9 = lay.colors()
    assert colors9.max() == 9
    rs = oracle.jacobian("forward", f, x, colors9, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    rb = oracle.jacobian("forward", f, x, colors9, kind=oracle.PAT_BANDEDBLOCKBANDED, blk_sizes=lay.blk_sizes, bl=1, bu=1,
                         lam=1, mu=1, block_starts=lay.block_starts, block_strides=lay.block_strides,
                         out_len=lay.data_len)
    assert rb["fcalls"] == 10
    rows, cols, dest = lay.entries()
    # the banded-block-banded pattern contains the 5-point stencil (plus the 4 diagonal neighbours, which are 0)
    dense_at = {}
    pos = dict(zip(zip(rows.tolist(), cols.tolist()), dest.tolist()))
    ccols = P.csc_cols(colptr)
    at = np.array([pos[(int(r), int(c))] for r, c in zip(rowval, ccols)])
    assert isapprox(rb["out"][at], rs["out"])
    rest = rb["out"].copy()
    rest[at] = 0
    assert np.max(np.abs(rest)) < 1e-6


@pytest.mark.parametrize("fdtype", ["forward", "central", "complex"])
def test_nonsquare(oracle, fdtype):
    # test/coloring_tests.jl:124-159
    n = 4
    x0 = np.concatenate([np.arange(1, n + 1) + 0.5, np.arange(1, n + 1) + 1.5])
    rows = np.concatenate([np.arange(1, n + 1), np.arange(1, n + 1)])
    cols = np.concatenate([np.arange(1, n + 1), np.arange(1, n + 1) + n])
    A = np.zeros((n, 2 * n))
    A[rows - 1, cols - 1] = 1
    colptr, rowval = P.csc_from_dense(A)
    colorvec = np.concatenate([np.full(n, 1), np.full(n, 2)])
    f = oracle.Fixture("nonsquare", n)
    # J_nonsquare1: dense uncoloured forward Jacobian
    J1 = oracle.jacobian("forward", f, x0, np.arange(1, 2 * n + 1), M=n)["out"]
    r = oracle.jacobian(fdtype, f, x0, colorvec, M=n, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    want = {"forward": 3, "central": 4, "complex": 2}[fdtype]
    assert r["fcalls"] == want
    J2 = P.csc_to_dense(n, 2 * n, colptr, rowval, r["out"])
    assert isapprox(J2, J1, rtol=1e-6)
    # analytic: dy_k/dx1_k = 2(x1-3)+x2 ; dy_k/dx2_k = x1 + 2(x2+4)
    x1, x2 = x0[:n], x0[n:]
    Jex = np.hstack([np.diag(2 * (x1 - 3) + x2), np.diag(x1 + 2 * (x2 + 4))])
    assert isapprox(J2, Jex, rtol=1e-6)


def test_findstructralnz_dense_order(oracle):
    # test/coloring_tests.jl:163-168 : column-major enumeration == findstructralnz(sparse(a))
    for a in ([[1, 1], [0, 1]], [[1, 1, 1]], [[1.0, 1.0], [1.0, 1.0], [1.0, 1.0]], [[True, True], [True, True]]):
        a = np.asarray(a, float)
        rows, cols = oracle.findstructralnz_dense(a)
        colptr, rowval = P.csc_from_dense(a)
        assert np.array_equal(rows, rowval)
        assert np.array_equal(cols, P.csc_cols(colptr))


DENSE_SPARSITY_CASES = [
    # (f(dx,x), theta, M, sparsity, expected)  test/coloring_tests.jl:171-219
    (lambda dx, x: dx.__setitem__(slice(None), [x[0] ** 2 + x[1] ** 2, x[0] + x[1]]), [5.0, 3.0], 2,
     [[1, 1], [1, 1]], [[10, 6], [1, 1]]),
    (lambda dx, x: dx.__setitem__(slice(None), [x[0] ** 2 + x[1] ** 2, x[0]]), [-3.0, 2.0], 2,
     [[1, 1], [1, 0]], [[-6, 4], [1, 0]]),
    (lambda dx, x: dx.__setitem__(slice(None), [x[0] ** 2 + x[1] ** 2 - x[0]]), [-3.0, 2.0], 1,
     [[1, 1]], [[-7, 4]]),
    (lambda dx, x: dx.__setitem__(slice(None), [x[0] ** 2 + x[1] ** 2 - x[0], x[0] * x[1], x[0] * x[2], x[0]]),
     [-3.0, 2.0, 13.3], 4, [[1, 1, 0], [1, 1, 0], [1, 0, 1], [1, 0, 0]],
     [[-7.0, 4.0, 0], [2.0, -3.0, 0.0], [13.3, 0.0, -3.0], [1.0, 0.0, 0.0]]),
    (lambda dx, x: dx.__setitem__(slice(None), [x[0] ** 2 + x[1] ** 2]), [5.0, 3.0], 1, [[1, 1]], [[10.0, 6.0]]),
]


@pytest.mark.parametrize("case", range(len(DENSE_SPARSITY_CASES)))
def test_dense_matrix_sparsity_known_answers(oracle, case):
    fn, theta, M, sp, expected = DENSE_SPARSITY_CASES[case]
    N = len(theta)
    rows, cols = oracle.findstructralnz_dense(np.asarray(sp, float))
    f = oracle.PyF(fn, M, N)
    r = oracle.jacobian("forward", f, theta, np.arange(1, N + 1), M=M, kind=oracle.PAT_COO_DENSEJ,
                        rows_index=rows, cols_index=cols)
    assert isapprox(r["out"], expected)


J_REF = np.array([[2.0, 0.0], [0.0, 3.0], [4.0, 0.0]])  # test/cache_reuse_tests.jl:9


def _foo(y, x):
    y[0], y[1], y[2] = 2 * x[0], 3 * x[1], 4 * x[0]


@pytest.mark.parametrize("fdtype", ["forward", "central", "complex"])
def test_poisoned_cache_dense(oracle, fdtype):
    # test/cache_reuse_tests.jl:64-71
    f = oracle.PyF(_foo, 3, 2)
    cache = {"x1": np.full(2, 1e10), "fx": np.full(3, 1e10), "fx1": np.full(3, 1e10)}
    r = oracle.jacobian(fdtype, f, [1.0, 2.0], [1, 2], M=3, cache=cache)
    assert isapprox(r["out"], J_REF, atol=1e-6)


def test_central_sparse_restores_x_bitwise(oracle):
    # test/cache_reuse_tests.jl:73-83
    f = oracle.PyF(_foo, 3, 2)
    colptr, rowval = P.csc_from_dense(J_REF)
    x = np.array([1.0, 2.0])
    x_orig = x.copy()
    cache = {"x1": np.full(2, 1e10), "fx": np.full(3, 1e10), "fx1": np.full(3, 1e10)}
    r = oracle.jacobian("central", f, x, [1, 2], M=3, kind=oracle.PAT_CSC_DENSEJ, colptr=colptr, rowval=rowval,
                        cache=cache, mutate_x=True)
    assert isapprox(r["out"], J_REF, atol=1e-6)
    assert np.array_equal(x, x_orig)


def _iipf(fvec, x):  # test/finitedifftests.jl:399-402
    fvec[0] = (x[0] + 3) * (x[1] ** 3 - 7) + 18
    fvec[1] = np.sin(x[1] * np.exp(x[0]) - 1)


def test_dense_arm_tolerances(oracle):
    # test/finitedifftests.jl:455-462 : forward < 1e-6, central < 1e-8, complex < 1e-14 (max abs)
    x = np.random.default_rng(7).random(2)
    e = np.exp(x[0])
    J_ref = np.array([[-7 + x[1] ** 3, 3 * (3 + x[0]) * x[1] ** 2],
                      [e * x[1] * np.cos(1 - e * x[1]), e * np.cos(1 - e * x[1])]])
    f = oracle.PyF(_iipf, 2, 2)
    for fdtype, tol in (("forward", 1e-6), ("central", 1e-8), ("complex", 1e-14)):
        J = oracle.jacobian(fdtype, f, x, [1, 2])["out"]
        assert np.max(np.abs(J - J_ref)) < tol, fdtype
    # dir = -1 one-sided (finitedifftests.jl:456) and f_in reuse (:459)
    J = oracle.jacobian("forward", f, x, [1, 2], dir=-1.0)["out"]
    assert np.max(np.abs(J - J_ref)) < 1e-6
    fin = np.zeros(2)
    _iipf(fin, x)
    r = oracle.jacobian("forward", f, x, [1, 2], f_in=fin)
    assert r["fcalls"] == 2 and np.max(np.abs(r["out"] - J_ref)) < 1e-6


def test_config1_dense_sin(oracle):
    # BASELINE config 1: out-of-place dense forward Jacobian of sin.(x), N=1000 (src/jacobians.jl:319-331)
    N = 1000
    x = np.random.default_rng(1).random(N)
    J = oracle.jacobian_oop_dense_forward(oracle.Fixture("sin", N), x)
    assert np.max(np.abs(J - np.diag(np.cos(x)))) < 1e-6


def test_epsilon_rule(oracle):
    # src/epsilons.jl:26-29,133-144 and src/jacobians.jl:559-561: eps = max(rel*sqrt(norm(x.*mask)), abs)*dir
    assert oracle.default_relstep("forward") == np.sqrt(np.finfo(float).eps)
    assert oracle.default_relstep("central") == np.cbrt(np.finfo(float).eps)
    N = 30
    x = np.random.default_rng(0).random(N)
    colors = np.tile([1, 2, 3], 10)
    rel = np.sqrt(np.finfo(float).eps)
    # recover epsilon from a linear f: fx1 - fx = eps * J[:,cols of colour]; use identity-like f
    f = oracle.PyF(lambda y, xx: y.__setitem__(slice(None), xx), N, N)
    colptr = np.arange(1, N + 2, dtype=np.int64)
    rowval = np.arange(1, N + 1, dtype=np.int64)
    r = oracle.jacobian("forward", f, x, colors, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    assert np.allclose(r["out"], 1.0, rtol=0, atol=1e-7)
    for c in (1, 2, 3):
        eps = max(rel * np.sqrt(np.linalg.norm(x * (colors == c))), rel)
        assert 1e-8 < eps < 1e-7


def test_jvp_reference_fixture(oracle):
    # test/finitedifftests.jl:399-424,440-448 : JVP of iipf, forward < 1e-6, central < 1e-8, dir=-1, f_in
    rng = np.random.default_rng(17)
    x, vdir = rng.random(2), rng.random(2)
    e = np.exp(x[0])
    J_ref = np.array([[-7 + x[1] ** 3, 3 * (3 + x[0]) * x[1] ** 2],
                      [e * x[1] * np.cos(1 - e * x[1]), e * np.cos(1 - e * x[1])]])
    jvp_ref = J_ref @ vdir
    f = oracle.PyF(_iipf, 2, 2)
    assert np.max(np.abs(oracle.jvp("forward", f, x, vdir)["jvp"] - jvp_ref)) < 1e-6
    assert np.max(np.abs(oracle.jvp("forward", f, x, vdir, dir=-1.0)["jvp"] - jvp_ref)) < 1e-6
    assert np.max(np.abs(oracle.jvp("central", f, x, vdir)["jvp"] - jvp_ref)) < 1e-8
    fin = np.zeros(2)
    _iipf(fin, x)
    assert np.max(np.abs(oracle.jvp("forward", f, x, vdir, f_in=fin)["jvp"] - jvp_ref)) < 1e-6
    with pytest.raises(ValueError):
        oracle.jvp("complex", f, x, vdir)


@pytest.mark.parametrize("fdtype", ["forward", "central", "complex"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_c_oracle_agrees_with_independent_numpy_restatement(oracle, fdtype, seed):
    # two restatements of src/jacobians.jl:504-653 written separately (C, pass-for-pass; numpy, literal) must agree
    from oracle import np_oracle
    rng = np.random.default_rng(100 + seed)
    M, N = (23, 31) if seed == 0 else (40, 40) if seed == 1 else (37, 19)
    A = (rng.random((M, N)) < 0.15)
    A[rng.integers(0, M, N), np.arange(N)] = True
    W = rng.random((M, N)) * A
    x = rng.random(N) + 0.1
    colptr, rowval = P.csc_from_dense(A.astype(float))
    # a valid colouring by brute force: columns sharing a row get different colours
    colors = np.zeros(N, np.int64)
    for j in range(N):
        used = {colors[k] for k in range(j) if np.any(A[:, k] & A[:, j])}
        c = 1
        while c in used:
            c += 1
        colors[j] = c

    def fn(fx, xx):
        fx[:] = W @ (xx * xx) + (0.5 * W) @ np.sin(xx)      # couples rows and columns only inside the pattern

    Jn, calls_n = np_oracle.jacobian(fn, x, colors, A, fdtype)
    rc = oracle.jacobian(fdtype, oracle.PyF(fn, M, N), x, colors, M=M, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    Jc = P.csc_to_dense(M, N, colptr, rowval, rc["out"])
    assert rc["fcalls"] == calls_n
    scale = np.abs(Jn).max()
    tol = {"forward": 1e-7, "central": 1e-9, "complex": 1e-14}[fdtype] * scale
    assert np.max(np.abs(Jc - Jn)) <= tol


@pytest.mark.parametrize("fdtype", ["forward", "central", "complex"])
@pytest.mark.parametrize("coloring", ["valid", "invalid_single_colour", "invalid_two_colours"])
def test_broadcast_accumulate_arm_equals_assignment(oracle, fdtype, coloring):
    # the arm the reference takes when x1 has no fast scalar indexing (a GPU array),
    # fast_jacobian_setindex! (src/jacobians.jl:574-581, 665-671): J[r,c] += (color[c] == color_i) * vfx[r] over ALL
    # listed entries, for every colour, into a J zeroed by fill_matrix!.  Every listed (r, c) is a distinct location and
    # only ITS column's colour adds a non-zero term (Bool * Float64 is a strong zero: false * NaN == 0.0), so the
    # result equals the assignment form (src/iteration_utils.jl:25-32) for ANY colouring, valid or not -- an invalid
    # colouring corrupts the differences themselves, identically in both arms.  This pins the ABI contract: the device
    # path implements assignment, which IS the accumulate arm's result.
    M, N = 41, 37
    rng = np.random.default_rng(123)
    A = (rng.random((M, N)) < 0.15).astype(np.float64)
    A[np.arange(N), np.arange(N)] = 1.0
    W = rng.random((M, N)) * A
    rows, cols = oracle.findstructralnz_dense(A)

    def fn(fx, x):
        fx[:] = W @ (x * x) + np.sin(x[:1]) * 0   # keeps complex inputs complex

    if coloring == "valid":
        colors = np.arange(1, N + 1, dtype=np.int64)
    elif coloring == "invalid_single_colour":
        colors = np.ones(N, dtype=np.int64)
    else:
        colors = (np.arange(N) % 2 + 1).astype(np.int64)
    x = rng.random(N) + 0.1
    f = oracle.PyF(fn, M, N)
    a = oracle.jacobian(fdtype, f, x, colors, M, kind=oracle.PAT_COO_DENSEJ, rows_index=rows, cols_index=cols)
    b = oracle.jacobian(fdtype, f, x, colors, M, kind=oracle.PAT_COO_DENSEJ_ACCUM, rows_index=rows, cols_index=cols)
    assert a["fcalls"] == b["fcalls"]
    assert np.array_equal(a["out"], b["out"])          # (== treats -0.0 and 0.0 alike: the only representable difference)
    if coloring == "valid":
        J = 2 * W * x[None, :]
        assert np.allclose(a["out"], J, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("fdtype,tol", [("forward", 1e-4), ("central", 1e-8)])
def test_complex_valued_x_oracle_meets_the_reference_known_answer(fdtype, tol):
    # test/finitedifftests.jl:480-513 ("f : C^N -> C^N"): iipf, the analytic J_ref and the tolerances of the reference's own test,
    # on the numpy restatement's complex-valued-x arm (oracle/np_oracle.py::jacobian_complex_x) -- dense arm, cache-less call,
    # with f_in, with relstep = sqrt(eps) -- plus a coloured sparse case against the analytic Jacobian
    from oracle import np_oracle as O
    rng = np.random.default_rng(480)
    x = rng.random(2) + 1j * rng.random(2)

    def iipf(fv, xx):
        fv[0] = (1j * xx[0] + 3) * (xx[1] ** 3 - 7) + 18
        fv[1] = np.sin(xx[1] * np.exp(xx[0]) - 1)

    J_ref = np.array([[1j * (-7 + x[1] ** 3), 3 * (3 + 1j * x[0]) * x[1] ** 2],
                      [np.exp(x[0]) * x[1] * np.cos(1 - np.exp(x[0]) * x[1]), np.exp(x[0]) * np.cos(1 - np.exp(x[0]) * x[1])]])
    y = np.zeros(2, complex)
    iipf(y, x)
    for kw in ({}, {"relstep": np.sqrt(np.finfo(float).eps)}, {"f_in": y}):
        if fdtype == "central" and kw:
            continue
        J, ncalls = O.jacobian_complex_x(iipf, x, np.arange(1, 3), None, fdtype, **kw)
        assert np.abs(J - J_ref).max() < tol
        assert ncalls == (4 if fdtype == "central" else (2 if "f_in" in kw else 3))
    # coloured sparse arm: tridiagonal, complex-analytic f_i = (i x_{i-1} - 2 x_i) + x_{i+1} + x_i^2 x_{i+1}
    N = 30
    xs = rng.random(N) + 1j * rng.random(N)
    pat = np.abs(np.subtract.outer(np.arange(N), np.arange(N))) <= 1

    def f_tri(fv, xx):
        xm = np.concatenate([[0], xx[:-1]])
        xp = np.concatenate([xx[1:], [0]])
        fv[:] = (1j * xm - 2 * xx) + xp + xx * xx * xp

    J, ncalls = O.jacobian_complex_x(f_tri, xs, (np.arange(N) % 3) + 1, pat, fdtype)
    xp = np.concatenate([xs[1:], [0]])
    want = np.diag(-2 + 2 * xs * xp) + np.diag(1 + xs[:-1] ** 2, 1) + np.diag(np.full(N - 1, 1j), -1)
    assert np.abs(J - want).max() < (2e-6 if fdtype == "forward" else 2e-9)
    assert ncalls == (4 if fdtype == "forward" else 6)
