"""The consumer of the Jacobian path: fd_tridiag_solve_* against scipy.linalg.solve_banded.

(alpha*I + beta*J) y = b with J in the storage the Jacobian plans fill (Tridiagonal diagonals / tridiagonal CSC nzval),
whole and sharded by column ranges (the shards run one after the other on the one GPU; the packet exchange of the
multi-GPU solve is replaced by writing the packets into one tensor -- the single-rank communicator test covers the RCCL
call itself), and end to end: coloured Jacobian on the device -> solve, nothing copied in between
(test/downstream/ordinarydiffeq_tridiagonal_solve.jl:18-30: a Rosenbrock step with a Tridiagonal jac_prototype)."""
import numpy as np
import pytest
from scipy.linalg import solve_banded

import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P
from finitediff_jl_amd import sharded as S

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _dev(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


def _system(N, seed, dominance=0.2):
    rng = np.random.default_rng(seed)
    dl, du = rng.random(max(N - 1, 0)) - 0.5, rng.random(max(N - 1, 0)) - 0.5
    d = rng.random(N) - 0.5
    # alpha*I + beta*J strictly diagonally dominant: |alpha + beta d_i| >= |beta|(|dl| + |du|) + dominance
    beta = -0.7
    alpha = 0.7 * (0.5 + 1.0) + dominance + 0.5
    b = rng.random(N) - 0.5
    return dl, d, du, b, alpha, beta


def _reference(dl, d, du, b, alpha, beta):
    N = d.size
    ab = np.zeros((3, N))
    ab[0, 1:] = beta * du
    ab[1, :] = alpha + beta * d
    ab[2, :-1] = beta * dl
    return solve_banded((1, 1), ab, b)


def _csc_nzval(dl, d, du):
    """nzval of sparse(Tridiagonal(dl, d, du)): column j holds (du[j-1], d[j], dl[j])."""
    N = d.size
    out = np.empty(3 * N - 2 if N > 1 else 1)
    p = 0
    for j in range(N):
        if j > 0:
            out[p] = du[j - 1]; p += 1
        out[p] = d[j]; p += 1
        if j + 1 < N:
            out[p] = dl[j]; p += 1
    return out


def _csc_nzval_fast(dl, d, du):
    N = d.size
    if N < 50:
        return _csc_nzval(dl, d, du)
    out = np.empty(3 * N - 2)
    j = np.arange(N)
    out[np.where(j > 0, 3 * j, 0)] = d
    out[3 * j[1:] - 1] = du
    out[3 * j[:-1] + 1] = dl
    return out


@pytest.mark.parametrize("layout", ["diagonals", "csc"])
@pytest.mark.parametrize("N", [1, 2, 3, 7, 8, 9, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 2113, 4095, 4096, 4097, 16385, 32769,
                               100003, 262143, 262144, 262145, 10 ** 6, 3 * 10 ** 6 + 1])
def test_tridiagonal_solve_matches_scipy(layout, N):
    # sizes around every tile / level boundary (tiles of 2048 rows, a factor of 8 per level, <= 512 rows at the top)
    dl, d, du, b, alpha, beta = _system(N, 100 + N)
    want = _reference(dl, d, du, b, alpha, beta)
    y = torch.full((N,), float("nan"), dtype=torch.float64, device="cuda")
    solver = fd.TridiagSolver(N, layout)
    if layout == "diagonals":
        J = fd.Tridiagonal(_dev(dl), _dev(d), _dev(du))
    else:
        J = [_dev(_csc_nzval_fast(dl, d, du))]
    bd = _dev(b)
    solver.solve(J, bd, y, alpha, beta)
    got = y.cpu().numpy()
    assert not np.isnan(got).any()
    assert np.max(np.abs(got - want)) <= 1e-11 * max(1.0, np.max(np.abs(want)))
    # the inputs are untouched and a second solve gives the same bits
    assert torch.equal(bd, _dev(b))
    y2 = torch.full_like(y, float("nan"))
    solver.solve(J, bd, y2, alpha, beta)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("layout", ["diagonals", "csc"])
@pytest.mark.parametrize("N", [4097, 4104, 4105, 16384, 16385, 16392, 16393, 32768 + 9, 131072, 131073, 262144, 262145, 262152, 262153, 262209,
                               2 ** 21 + 65, 10 ** 6 + 3, 16777216 + 520])
@pytest.mark.parametrize("start", ["0", "2"])
def test_two_levels_per_launch_against_one_level_per_launch(layout, N, start, monkeypatch):
    # the default schedule (k_tri_reduce2 / k_tri_top2 / k_tri_backsub2: two levels per launch, the middle level only in LDS) against
    # the one-level kernels: another arithmetic (one reciprocal per row, identity-padded chunks, back-substitution from the swept rows)
    # -- rounding-level differences only; sizes around tile (2048 rows), halo (8 rows), 64:1 and top (4096) boundaries
    dl, d, du, b, alpha, beta = _system(N, 300 + N % 1000)
    J = fd.Tridiagonal(_dev(dl), _dev(d), _dev(du)) if layout == "diagonals" else [_dev(_csc_nzval_fast(dl, d, du))]
    bd = _dev(b)
    solver = fd.TridiagSolver(N, layout)
    monkeypatch.setenv("FDJAC_SOLVE_TWO_LEVEL", start)      # two levels per launch from level 0 (everything) / 2 (the default)
    y2 = torch.full((N,), float("nan"), dtype=torch.float64, device="cuda")
    solver.solve(J, bd, y2, alpha, beta)
    monkeypatch.setenv("FDJAC_SOLVE_TWO_LEVEL", "-1")
    y1 = torch.full((N,), float("nan"), dtype=torch.float64, device="cuda")
    solver.solve(J, bd, y1, alpha, beta)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(y1).any()) and not bool(torch.isnan(y2).any())
    assert float((y1 - y2).abs().max()) <= 1e-13 * max(1.0, float(y1.abs().max()))
    y3 = torch.full_like(y2, float("nan"))
    monkeypatch.setenv("FDJAC_SOLVE_TWO_LEVEL", start)
    solver.solve(J, bd, y3, alpha, beta)
    assert torch.equal(y3, y2)                            # run to run: the same bits
    if N <= 3 * 10 ** 6:
        want = _reference(dl, d, du, b, alpha, beta)
        assert np.max(np.abs(y2.cpu().numpy() - want)) <= 1e-11 * max(1.0, np.max(np.abs(want)))


@pytest.mark.parametrize("layout", ["diagonals", "csc"])
@pytest.mark.parametrize("N,W", [(40, 2), (1000, 3), (5003, 8), (300007, 4), (10 ** 6 + 1, 8), (17, 8)])
def test_sharded_solve_equals_global_solve(layout, N, W):
    # each "rank" owns a column range and holds only its slice of J (as the column-window Jacobian plans leave it),
    # its rows of b and of y; phase A per rank -> packets -> phases B + C per rank
    dl, d, du, b, alpha, beta = _system(N, 7 * N + W)
    want = _reference(dl, d, du, b, alpha, beta)
    colptr, _rv = P.tridiag_csc(N)
    cuts = S.partition_columns(colptr, W)
    if np.any(np.diff(cuts) < 1):
        cuts = np.linspace(0, N, W + 1).astype(np.int64)
    nz = _csc_nzval_fast(dl, d, du)
    packets = torch.full((W, 8), float("nan"), dtype=torch.float64, device="cuda")
    ranks = []
    for r in range(W):
        c0, c1 = int(cuts[r]), int(cuts[r + 1])
        if layout == "diagonals":      # window-relative dl (A[j+1,j]), d, du (A[j-1,j], from column max(c0-1,0)+1 on) of columns [c0,c1)
            J = fd.Tridiagonal(_dev(dl[c0:min(c1, N - 1)]), _dev(d[c0:c1]), _dev(du[max(c0 - 1, 0):c1 - 1]))
        else:
            e0 = 3 * c0 - 1 if c0 > 0 else 0
            e1 = 3 * c1 - 1 if c1 < N else 3 * N - 2
            J = [_dev(nz[e0:e1])]
        solver = fd.TridiagSolver(N, layout, rows=(c0, c1))
        bl = _dev(b[c0:c1])
        solver.interface(J, bl, packets[r], alpha, beta)
        ranks.append((solver, J, bl, c0, c1))
    got = np.full(N, np.nan)
    for r, (solver, J, bl, c0, c1) in enumerate(ranks):
        yl = torch.full((c1 - c0,), float("nan"), dtype=torch.float64, device="cuda")
        solver.finish(J, bl, packets, r, W, yl, alpha, beta)
        got[c0:c1] = yl.cpu().numpy()
    assert not np.isnan(got).any()
    assert np.max(np.abs(got - want)) <= 1e-11 * max(1.0, np.max(np.abs(want)))


@pytest.mark.parametrize("seed", range(16))
def test_solver_randomised(seed):
    # random sizes (around every tile / level boundary by chance), layouts, shifts and rank counts with UNEVEN cuts: whole and sharded
    # solves against SciPy's banded solve; the sharded pieces go through the two-levels-per-launch schedule whenever a rank's rows allow
    rng = np.random.default_rng(5000 + seed)
    N = int(rng.choice([rng.integers(1, 70), rng.integers(500, 5000), rng.integers(4000, 70000), rng.integers(250000, 600000)]))
    layout = "csc" if rng.random() < 0.5 else "diagonals"
    dl, d, du, b, alpha, beta = _system(N, 900 + seed, dominance=float(rng.choice([0.05, 0.2, 2.0])))
    want = _reference(dl, d, du, b, alpha, beta)
    nz = _csc_nzval_fast(dl, d, du)
    tol = 1e-11 * max(1.0, np.max(np.abs(want)))
    # whole
    solver = fd.TridiagSolver(N, layout)
    J = fd.Tridiagonal(_dev(dl), _dev(d), _dev(du)) if layout == "diagonals" else [_dev(nz)]
    y = torch.full((N,), float("nan"), dtype=torch.float64, device="cuda")
    solver.solve(J, _dev(b), y, alpha, beta)
    assert np.max(np.abs(y.cpu().numpy() - want)) <= tol
    # sharded, uneven cuts
    W = int(rng.integers(2, 6))
    if N < 2 * W:
        return
    cuts = np.concatenate([[0], np.sort(rng.choice(np.arange(1, N), size=W - 1, replace=False)), [N]]).astype(np.int64)
    packets = torch.full((W, 8), float("nan"), dtype=torch.float64, device="cuda")
    ranks = []
    for r in range(W):
        c0, c1 = int(cuts[r]), int(cuts[r + 1])
        if layout == "diagonals":
            Jr = fd.Tridiagonal(_dev(dl[c0:min(c1, N - 1)]), _dev(d[c0:c1]), _dev(du[max(c0 - 1, 0):c1 - 1]))
        else:
            e0 = 3 * c0 - 1 if c0 > 0 else 0
            e1 = 3 * c1 - 1 if c1 < N else 3 * N - 2
            Jr = [_dev(nz[e0:e1])]
        sr = fd.TridiagSolver(N, layout, rows=(c0, c1))
        bl = _dev(b[c0:c1])
        sr.interface(Jr, bl, packets[r], alpha, beta)
        ranks.append((sr, Jr, bl, c0, c1))
    got = np.full(N, np.nan)
    for r, (sr, Jr, bl, c0, c1) in enumerate(ranks):
        yl = torch.full((c1 - c0,), float("nan"), dtype=torch.float64, device="cuda")
        sr.finish(Jr, bl, packets, r, W, yl, alpha, beta)
        got[c0:c1] = yl.cpu().numpy()
    assert not np.isnan(got).any() and np.max(np.abs(got - want)) <= tol, (N, layout, W, cuts)


def test_solve_through_a_single_rank_communicator():
    N = 20011
    dl, d, du, b, alpha, beta = _system(N, 5)
    want = _reference(dl, d, du, b, alpha, beta)
    ctx = fd.Context.default()
    comm = fd.Comm(ctx, 1, 0, fd.Comm.unique_id())
    solver = fd.TridiagSolver(N, "diagonals", rows=(0, N))
    y = torch.full((N,), float("nan"), dtype=torch.float64, device="cuda")
    solver.solve(fd.Tridiagonal(_dev(dl), _dev(d), _dev(du)), _dev(b), y, alpha, beta, comm=comm)   # interface -> all-gather -> finish
    assert np.max(np.abs(y.cpu().numpy() - want)) <= 1e-11 * np.max(np.abs(want))
    # a solver for a row range refuses to run without a communicator
    part = fd.TridiagSolver(N, "diagonals", rows=(0, N // 2))
    with pytest.raises(fd.lib.FdError):
        part.solve(fd.Tridiagonal(_dev(dl), _dev(d), _dev(du)), _dev(b), y, alpha, beta)


@pytest.mark.parametrize("storage", ["Tridiagonal", "SparseMatrixCSC"])
def test_jacobian_then_solve_on_the_device(oracle, storage):
    # one implicit-Euler / Rosenbrock stage: W = I - gamma*J(x) with J from finite_difference_jacobian! (coloured, on the
    # device) consumed where it lies; reference: the CPU oracle's Jacobian + scipy's banded solve
    N, gamma = 200003, 0.05
    colors = P.cyclic_colors(N, 3)
    xh = np.random.default_rng(1).random(N)
    bh = np.random.default_rng(2).random(N)
    x, b = _dev(xh), _dev(bh)
    f = fd.BuiltinF("tridiag_nl", N)
    colptr, rowval = P.tridiag_csc(N)
    ref = oracle.jacobian("central", oracle.Fixture("tridiag_nl", N), xh, colors, kind=oracle.PAT_CSC_COMMON, colptr=colptr,
                          rowval=rowval)["out"]
    jj = np.arange(N)
    dref = ref[np.where(jj > 0, 3 * jj, 0)]
    duref, dlref = ref[3 * jj[1:] - 1], ref[3 * jj[:-1] + 1]
    want = _reference(dlref, dref, duref, bh, 1.0, -gamma)
    y = torch.full((N,), float("nan"), dtype=torch.float64, device="cuda")
    if storage == "Tridiagonal":
        J = fd.Tridiagonal(torch.empty(N - 1, dtype=torch.float64, device="cuda"), torch.empty(N, dtype=torch.float64, device="cuda"),
                           torch.empty(N - 1, dtype=torch.float64, device="cuda"))
        fd.finite_difference_jacobian_b(J, f, x, "central", colorvec=colors)
        fd.TridiagSolver(N, "diagonals").solve(J, b, y, 1.0, -gamma)
    else:
        J = fd.SparseMatrixCSC(N, N, colptr, rowval, torch.empty(rowval.size, dtype=torch.float64, device="cuda"))
        fd.finite_difference_jacobian_b(J, f, x, "central", colorvec=colors)
        fd.TridiagSolver(N, "csc").solve(J, b, y, 1.0, -gamma)
    got = y.cpu().numpy()
    assert np.max(np.abs(got - want)) <= 1e-8 * np.max(np.abs(want))   # J itself agrees with the oracle to ~1e-9
    # residual with the oracle's matrix
    res = (1.0 - gamma * dref) * got
    res[1:] += -gamma * dlref * got[:-1]
    res[:-1] += -gamma * duref * got[1:]
    assert np.max(np.abs(res - bh)) <= 1e-8


def test_float32_solver():
    N = 50021
    dl, d, du, b, alpha, beta = _system(N, 77, dominance=1.0)
    want = _reference(dl, d, du, b, alpha, beta)
    f32 = torch.float32
    solver = fd.TridiagSolver(N, "diagonals", dtype=np.float32)
    y = torch.full((N,), float("nan"), dtype=f32, device="cuda")
    solver.solve(fd.Tridiagonal(_dev(dl, f32), _dev(d, f32), _dev(du, f32)), _dev(b, f32), y, alpha, beta)
    assert np.max(np.abs(y.cpu().numpy().astype(np.float64) - want)) <= 2e-6 * max(1.0, np.max(np.abs(want)))


def test_solver_flags_systems_that_are_not_diagonally_dominant():
    # the elimination does not pivot: a solve whose matrix has a row with |b| < |a| + |c| raises the solver's status word
    # (fd_tridiag_solver_status); the Rosenbrock matrix I - gamma J of the diffusion fixture does not
    N = 100_003
    dl = torch.ones(N - 1, dtype=torch.float64, device="cuda")
    d = torch.full((N,), -2.0, dtype=torch.float64, device="cuda")
    du = torch.ones(N - 1, dtype=torch.float64, device="cuda")
    b = torch.ones(N, dtype=torch.float64, device="cuda")
    y = torch.empty_like(b)
    s = fd.TridiagSolver(N, "diagonals")
    s.solve([dl, d, du], b, y, alpha=1.0, beta=-0.05)          # I + 0.05 * tridiag(-1, 2, -1): dominant
    assert s.status() == 0
    s.solve([dl, d, du], b, y, alpha=0.0, beta=1.0)            # tridiag(1, -2, 1) itself: |b| == |a| + |c|, weakly dominant -> fine
    assert s.status() == 0
    d2 = d.clone(); d2[N // 2] = 0.5
    s.solve([dl, d2, du], b, y, alpha=0.0, beta=1.0)           # one row with |0.5| < 2
    assert s.status() == 1 and bool(torch.isnan(y).all())      # refused: never a silently wrong solution
    s.set_policy(True)                                         # the caller vouches for the matrix: the elimination's result
    s.solve([dl, d2, du], b, y, alpha=0.0, beta=1.0)
    assert s.status() == 1 and not bool(torch.isnan(y).any())
    s.set_policy(False)
    tiny = fd.TridiagSolver(100, "diagonals")                  # a system that starts at the top level is guarded too
    yt = torch.empty(100, dtype=torch.float64, device="cuda")
    dd = d[:100].clone(); dd[50] = 0.5
    tiny.solve([dl[:99], dd, du[:99]], b[:100], yt, alpha=0.0, beta=1.0)
    assert tiny.status() == 1 and bool(torch.isnan(yt).all())
    tiny.solve([dl[:99], d[:100].clone(), du[:99]], b[:100], yt, alpha=1.0, beta=-0.05)
    assert tiny.status() == 0 and not bool(torch.isnan(yt).any())
    s.solve([dl, d, du], b, y, alpha=1.0, beta=-0.05)          # the flag belongs to the LAST solve
    assert s.status() == 0


def test_a_rank_that_refuses_refuses_for_every_rank():
    # the multi-rank solve in its two phases, three "ranks" on this GPU: ONE rank's rows contain a row that is not diagonally dominant.
    # Its packet carries NaN: every rank's y is NaN and every rank's status says why -- bit 0 on the rank that owns the row, bit 1
    # ("a peer refused") on all of them.  (Until round 6 the other ranks returned ordinary numbers computed from the refused rank's
    # unreliable interface values, with status 0.)
    N, W = 90_000, 3
    dl, d, du, b, alpha, beta = _system(N, 9, dominance=1.0)
    d = d.copy()
    cuts = [0, 30_000, 60_000, N]
    bad_row = 45_000                                   # rank 1's
    d[bad_row] = (abs(beta * dl[bad_row - 1]) + abs(beta * du[bad_row])) * 0.25 / beta - alpha / beta      # |alpha + beta d| = a quarter of the off-diagonal sum
    packets = torch.full((W, 8), float("nan"), dtype=torch.float64, device="cuda")
    ranks = []
    for r in range(W):
        c0, c1 = cuts[r], cuts[r + 1]
        J = fd.Tridiagonal(_dev(dl[c0:min(c1, N - 1)]), _dev(d[c0:c1]), _dev(du[max(c0 - 1, 0):c1 - 1]))
        solver = fd.TridiagSolver(N, "diagonals", rows=(c0, c1))
        bl = _dev(b[c0:c1])
        solver.interface(J, bl, packets[r], alpha, beta)
        ranks.append((solver, J, bl, c0, c1))
    for r, (solver, J, bl, c0, c1) in enumerate(ranks):
        yl = torch.zeros(c1 - c0, dtype=torch.float64, device="cuda")
        solver.finish(J, bl, packets, r, W, yl, alpha, beta)
        st = solver.status()
        assert bool(torch.isnan(yl).all()), r
        assert st == (3 if r == 1 else 2), (r, st)
