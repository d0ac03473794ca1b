"""The consumer for block-banded Jacobians (round 6): fd_blocktridiag_solve_async -- (alpha I + beta J) y = b for a block-tridiagonal J
of dense b x b blocks in BlockBandedMatrix data, block cyclic reduction on the device -- against SciPy's sparse LU, and end to end
behind the block-coupled Jacobian of BASELINE's config 5."""
import numpy as np
import pytest
import scipy.sparse
import scipy.sparse.linalg

import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _sparse_from_data(lay, data, alpha, beta):
    """alpha I + beta J as a SciPy CSC matrix from BlockBandedMatrix data (uniform blocks, block bandwidths (1, 1))."""
    b = int(lay.blk_sizes[0]); nb = lay.nblk
    rows, cols, vals = [], [], []
    rr, cc = np.meshgrid(np.arange(b), np.arange(b), indexing="ij")
    for J in range(nb):
        K0, K1 = max(0, J - 1), min(nb - 1, J + 1)
        st = int(lay.block_strides[J])
        s0 = int(lay.block_starts[(lay.bu + K0 - J) + (lay.bl + lay.bu + 1) * J]) - 1
        panel = np.asarray(data[s0:s0 + st * b]).reshape((st, b), order="F")
        for K in range(K0, K1 + 1):
            blk = panel[(K - K0) * b:(K - K0 + 1) * b, :]
            rows.append((K * b + rr).ravel()); cols.append((J * b + cc).ravel()); vals.append(beta * blk.ravel())
    N = nb * b
    A = scipy.sparse.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N)).tocsc()
    return A + alpha * scipy.sparse.identity(N, format="csc")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nb,b", [(1, 32), (2, 32), (3, 8), (7, 32), (100, 32), (1000, 32), (333, 16), (64, 5), (4097, 4), (50, 1)])
def test_block_tridiagonal_solve_matches_scipy(dtype, nb, b):
    rng = np.random.default_rng(nb + b)
    lay = P.BlockBandedLayout([b] * nb, 1, 1)
    data = rng.standard_normal(lay.data_len).astype(dtype)
    rhs = rng.standard_normal(nb * b).astype(dtype)
    gamma = 0.2 / (3 * b)                                  # I - gamma J: rows of ~3b entries of size ~1 stay diagonally dominant
    A = _sparse_from_data(lay, data.astype(np.float64), 1.0, -gamma)
    ref = scipy.sparse.linalg.spsolve(A, rhs.astype(np.float64))
    Jd = torch.as_tensor(data, device="cuda")
    bd = torch.as_tensor(rhs, device="cuda")
    y = torch.full((nb * b,), float("nan"), dtype=Jd.dtype, device="cuda")
    s = fd.BlockTridiagSolver(nb, b, dtype=dtype)
    for _ in range(2):
        y.fill_(float("nan"))
        s.solve(Jd, bd, y, alpha=1.0, beta=-gamma)
        assert s.status() == 0
        got = y.cpu().numpy().astype(np.float64)
        tol = 1e-11 if dtype == np.float64 else 2e-5
        assert np.max(np.abs(got - ref)) <= tol * max(1.0, np.max(np.abs(ref))), (nb, b, np.max(np.abs(got - ref)))


def test_block_tridiagonal_solve_refuses_without_dominance():
    nb, b = 40, 32
    rng = np.random.default_rng(1)
    lay = P.BlockBandedLayout([b] * nb, 1, 1)
    Jd = torch.as_tensor(rng.standard_normal(lay.data_len), device="cuda")
    bd = torch.as_tensor(rng.standard_normal(nb * b), device="cuda")
    y = torch.zeros(nb * b, dtype=torch.float64, device="cuda")
    s = fd.BlockTridiagSolver(nb, b)
    s.solve(Jd, bd, y, alpha=1.0, beta=-1.0)
    assert s.status() & 1 and bool(torch.isnan(y).all())
    s.set_policy(True)
    s.solve(Jd, bd, y, alpha=1.0, beta=-1.0)
    assert s.status() & 1 and not bool(torch.isnan(y).all())
    s.set_policy(False)
    s.solve(Jd, bd, y, alpha=1.0, beta=-0.001)
    assert s.status() == 0 and not bool(torch.isnan(y).any())


def test_implicit_step_behind_the_block_coupled_jacobian():
    # BASELINE's config 5 in small: the complex-step Jacobian of the block-coupled residual lands in BlockBandedMatrix data, and
    # (I - gamma J) y = b is solved on that storage; reference: SciPy on the very values the library stored
    nb, b = 300, 32
    lay = P.BlockBandedLayout([b] * nb, 1, 1)
    N = lay.N
    colors = lay.colors()
    f = fd.BuiltinF("blockcoupled", nb, b)
    J = fd.BlockBandedMatrix(torch.zeros(lay.data_len, dtype=torch.float64, device="cuda"), lay)
    x = torch.as_tensor(np.random.default_rng(2).random(N), device="cuda")
    fd.finite_difference_jacobian_b(J, f, x, "complex", colorvec=colors)
    vals = J.data.cpu().numpy()
    scale = np.max(np.abs(vals))
    gamma = 0.2 / (3 * b * scale)
    rhs = np.random.default_rng(3).standard_normal(N)
    y = torch.full((N,), float("nan"), dtype=torch.float64, device="cuda")
    s = fd.BlockTridiagSolver(nb, b)
    s.solve(J, torch.as_tensor(rhs, device="cuda"), y, alpha=1.0, beta=-gamma)
    assert s.status() == 0
    ref = scipy.sparse.linalg.spsolve(_sparse_from_data(lay, vals, 1.0, -gamma), rhs)
    assert np.max(np.abs(y.cpu().numpy() - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref)))
