"""The consumer for wider bands (round 6): fd_banded_solve_async -- (alpha I + beta J) y = b for a banded J (l, u <= 4) in the storage
the banded plans write, block cyclic reduction on the device -- against SciPy's banded LU, and end to end behind a Jacobian the library
has just computed."""
import numpy as np
import pytest
import scipy.linalg

import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _band(N, l, u, rng, dominant=True):
    """BandedMatrix data (l+u+1) x N column-major: data[u + i - j, j] = A[i, j]; slots outside the matrix hold 0."""
    w = l + u + 1
    data = rng.standard_normal((w, N))
    for j in range(N):
        for k in range(w):
            i = j - u + k
            if i < 0 or i >= N:
                data[k, j] = 0.0
    return data


def _scipy_solve(data, N, l, u, alpha, beta, b):
    ab = beta * data.copy()
    ab[u, :] += alpha          # row u of the (l+u+1) x N band holds the diagonal (scipy's layout is BandedMatrices' layout)
    return scipy.linalg.solve_banded((l, u), ab, b)


def _csc_values(data, N, l, u):
    out = []
    for j in range(N):
        for i in range(max(0, j - u), min(N - 1, j + l) + 1):
            out.append(data[u + i - j, j])
    return np.array(out)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("layout", ["banded", "csc"])
@pytest.mark.parametrize("N,l,u", [(1, 1, 1), (2, 2, 2), (7, 2, 2), (1000, 2, 2), (4097, 2, 1), (100003, 1, 2), (50001, 3, 3), (30000, 4, 4),
                                   (2049, 0, 2), (3000, 4, 0), (10 ** 6 + 1, 2, 2)])
def test_banded_solve_matches_scipy(dtype, layout, N, l, u):
    rng = np.random.default_rng(N + 10 * l + u)
    data = _band(N, l, u, rng)
    b = rng.standard_normal(N)
    gamma = 0.05                                   # W = I - gamma J: diagonally dominant for |J| ~ 1
    data_t = data.astype(dtype)
    b_t = b.astype(dtype)
    ref = _scipy_solve(data_t.astype(np.float64), N, l, u, 1.0, -gamma, b_t.astype(np.float64))
    vals = data_t.T.reshape(-1) if layout == "banded" else _csc_values(data_t, N, l, u)
    Jd = torch.as_tensor(np.ascontiguousarray(vals), device="cuda")
    bd = torch.as_tensor(b_t, device="cuda")
    y = torch.full((N,), float("nan"), dtype=Jd.dtype, device="cuda")
    s = fd.BandedSolver(N, l, u, layout=layout, dtype=dtype)
    for _ in range(2):
        y.fill_(float("nan"))
        s.solve(Jd, bd, y, alpha=1.0, beta=-gamma)
        assert s.status() == 0
        got = y.cpu().numpy().astype(np.float64)
        tol = 1e-11 if dtype == np.float64 else 2e-5
        assert np.max(np.abs(got - ref)) <= tol * max(1.0, np.max(np.abs(ref))), (N, l, u, np.max(np.abs(got - ref)))


def test_banded_solve_refuses_a_system_without_diagonal_dominance():
    N, l, u = 5000, 2, 2
    rng = np.random.default_rng(3)
    data = _band(N, l, u, rng)
    Jd = torch.as_tensor(np.ascontiguousarray(data.T.reshape(-1)), device="cuda")
    bd = torch.as_tensor(rng.standard_normal(N), device="cuda")
    y = torch.zeros(N, dtype=torch.float64, device="cuda")
    s = fd.BandedSolver(N, l, u)
    s.solve(Jd, bd, y, alpha=1.0, beta=-5.0)          # I - 5 J: not dominant
    assert s.status() & 1
    assert bool(torch.isnan(y).all())
    s.set_policy(True)                                  # the caller vouches: the elimination's result, flag still raised
    s.solve(Jd, bd, y, alpha=1.0, beta=-5.0)
    assert s.status() & 1
    assert not bool(torch.isnan(y).any())
    s.set_policy(False)
    s.solve(Jd, bd, y, alpha=1.0, beta=-0.01)          # a dominant system on the same solver: clean again
    assert s.status() == 0 and not bool(torch.isnan(y).any())


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


@pytest.mark.parametrize("storage", ["banded", "csc"])
def test_implicit_step_on_a_pentadiagonal_jacobian_the_library_computed(storage):
    # the path end to end for a BandedMatrix jac_prototype: the coloured Jacobian of a pentadiagonal residual (a functor compiled at run
    # time, stored by its own launch) lands in BandedMatrix data / CSC nzval, and W y = b with W = I - gamma J is solved on that storage
    import struct
    N, l, u = 200001, 2, 2
    colors = P.cyclic_colors(N, l + u + 1)
    x = np.random.default_rng(8).random(N) * 0.5
    f = fd.JitF(PENTA, "Penta", N, N, params=struct.pack("q", N))
    if storage == "banded":
        J = fd.BandedMatrix(torch.zeros((N, 5), dtype=torch.float64, device="cuda").t(), N, l, u)
        sp, out = None, torch.zeros(5 * N, dtype=torch.float64, device="cuda")
    else:
        cp, rv = P.banded_csc(N, N, l, u)
        J = fd.SparseMatrixCSC(N, N, cp, rv, None)
        sp, out = J, torch.zeros(rv.size, dtype=torch.float64, device="cuda")
    plan = fd.make_plan(J, sp, colors, "forward")
    plan.set_lazy(f)
    plan.jacobian(f, torch.as_tensor(x, device="cuda"), [out])
    gamma = 0.02
    b = np.random.default_rng(9).standard_normal(N)
    y = torch.full((N,), float("nan"), dtype=torch.float64, device="cuda")
    s = fd.BandedSolver(N, l, u, layout=storage)
    s.solve(out, torch.as_tensor(b, device="cuda"), y, alpha=1.0, beta=-gamma)
    assert s.status() == 0
    # reference: the analytic Jacobian of the residual, solved by SciPy
    xs = x
    i = np.arange(N)
    p2 = np.where(i + 2 < N, np.roll(xs, -2), 0.0)
    ab = np.zeros((5, N))               # scipy layout: ab[u + i - j, j] = A[i, j]
    ab[u + 2, : N - 2] = 1.0                                    # d f_i / d x_{i-2}: rows i = j + 2
    ab[u + 1, : N - 1] = -4.0                                   # d f_i / d x_{i-1}
    ab[u, :] = 6.0 + 2.0 * xs * p2                              # d f_i / d x_i
    ab[u - 1, 1:] = -4.0                                        # d f_i / d x_{i+1}: rows i = j - 1
    ab[u - 2, 2:] = 1.0 + xs[: N - 2] ** 2                      # d f_i / d x_{i+2}: row i = j - 2, (1 + x_i^2)
    W = -gamma * ab
    W[u, :] += 1.0
    ref = scipy.linalg.solve_banded((l, u), W, b)
    got = y.cpu().numpy()
    assert np.max(np.abs(got - ref)) <= 1e-6 * max(1.0, np.max(np.abs(ref)))     # (the Jacobian is a forward difference: 1e-6)
    # and against SciPy on the very values the library stored: the solver itself to 1e-11
    vals = out.cpu().numpy()
    if storage == "banded":
        stored = vals.reshape(N, 5).T.copy()
    else:
        stored = np.zeros((5, N)); k = 0
        for j in range(N):
            for r in range(max(0, j - u), min(N - 1, j + l) + 1):
                stored[u + r - j, j] = vals[k]; k += 1
    for j in range(N):                      # slots outside the matrix hold whatever the storing kernel left: not part of the matrix
        for k in range(5):
            if not (0 <= j - u + k < N):
                stored[k, j] = 0.0
        if j > 4 and j < N - 5:
            break
    for j in range(N - 5, N):
        for k in range(5):
            if not (0 <= j - u + k < N):
                stored[k, j] = 0.0
    W2 = -gamma * stored
    W2[u, :] += 1.0
    ref2 = scipy.linalg.solve_banded((l, u), W2, b)
    assert np.max(np.abs(got - ref2)) <= 1e-11 * max(1.0, np.max(np.abs(ref2)))
