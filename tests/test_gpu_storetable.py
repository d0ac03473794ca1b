"""The storing launch for GENERAL SparseMatrixCSC patterns (round 4): the plan's compact copy of the pattern (FD_PLAN_STORE_CSC,
fd_csc_store in include/fdjac_device.h) and the functor families that store column by column through it
(FD_F_LAP7: 3-D 7-point stencil; FD_F_SPARSE: any pattern).  Every case is checked three ways: bit for bit against the hand-over
path (materialised points -> plain f! -> k_decompress_*: the same operations on the same operands), against the CPU oracle within
the stated tolerance, and the number of f! evaluations against the reference's (1 + C / 2C).
Reference: src/jacobians.jl:559-568, 600-609; ext/FiniteDiffSparseArraysExt.jl:38-47."""
import os
import numpy as np
import pytest

import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
EPS64 = np.finfo(np.float64).eps


def _dev(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device="cuda")


def stencil7_csc(nx, ny, nz):
    N = nx * ny * nz
    k = np.arange(N, dtype=np.int64)
    i, j, l = k % nx, (k // nx) % ny, k // (nx * ny)
    has = np.stack([l > 0, j > 0, i > 0, np.ones(N, bool), i < nx - 1, j < ny - 1, l < nz - 1], axis=1)
    rows = np.stack([k - nx * ny, k - nx, k - 1, k, k + 1, k + nx, k + nx * ny], axis=1)
    colptr = np.empty(N + 1, np.int64)
    colptr[0] = 1
    np.cumsum(has.sum(axis=1), out=colptr[1:])
    colptr[1:] += 1
    return colptr, (rows[has] + 1).astype(np.int64), ((i + 2 * j + 3 * l) % 7 + 1).astype(np.int64)


def lap7_np(fx, xx, nx, ny, nz):
    """FD_F_LAP7 in numpy, the kernel's operation order (real or complex)."""
    X = xx.reshape(nz, ny, nx)
    Z = np.zeros_like(X)
    d, s, w, e, n, u = (Z.copy() for _ in range(6))
    d[1:] = X[:-1]
    u[:-1] = X[1:]
    s[:, 1:] = X[:, :-1]
    n[:, :-1] = X[:, 1:]
    w[:, :, 1:] = X[:, :, :-1]
    e[:, :, :-1] = X[:, :, 1:]
    fx[:] = (((((((d + s) + w) + e) + n) + u) - 6 * X) + (X * X) * e).reshape(-1)


def sparse_np_factory(M, N, colptr, rowval):
    """FD_F_SPARSE in numpy: f_r = sum over the entries (r, j), ascending j, left to right, of w(r, j) phi(x_j)."""
    cols = P.csc_cols(colptr) - 1
    rows = rowval - 1
    order = np.lexsort((cols, rows))             # by row, then by column
    rs, cs = rows[order], cols[order]
    cnt = np.bincount(rs, minlength=M)
    start = np.concatenate([[0], np.cumsum(cnt)])[:-1]
    maxlen = int(cnt.max()) if cnt.size else 0
    w = 1.0 + 0.125 * ((rs + 3 * cs) & 7)

    def f(fx, xx):
        t = w * (xx[cs] + (0.25 * xx[cs]) * xx[cs])
        out = np.zeros(M, dtype=xx.dtype)
        for k in range(maxlen):
            sel = np.nonzero(cnt > k)[0]
            out[sel] = t[start[sel] + k] if k == 0 else out[sel] + t[start[sel] + k]
        fx[:] = out
    return f


def _tol_ok(g, c, eps_min, fscale, what, rtol=1e-6):
    atol = 16 * EPS64 * fscale / abs(eps_min)
    bad = np.abs(g - c) > rtol * np.abs(c) + atol
    assert not bad.any(), "%s: %d entries off, worst %.3e (atol %.1e)" % (what, int(bad.sum()), float(np.max(np.abs(g - c))), atol)


def _run_pair(J, colors, fdtype, f, x, nnz, **plan_kw):
    """The same Jacobian through the column-centric store and through the hand-over path; returns (stored, handed over, plans)."""
    p_store = fd.make_plan(J, J, colors, fdtype, store_csc=True, **plan_kw)
    p_store.set_lazy(f)
    p_hand = fd.make_plan(J, J, colors, fdtype, **plan_kw)
    a = _dev(np.full(nnz, np.nan))
    b = _dev(np.full(nnz, np.nan))
    n0 = f.fcalls
    p_store.jacobian(f, x, [a])
    n1 = f.fcalls
    p_hand.jacobian(f, x, [b])
    n2 = f.fcalls
    return a, b, p_store, p_hand, (n1 - n0, n2 - n1)


@pytest.mark.parametrize("fdtype", ["forward", "central"])
# (100, 6, 5): wavefronts of all three kinds -- two cells from every face, across the end of a grid row, on a j / l face
@pytest.mark.parametrize("shape", [(20, 20, 20), (7, 5, 3), (33, 4, 9), (1, 1, 40), (64, 64, 2), (100, 6, 5)])
def test_lap7_column_store_bit_identical_and_oracle(oracle, fdtype, shape):
    nx, ny, nz = shape
    N = nx * ny * nz
    colptr, rowval, colors = stencil7_csc(nx, ny, nz)
    x = np.random.default_rng(nx + 10 * ny + 100 * nz).random(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    f = fd.BuiltinF("lap7", nx, ny, nz)
    assert f.lazy_caps == fd.lib.LAZY_CAP_STORE_CSC | fd.lib.LAZY_CAP_STORE_CSC_BASE | fd.lib.LAZY_CAP_STORE_CSC_COMPLEX
    a, b, ps, ph, calls = _run_pair(J, colors, fdtype, f, _dev(x), rowval.size)
    if fdtype == "forward":
        # FD_LAZY_CAP_STORE_CSC_BASE: no plain f(x) launch, the storing launch forms the unperturbed rows itself -- one launcher
        # invocation in all, the same 1 + C evaluations counted; withheld: one plain evaluation first; a caller's f_in: used as it is
        l0 = f.counts()[0]
        out1 = _dev(np.full(rowval.size, np.nan))
        ps.jacobian(f, _dev(x), [out1])
        assert f.counts()[0] - l0 == 1
        ps.set_lazy(f, csc_base=False)
        l0 = f.counts()[0]
        out2 = _dev(np.full(rowval.size, np.nan))
        ps.jacobian(f, _dev(x), [out2])
        assert f.counts()[0] - l0 == 2
        assert torch.equal(out1.view(torch.int64), a.view(torch.int64)) and torch.equal(out2.view(torch.int64), a.view(torch.int64))
        ps.set_lazy(f)
        fin = _dev(np.random.default_rng(5).random(N))          # (an arbitrary f_in: the subtrahend the reference would use)
        o3, o4 = _dev(np.full(rowval.size, np.nan)), _dev(np.full(rowval.size, np.nan))
        ps.jacobian(f, _dev(x), [o3], f_in=fin)
        ph.jacobian(f, _dev(x), [o4], f_in=fin)
        assert torch.equal(o3.view(torch.int64), o4.view(torch.int64)) and not torch.equal(o3.view(torch.int64), a.view(torch.int64))
    C = int(colors.max())
    assert ps.info(fd.lib.INFO_STORE_CSC) == rowval.size and ps.info(fd.lib.INFO_LAZY_STORE) == 1
    assert ph.info(fd.lib.INFO_STORE_CSC) == 0 and ph.info(fd.lib.INFO_LAZY_STORE) == 0
    assert calls[0] == calls[1] == (C + 1 if fdtype == "forward" else 2 * C)
    assert not torch.isnan(a).any()
    assert torch.equal(a.view(torch.int64), b.view(torch.int64))
    of = oracle.PyF(lambda fx, xx: lap7_np(fx, xx, nx, ny, nz), N, N)
    ref = oracle.jacobian(fdtype, of, x, colors, M=N, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    _tol_ok(a.cpu().numpy(), ref["out"], np.min(np.abs(ps.epsilons())), 14.0, "lap7 " + fdtype)


def _random_pattern(M, N, per_col, reach, seed):
    """`per_col` entries per column at random rows within +-reach of the scaled diagonal (duplicates removed), some empty columns."""
    rng = np.random.default_rng(seed)
    centre = (np.arange(N) * (M / N)).astype(np.int64)
    rows = np.sort(centre[:, None] + rng.integers(-reach, reach + 1, size=(N, per_col)), axis=1)
    keep = (rows >= 0) & (rows < M)              # (entries outside the matrix are dropped, not clipped: clipping would pile them into dense rows)
    keep[:, 1:] &= rows[:, 1:] != rows[:, :-1]
    keep[rng.random(N) < 0.03] = False           # empty columns
    cnt = keep.sum(axis=1)
    colptr = np.empty(N + 1, np.int64)
    colptr[0] = 1
    np.cumsum(cnt, out=colptr[1:])
    colptr[1:] += 1
    return colptr, (rows[keep] + 1).astype(np.int64)


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("case", [(300, 300, 4, 9, 1), (257, 411, 3, 20, 2), (411, 257, 5, 6, 3), (5000, 5000, 6, 300, 4), (40, 40, 1, 0, 5)])
def test_sparse_family_column_store_bit_identical_and_oracle(oracle, fdtype, case):
    M, N, per_col, reach, seed = case
    colptr, rowval = _random_pattern(M, N, per_col, reach, seed)
    J = fd.SparseMatrixCSC(M, N, colptr, rowval, None)
    colors = fd.matrix_colors(J)
    x = np.random.default_rng(seed).random(N) + 0.1
    f = fd.BuiltinF.sparse(M, N, colptr, rowval)
    a, b, ps, ph, calls = _run_pair(J, colors, fdtype, f, _dev(x), rowval.size)
    C = int(colors.max())
    assert ps.info(fd.lib.INFO_STORE_CSC) == rowval.size and ps.info(fd.lib.INFO_LAZY_STORE) == 1
    assert calls[0] == calls[1] == (C + 1 if fdtype == "forward" else 2 * C)
    assert not torch.isnan(a).any()
    assert torch.equal(a.view(torch.int64), b.view(torch.int64))
    of = oracle.PyF(sparse_np_factory(M, N, colptr, rowval), M, N)
    ref = oracle.jacobian(fdtype, of, x, colors, M=M, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    fs = float(np.abs(ref["fx"]).max()) if "fx" in ref else 10.0 * per_col
    _tol_ok(a.cpu().numpy(), ref["out"], np.min(np.abs(ps.epsilons())), max(fs, 1.0), "sparse " + fdtype)
    # and the analytic Jacobian: d f_r / d x_j = w(r, j) (1 + 0.5 x_j)
    cols = P.csc_cols(colptr) - 1
    want = (1.0 + 0.125 * (((rowval - 1) + 3 * cols) & 7)) * (1.0 + 0.5 * x[cols])
    assert np.max(np.abs(a.cpu().numpy() - want)) < (5e-6 if fdtype == "forward" else 5e-8) * 10


@pytest.mark.parametrize("shape", [(61, 47), (3, 3), (200, 5)])
def test_user_kernel_stores_a_general_pattern_row_by_row(tmp_path, shape):
    # examples/user_terms_store.hip: a USER's separable residual (nine-point stencil, given by its term) compiled apart from libfdjac
    # against the two public headers; on the plan whose row lists it was bound to (FD_PLAN_STORE_CSC_ROWS, fd_plan_row_lists) its lazy
    # launcher runs fd_csc_store_rows, on another plan fd_csc_store_cols.  examples/user_terms_client.c (plain C) checks: the lists, which
    # kernel ran, both bit-identical to the plain launcher + the library's decompression, analytic values, f! counts.
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc, libdir = os.path.join(root, "include"), os.path.join(root, "finitediff.jl_amd", "lib")
    user_so = str(tmp_path / "libuser_tm.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-I" + inc,
                           os.path.join(root, "examples", "user_terms_store.hip"), "-o", user_so])
    assert "libfdjac" not in subprocess.run(["ldd", user_so], capture_output=True, text=True).stdout
    exe = str(tmp_path / "user_terms_client")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + inc, os.path.join(root, "examples", "user_terms_client.c"), "-o", exe,
                           "-L" + str(tmp_path), "-luser_tm", "-L" + libdir, "-lfdjac", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + str(tmp_path), "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(shape[0]), str(shape[1])], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "user_terms_client ok" in out.stdout and "FAILED" not in out.stdout


@pytest.mark.parametrize("fdtype", ["forward", "central"])
# (3000, 12, 40): ~12 entries per row -- more than a tile's staged run holds: the overflow rows read their lists from memory
@pytest.mark.parametrize("case", [(300, 4, 9, 1), (5000, 6, 300, 4), (70001, 6, 300, 7), (3000, 12, 40, 8), (2000, 3, 700, 9)])
# True: the plan keeps its pattern by rows (store_rows) -- then the verified launches are fd_csc_store_rows on the plan's lists; "ents": the
# entry-parallel form of the same store (fd_csc_store_ents, plans built with the test switch FDJAC_ROWS_ENTS=1)
@pytest.mark.parametrize("lists", [False, True, "ents"])
def test_sparse_family_row_wise_store_after_the_first_call(monkeypatch, fdtype, case, lists):
    if lists == "ents":
        monkeypatch.setenv("FDJAC_ROWS_ENTS", "1")
    # k_f_sparse_store_rows: the first launch on a plan only CHECKS that the plan's pattern is the one the residual was created from (the
    # column kernel stores); once the verdict has reached the host (read back asynchronously) the launches go row by row -- every row's
    # plain terms once, prefix carried, suffix added: the additions of the full evaluation in the same order => the hand-over path's bits.
    N, per_col, reach, seed = case
    colptr, rowval = _random_pattern(N, N, per_col, reach, seed)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    colors = fd.matrix_colors(J)
    f = fd.BuiltinF.sparse(N, N, colptr, rowval)
    ps = fd.make_plan(J, J, colors, fdtype, store_csc=True, store_rows=bool(lists))
    ps.set_lazy(f)
    ph = fd.make_plan(J, J, colors, fdtype)
    rng = np.random.default_rng(seed)
    n_rows = []
    for it in range(5):
        x = _dev(rng.random(N) + 0.1 * (it + 1))
        a, b = _dev(np.full(rowval.size, np.nan)), _dev(np.full(rowval.size, np.nan))
        ps.jacobian(f, x, [a])
        ph.jacobian(f, x, [b])
        torch.cuda.synchronize()
        assert not torch.isnan(a).any() and torch.equal(a.view(torch.int64), b.view(torch.int64)), it
        n_rows.append(f.row_stores())
    assert n_rows[0] == 0 and n_rows[-1] >= 3, n_rows          # (the first call checks; the verdict is on the host by the third at the latest)
    # another plan, whose pattern is NOT the residual's (one more stored entry per column 0: a structural zero of this f): the check
    # fails, the column kernel keeps storing -- correct values, no row-wise launch
    cp2, rv2 = colptr.copy(), rowval.copy()
    extra = int(np.setdiff1d(np.arange(1, min(N, 50) + 1), rowval[cp2[0] - 1:cp2[1] - 1])[0])
    col0 = np.sort(np.append(rowval[cp2[0] - 1:cp2[1] - 1], extra))
    rv2 = np.concatenate([col0, rowval[cp2[1] - 1:]])
    cp2[1:] += 1
    J2 = fd.SparseMatrixCSC(N, N, cp2, rv2, None)
    colors2 = fd.matrix_colors(J2)
    ps2 = fd.make_plan(J2, J2, colors2, fdtype, store_csc=True, store_rows=bool(lists))
    ps2.set_lazy(f)
    ph2 = fd.make_plan(J2, J2, colors2, fdtype)
    before = f.row_stores()
    for it in range(4):
        x = _dev(rng.random(N) + 0.2)
        a, b = _dev(np.full(rv2.size, np.nan)), _dev(np.full(rv2.size, np.nan))
        ps2.jacobian(f, x, [a])
        ph2.jacobian(f, x, [b])
        torch.cuda.synchronize()
        assert not torch.isnan(a).any() and torch.equal(a.view(torch.int64), b.view(torch.int64)), it
    assert f.row_stores() == before


@pytest.mark.parametrize("family", ["sparse", "lap7"])
def test_complex_step_column_store_bit_identical_and_oracle(oracle, family):
    # FD_LAZY_CAP_STORE_CSC_COMPLEX: the complex step through the column store -- every stored entry's row at x + i eps e_j,
    # imag / eps stored by the one launch (src/jacobians.jl:623-648 + ext/FiniteDiffSparseArraysExt.jl:38-47); the bits of the
    # hand-over path (complex points materialised, imag parts decompressed), the oracle to a few ulp, C evaluations
    if family == "sparse":
        M = N = 30011
        colptr, rowval = _random_pattern(M, N, 6, 50, 5)
        f = fd.BuiltinF.sparse(M, N, colptr, rowval)
        J = fd.SparseMatrixCSC(M, N, colptr, rowval, None)
        colors = fd.matrix_colors(J)
    else:
        nx, ny, nz = 23, 11, 7
        M = N = nx * ny * nz
        colptr, rowval, colors = stencil7_csc(nx, ny, nz)
        f = fd.BuiltinF("lap7", nx, ny, nz)
        J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    x = np.random.default_rng(9).random(N) + 0.1
    a, b, ps, ph, calls = _run_pair(J, colors, "complex", f, _dev(x), rowval.size)
    C = int(colors.max())
    assert ps.info(fd.lib.INFO_STORE_CSC) == rowval.size and ps.info(fd.lib.INFO_LAZY_STORE) == 1 and ph.info(fd.lib.INFO_LAZY_STORE) == 0
    assert ps.fcalls_last == C and ph.fcalls_last == C
    assert not torch.isnan(a).any() and torch.equal(a.view(torch.int64), b.view(torch.int64))
    if family == "sparse":
        cols = P.csc_cols(colptr) - 1
        want = (1.0 + 0.125 * (((rowval - 1) + 3 * cols) & 7)) * (1.0 + 0.5 * x[cols])
        assert np.max(np.abs(a.cpu().numpy() - want)) < 1e-13
    # an INVALID colouring: the whole colour's point is formed, as the reference does -- same (meaningless) values as the hand-over path
    bad = np.ones(N, dtype=np.int64)
    a2, b2, ps2, _ph2, _c = _run_pair(J, bad, "complex", f, _dev(x), rowval.size)
    assert ps2.info(fd.lib.INFO_LAZY_STORE) == 1 and torch.equal(a2.view(torch.int64), b2.view(torch.int64))


def test_column_store_windows_chunks_uncoloured_and_invalid_colourings():
    # column windows (multi-GPU shards), colour chunks, colour ownership, columns without a colour, an INVALID colouring, a dense
    # row -- always the bits of the hand-over path
    M = N = 4000
    colptr, rowval = _random_pattern(M, N, 5, 40, 11)
    J = fd.SparseMatrixCSC(M, N, colptr, rowval, None)
    colors = fd.matrix_colors(J)
    C = int(colors.max())
    x = _dev(np.random.default_rng(12).random(N) + 0.1)
    f = fd.BuiltinF.sparse(M, N, colptr, rowval)
    full = _dev(np.full(rowval.size, np.nan))
    ph = fd.make_plan(J, J, colors, "forward")
    ph.jacobian(f, x, [full])
    # column windows
    for (c0, c1) in [(0, 1000), (1000, 1001), (1234, 3999), (3999, 4000)]:
        p = fd.make_plan(J, J, colors, "forward", store_csc=True, col_window=(c0, c1))
        p.set_lazy(f)
        n = int(colptr[c1] - colptr[c0])
        out = _dev(np.full(n, np.nan))
        p.jacobian(f, x, [out])
        assert p.info(fd.lib.INFO_STORE_CSC) == n and (n == 0 or p.info(fd.lib.INFO_LAZY_STORE) == 1)
        assert torch.equal(out.view(torch.int64), full[int(colptr[c0] - 1):int(colptr[c1] - 1)].view(torch.int64))
    # colour chunks (a scratch cap that holds 2 colours at a time) and colour ownership
    p = fd.make_plan(J, J, colors, "central", store_csc=True, scratch_bytes=2 * 2 * 2 * N * 8 + 4096)
    p.set_lazy(f)
    pc = fd.make_plan(J, J, colors, "central")
    o1, o2 = _dev(np.full(rowval.size, np.nan)), _dev(np.full(rowval.size, np.nan))
    p.jacobian(f, x, [o1])
    pc.jacobian(f, x, [o2])
    assert p.info(fd.lib.INFO_NCHUNKS) > 1 and torch.equal(o1.view(torch.int64), o2.view(torch.int64))
    acc = torch.zeros(rowval.size, dtype=torch.float64, device="cuda")
    for (a0, a1) in [(0, 2), (2, C)]:
        po = fd.make_plan(J, J, colors, "forward", store_csc=True, color_range=(a0, a1))
        po.set_lazy(f)
        part = torch.zeros_like(acc)
        po.jacobian(f, x, [part])
        acc += part
    assert torch.equal(acc.view(torch.int64), full.view(torch.int64))
    # columns without a colour: their stored values are 0, everything else as the hand-over path computes it
    c2 = colors.copy()
    c2[[5, 77, 1999]] = 0
    a, b, ps, _ph, _calls = _run_pair(J, c2, "forward", f, x, rowval.size)
    assert ps.info(fd.lib.INFO_STORE_CSC) == rowval.size and torch.equal(a.view(torch.int64), b.view(torch.int64))
    for j in (5, 77, 1999):
        assert torch.all(a[int(colptr[j] - 1):int(colptr[j + 1] - 1)] == 0)
    # a dense row
    cpd, rvd = _random_pattern(300, 300, 3, 5, 21)
    dense = np.zeros((300, 300)); dense[(rvd - 1), P.csc_cols(cpd) - 1] = 1; dense[7, :] = 1
    cpd, rvd = P.csc_from_dense(dense)
    Jd = fd.SparseMatrixCSC(300, 300, cpd, rvd, None)
    fdn = fd.BuiltinF.sparse(300, 300, cpd, rvd)
    a, b, ps, _ph, _calls = _run_pair(Jd, fd.matrix_colors(Jd), "forward", fdn, _dev(np.random.default_rng(5).random(300) + 0.1), rvd.size)
    assert ps.info(fd.lib.INFO_STORE_CSC) == rvd.size and torch.equal(a.view(torch.int64), b.view(torch.int64))
    # an INVALID colouring (every column the same colour): the column-centric store forms the whole colour's point, as the
    # reference does -- the same (meaningless) values as the hand-over path, bit for bit
    bad = np.ones(N, dtype=np.int64)
    a, b, ps, _ph, _calls = _run_pair(J, bad, "forward", f, x, rowval.size)
    assert ps.info(fd.lib.INFO_STORE_CSC) == rowval.size and ps.info(fd.lib.INFO_LAZY_STORE) == 1
    assert torch.equal(a.view(torch.int64), b.view(torch.int64))


def test_column_store_float32_device_pattern_and_dropin():
    nx, ny, nz = 24, 10, 6
    N = nx * ny * nz
    colptr, rowval, colors = stencil7_csc(nx, ny, nz)
    x64 = np.random.default_rng(3).random(N)
    # Float32 instantiation
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    f32 = fd.BuiltinF("lap7", nx, ny, nz, dtype=np.float32)
    x32 = _dev(x64, torch.float32)
    p1 = fd.make_plan(J, J, colors, "central", store_csc=True, dtype=np.float32)
    p1.set_lazy(f32)
    p2 = fd.make_plan(J, J, colors, "central", dtype=np.float32)
    o1 = torch.full((rowval.size,), float("nan"), dtype=torch.float32, device="cuda")
    o2 = o1.clone()
    p1.jacobian(f32, x32, [o1])
    p2.jacobian(f32, x32, [o2])
    assert p1.info(fd.lib.INFO_LAZY_STORE) == 1 and torch.equal(o1.view(torch.int32), o2.view(torch.int32))
    # a device-resident pattern (Int32, as rocSPARSE holds it): the pattern copy is made from the device arrays
    f = fd.BuiltinF("lap7", nx, ny, nz)
    x = _dev(x64)
    nz_d = _dev(np.full(rowval.size, np.nan))
    Jd = fd.DevicePatternCSC(N, N, torch.as_tensor(colptr.astype(np.int32), device="cuda"), torch.as_tensor(rowval.astype(np.int32), device="cuda"), nz_d)
    cache = fd.JacobianCache(x, "forward", colorvec=torch.as_tensor(colors.astype(np.int32), device="cuda"), sparsity=Jd)
    fd.finite_difference_jacobian_b(Jd, f, x, cache)          # the drop-in call asks for the pattern copy by itself (f can store)
    assert cache.last_plan.info(fd.lib.INFO_STORE_CSC) == rowval.size and cache.last_plan.info(fd.lib.INFO_LAZY_STORE) == 1
    ref = _dev(np.full(rowval.size, np.nan))
    fd.make_plan(J, J, colors, "forward").jacobian(f, x, [ref])
    assert torch.equal(nz_d.view(torch.int64), ref.view(torch.int64))
    # the complex step has no column store: the family's plain launcher on materialised complex points
    Jc = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.full(rowval.size, np.nan)))
    fd.finite_difference_jacobian_b(Jc, f, x, "complex", colorvec=colors)
    assert torch.allclose(Jc.nzval, ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [(61, 47), (3, 3), (200, 5)])
def test_user_kernel_stores_a_general_pattern_column_by_column(tmp_path, shape):
    # examples/user_csc_store.hip: a USER's residual on the nine-point stencil, written once as a device functor f(r, X) and
    # compiled apart from libfdjac against the two public headers; inside fd_csc_store_cols (include/fdjac_device.h) it stores
    # the Jacobian column by column.  examples/user_csc_client.c (plain C) checks: pattern copy built and used,
    # bit-identical to the same plan through the user's plain launcher + the library's decompression, analytic values, f! counts.
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc, libdir = os.path.join(root, "include"), os.path.join(root, "finitediff.jl_amd", "lib")
    user_so = str(tmp_path / "libuser_rl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-I" + inc,
                           os.path.join(root, "examples", "user_csc_store.hip"), "-o", user_so])
    assert "libfdjac" not in subprocess.run(["ldd", user_so], capture_output=True, text=True).stdout
    exe = str(tmp_path / "user_csc_client")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + inc, os.path.join(root, "examples", "user_csc_client.c"), "-o", exe,
                           "-L" + str(tmp_path), "-luser_rl", "-L" + libdir, "-lfdjac", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + str(tmp_path), "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(shape[0]), str(shape[1])], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "user_csc_client ok" in out.stdout and "FAILED" not in out.stdout


@pytest.mark.parametrize("fdtype", ["forward", "central"])
def test_lap7_pattern_note_remembers_a_verified_pattern(fdtype):
    # fd_csc_store.note: the 7-point launcher checks on its first launch that the plan's columns hold exactly its stencil and then
    # stops reading row indices.  Repeated calls on one plan (checking launch, then verified launches), new contents of x, a plan
    # whose pattern is a SUPERSET of the stencil (never verified: every call through the checked path), the same plan driven by a
    # launcher of another grid with the same N on a pattern that holds both stencils (the note's key differs per launcher: each
    # is checked and found "not the stencil") -- always the bits of the hand-over path.
    nx, ny, nz = 23, 17, 11
    N = nx * ny * nz
    colptr, rowval, colors = stencil7_csc(nx, ny, nz)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    f = fd.BuiltinF("lap7", nx, ny, nz)
    rng = np.random.default_rng(3)
    ps = fd.make_plan(J, J, colors, fdtype, store_csc=True)
    ps.set_lazy(f)
    ph = fd.make_plan(J, J, colors, fdtype)
    for rep in range(4):
        x = _dev(rng.random(N))
        a, b = _dev(np.full(rowval.size, np.nan)), _dev(np.full(rowval.size, np.nan))
        ps.jacobian(f, x, [a])
        ph.jacobian(f, x, [b])
        assert torch.equal(a.view(torch.int64), b.view(torch.int64)), rep
    # a superset pattern (every column also lists row (k + 5) mod N): valid as a pattern of f, never "the stencil"
    cols = np.repeat(np.arange(N), np.diff(colptr))
    rows = rowval - 1
    extra_r, extra_c = (np.arange(N) + 5) % N, np.arange(N)
    allr, allc = np.concatenate([rows, extra_r]), np.concatenate([cols, extra_c])
    key = np.unique(allc * N + allr)
    c2, r2 = key // N, key % N
    cp2 = np.concatenate([[0], np.cumsum(np.bincount(c2, minlength=N))]).astype(np.int64) + 1
    J2 = fd.SparseMatrixCSC(N, N, cp2, (r2 + 1).astype(np.int64), None)
    col2 = fd.matrix_colors(J2)
    ps2 = fd.make_plan(J2, J2, col2, fdtype, store_csc=True)
    ps2.set_lazy(f)
    ph2 = fd.make_plan(J2, J2, col2, fdtype)
    for rep in range(3):
        x = _dev(rng.random(N))
        a, b = _dev(np.full(r2.size, np.nan)), _dev(np.full(r2.size, np.nan))
        ps2.jacobian(f, x, [a])
        ph2.jacobian(f, x, [b])
        assert ps2.info(fd.lib.INFO_LAZY_STORE) == 1
        assert torch.equal(a.view(torch.int64), b.view(torch.int64)), rep
    # the union of the stencils of two grids with the same N, driven alternately by both launchers
    cpB, rvB, _ = stencil7_csc(ny, nx, nz)
    colsB = np.repeat(np.arange(N), np.diff(cpB))
    key = np.unique(np.concatenate([cols * N + rows, colsB * N + (rvB - 1)]))
    c3, r3 = key // N, key % N
    cp3 = np.concatenate([[0], np.cumsum(np.bincount(c3, minlength=N))]).astype(np.int64) + 1
    J3 = fd.SparseMatrixCSC(N, N, cp3, (r3 + 1).astype(np.int64), None)
    col3 = fd.matrix_colors(J3)
    g = fd.BuiltinF("lap7", ny, nx, nz)
    ps3 = fd.make_plan(J3, J3, col3, fdtype, store_csc=True)
    ph3 = fd.make_plan(J3, J3, col3, fdtype)
    for rep in range(4):
        fn = f if rep % 2 == 0 else g
        ps3.set_lazy(fn)
        x = _dev(rng.random(N))
        a, b = _dev(np.full(r3.size, np.nan)), _dev(np.full(r3.size, np.nan))
        ps3.jacobian(fn, x, [a])
        ph3.jacobian(fn, x, [b])
        assert torch.equal(a.view(torch.int64), b.view(torch.int64)), rep


@pytest.mark.parametrize("seed", list(range(14)))
def test_random_grids_through_the_7_point_column_kernel(seed):
    # Randomised: grids of random shape (thin, flat, degenerate: a dimension of 1 or 2), random column windows, colour chunks,
    # uncoloured columns, valid and invalid colourings, forward (f(x) rows formed inside the launch / from f_in) and central
    # differences, repeated calls on one plan (checking launch, then launches that trust the plan's note) -- the window kernel, its
    # boundary path and the functor fallback must give the bits of the hand-over path every time.  (A grid that degenerates to an
    # exact band -- nx x 1 x 1 -- keeps the band's own store capability, which this launcher does not have: the hand-over path.)
    rng = np.random.default_rng(int(os.environ.get("FDJAC_TEST_SEED_BASE", "7000")) + seed)
    shapes = [(int(rng.integers(3, 40)), int(rng.integers(3, 30)), int(rng.integers(3, 20))), (int(rng.integers(1, 4)), int(rng.integers(1, 60)), int(rng.integers(1, 60))),
              (int(rng.integers(3, 200)), int(rng.integers(1, 3)), int(rng.integers(1, 8))), (int(rng.integers(2, 70)), int(rng.integers(2, 70)), 2)]
    nx, ny, nz = shapes[seed % 4]
    N = nx * ny * nz
    colptr, rowval, colors = stencil7_csc(nx, ny, nz)
    style = rng.random()
    if style < 0.2:
        colors = colors.copy()
        colors[rng.integers(0, N, size=min(5, N))] = 0
    elif style < 0.35:
        colors = np.ones(N, dtype=np.int64)                 # invalid: every column one colour
    elif style < 0.5:
        colors = (np.arange(N) % 11 + 1).astype(np.int64)   # invalid for most grids
    fdtype = "forward" if rng.random() < 0.5 else "central"
    kw = {}
    if rng.random() < 0.35 and N > 8:
        a = int(rng.integers(0, N // 2))
        kw["col_window"] = (a, int(rng.integers(a + 1, N + 1)))
    C = int(colors.max())
    if rng.random() < 0.3 and C > 2:
        kw["scratch_bytes"] = 2 * 2 * 2 * N * 8 + 4096       # two colours per chunk
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    dtype = np.float32 if rng.random() < 0.3 else np.float64        # (fd32_*: the second instantiation of every kernel)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    if dtype == np.float32 and "scratch_bytes" in kw:
        kw["scratch_bytes"] //= 2

    def _dev(a):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=tdt, device="cuda")

    f = fd.BuiltinF("lap7", nx, ny, nz, dtype=dtype)
    ps = fd.make_plan(J, J, colors, fdtype, store_csc=True, dtype=dtype, **kw)
    ps.set_lazy(f)
    ph = fd.make_plan(J, J, colors, fdtype, dtype=dtype, **kw)
    n = ps.out_len(0)
    assert n == ph.out_len(0)
    for rep in range(3):
        x = _dev(rng.random(N) + 0.05)
        fin = _dev(rng.random(N)) if (fdtype == "forward" and rep == 2) else None
        a, b = _dev(np.full(n, np.nan)), _dev(np.full(n, np.nan))
        ps.jacobian(f, x, [a], f_in=fin)
        ph.jacobian(f, x, [b], f_in=fin)
        assert not torch.isnan(b).any() or n == 0
        assert torch.equal(a, b), (nx, ny, nz, fdtype, kw, rep, style, dtype)


@pytest.mark.parametrize("seed", range(24))
def test_sparse_family_row_wise_store_randomised(monkeypatch, seed):
    if seed % 4 == 3:
        monkeypatch.setenv("FDJAC_ROWS_ENTS", "1")      # (the entry-parallel form of the row-wise store on a quarter of the seeds)
    # the row-wise store (k_f_sparse_store_rows) under everything a plan can ask of it: column windows (other ranks' columns are terms
    # of its rows but not its entries), colour chunks and colour ownership, uncoloured columns, a caller's f_in, Float32, rows longer
    # than a tile's staged run or than the register path -- five calls per plan (check, then row-wise), the hand-over path's bits
    rng = np.random.default_rng(int(os.environ.get("FDJAC_TEST_SEED_BASE", "9000")) + seed)
    N = int(rng.choice([257, 1000, 4097, 30011]))
    per_col = int(rng.choice([1, 3, 6, 11, 18]))
    reach = int(rng.choice([1, 7, 60, 300, 650]))
    colptr, rowval = _random_pattern(N, N, per_col, reach, 100 + seed)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    colors = fd.matrix_colors(J)
    if rng.random() < 0.3:
        colors = colors.copy()
        colors[rng.integers(0, N, size=5)] = 0                     # uncoloured columns: zeros
    C = int(colors.max())
    fdtype = "forward" if rng.random() < 0.6 else "central"
    kw = {}
    if rng.random() < 0.4:
        a = int(rng.integers(0, N // 2))
        kw["col_window"] = (a, int(rng.integers(a + 1, N + 1)))
    if rng.random() < 0.3 and C > 3:
        kw["scratch_bytes"] = 2 * 2 * 2 * N * 8 + 4096             # a few colours per chunk
    if rng.random() < 0.25 and C > 2:
        c0 = int(rng.integers(0, C - 1))
        kw["color_range"] = (c0, int(rng.integers(c0 + 1, C + 1)))  # colour ownership: the other colours' entries stay untouched
    dtype = np.float32 if rng.random() < 0.25 else np.float64
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    if dtype == np.float32 and "scratch_bytes" in kw:
        kw["scratch_bytes"] //= 2

    def dev(a):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=tdt, device="cuda")

    f = fd.BuiltinF.sparse(N, N, colptr, rowval, dtype=dtype)
    # (odd seeds: the plan keeps its pattern by rows -- without a column window the verified launches are then fd_csc_store_rows,
    #  the kernel every separable user functor gets, on the plan's lists)
    ps = fd.make_plan(J, J, colors, fdtype, store_csc=True, store_rows=(seed % 2 == 1), dtype=dtype, **kw)
    ps.set_lazy(f)
    ph = fd.make_plan(J, J, colors, fdtype, dtype=dtype, **kw)
    n = ps.out_len(0)
    assert n == ph.out_len(0)
    for rep in range(5):
        x = dev(rng.random(N) + 0.05)
        fin = dev(rng.random(N)) if (fdtype == "forward" and rep == 3) else None
        fill = float(rep) + 0.5                                     # (entries of colours this plan does not own keep what was there)
        a, b = dev(np.full(n, fill)), dev(np.full(n, fill))
        ps.jacobian(f, x, [a], f_in=fin)
        ph.jacobian(f, x, [b], f_in=fin)
        torch.cuda.synchronize()
        assert torch.equal(a, b), (N, per_col, reach, fdtype, kw, rep, dtype)


def test_window_plan_checks_its_colouring_against_all_columns():
    # Found by the sweep above: a colouring that is valid among the columns of a column window but not against the columns outside
    # it (which the colour's point perturbs too).  The plan must not call it valid -- single-coordinate differences would differ
    # from what the reference's colliding columns give.  Grid 87 x 1 x 3, colours k mod 11 + 1, columns [98, 132).
    nx, ny, nz = 87, 1, 3
    N = nx * ny * nz
    colptr, rowval, _ = stencil7_csc(nx, ny, nz)
    colors = (np.arange(N) % 11 + 1).astype(np.int64)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    f = fd.BuiltinF("lap7", nx, ny, nz)
    x = _dev(np.random.default_rng(2).random(N) + 0.05)
    for fdtype in ("forward", "central"):
        ps = fd.make_plan(J, J, colors, fdtype, store_csc=True, col_window=(98, 132))
        ps.set_lazy(f)
        ph = fd.make_plan(J, J, colors, fdtype, col_window=(98, 132))
        a, b = _dev(np.full(ps.out_len(0), np.nan)), _dev(np.full(ps.out_len(0), np.nan))
        ps.jacobian(f, x, [a])
        ph.jacobian(f, x, [b])
        assert ps.info(fd.lib.INFO_LAZY_STORE) == 1
        assert torch.equal(a.view(torch.int64), b.view(torch.int64)), fdtype
