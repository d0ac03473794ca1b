"""Edge cases of the device path (through the C ABI): empty / ragged patterns, degenerate colourings, invalid
inputs and their status codes, failing f! launchers, repeated and concurrent use of plans."""
import threading

import os

import numpy as np
import pytest

import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
FDTYPES = ["forward", "central", "complex"]


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


def _ragged_pattern(M, N, seed):
    """Random pattern with empty columns, empty rows, a full column and single-entry columns."""
    rng = np.random.default_rng(seed)
    A = (rng.random((M, N)) < 0.08).astype(float)
    A[:, [1, N // 2]] = 0          # empty columns
    A[[0, M - 1], :] = 0           # empty rows
    A[1:M - 1, 3] = 1              # a dense column
    A[2, N - 1] = 1
    return A


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("shape", [(57, 41), (41, 57), (300, 300)])
def test_ragged_random_pattern_matches_oracle(oracle, fdtype, shape):
    # a general sparse pattern (no band structure: the gather kernels), greedy colouring, f = A .* sin-ish coupling
    M, N = shape
    A = _ragged_pattern(M, N, 7 + M)
    colptr, rowval = P.csc_from_dense(A)
    J = fd.SparseMatrixCSC(M, N, colptr, rowval, _dev(np.full(rowval.size, np.nan)))
    colors = fd.matrix_colors(J)
    W = np.random.default_rng(3).random((M, N)) * A
    Wt = _dev(W)
    x = np.random.default_rng(4).random(N) + 0.2

    def fn(fx, xx):      # f_i = sum_j W_ij * x_j^2   (complex-analytic)
        fx.copy_(Wt.to(xx.dtype) @ (xx * xx))

    f = fd.TorchF(fn, M, N)
    fd.finite_difference_jacobian_b(J, f, _dev(x), fdtype, colorvec=colors)
    C = int(colors.max())
    assert f.fcalls == {"forward": C + 1, "central": 2 * C, "complex": C}[fdtype]
    of = oracle.PyF(lambda fx, xx: fx.__setitem__(slice(None), W @ (xx * xx)), M, N)
    ref = oracle.jacobian(fdtype, of, x, colors, M=M, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    got = J.nzval.cpu().numpy()
    eps = 2.2e-16 if fdtype == "complex" else fd.default_relstep(fdtype)
    atol = 16 * 2.2e-16 * float(np.abs(W @ (x * x)).max() + 1) / eps
    assert np.all(np.abs(got - ref["out"]) <= 1e-6 * np.abs(ref["out"]) + atol)
    exact = (2 * W * x[None, :])[A != 0]      # column-major order of the stored entries
    assert np.allclose(P.csc_to_dense(M, N, colptr, rowval, got)[A != 0], exact, rtol=2e-5 if fdtype == "forward" else 1e-7, atol=1e-6)


@pytest.mark.parametrize("fdtype", FDTYPES)
def test_empty_pattern_and_empty_window(fdtype):
    # nnz = 0: nothing to write, no f! evaluation beyond what the reference does (it still loops over the colours)
    N = 50
    colptr = np.ones(N + 1, np.int64)
    rowval = np.zeros(0, np.int64)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, _dev(np.zeros(0)))
    f = fd.BuiltinF("tridiag", N)
    fd.finite_difference_jacobian_b(J, f, _dev(np.ones(N)), fdtype, colorvec=P.cyclic_colors(N, 3))
    # an empty column window of a non-empty pattern
    cp, rv = P.tridiag_csc(N)
    Jf = fd.SparseMatrixCSC(N, N, cp, rv)
    plan = fd.make_plan(Jf, Jf, P.cyclic_colors(N, 3), fdtype, col_window=(20, 20))
    assert plan.out_len(0) == 0
    plan.jacobian(fd.BuiltinF("tridiag", N), _dev(np.ones(N)), [_dev(np.zeros(1))[:0]])


@pytest.mark.parametrize("fdtype", FDTYPES)
def test_all_columns_uncoloured(fdtype):
    # maximum(colorvec) < 1: no colour loop at all; fill_matrix! leaves every stored value at zero
    N = 40
    cp, rv = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, cp, rv, _dev(np.full(rv.size, np.nan)))
    f = fd.BuiltinF("tridiag", N)
    fd.finite_difference_jacobian_b(J, f, _dev(np.ones(N)), fdtype, colorvec=np.zeros(N, np.int64))
    assert torch.all(J.nzval == 0)
    assert f.fcalls == (1 if fdtype == "forward" else 0)


def test_single_colour_and_invalid_colouring_semantics(oracle):
    # one colour for every column of a tridiagonal pattern is NOT a valid colouring: the reference then stores, in
    # every entry of row r, the derivative along the all-ones direction -- assignment, not accumulation
    # (ext/FiniteDiffSparseArraysExt.jl:38-47).  The device must reproduce exactly that.
    N = 64
    cp, rv = P.tridiag_csc(N)
    x = np.random.default_rng(8).random(N)
    colors = np.ones(N, np.int64)
    J = fd.SparseMatrixCSC(N, N, cp, rv, _dev(np.zeros(rv.size)))
    fd.finite_difference_jacobian_b(J, fd.BuiltinF("tridiag_nl", N), _dev(x), "forward", colorvec=colors)
    ref = oracle.jacobian("forward", oracle.Fixture("tridiag_nl", N), x, colors, kind=oracle.PAT_CSC_COMMON, colptr=cp, rowval=rv)
    eps = fd.default_relstep("forward") * np.sqrt(np.linalg.norm(x))
    assert np.all(np.abs(J.nzval.cpu().numpy() - ref["out"]) <= 1e-6 * np.abs(ref["out"]) + 16 * 2.2e-16 * 5 / eps)


def test_status_codes_and_messages():
    L = fd.lib.load()
    N = 10
    cp, rv = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, cp, rv)
    colors = P.cyclic_colors(N, 3)
    with pytest.raises(ValueError, match="DimensionMismatch"):            # src/jacobians.jl:516
        fd.make_plan(J, J, colors[:-1], "forward")
    bad = rv.copy()
    bad[4] = N + 5
    with pytest.raises(fd.lib.FdError, match="rowval"):                   # FD_ERR_SHAPE
        fd.make_plan(fd.SparseMatrixCSC(N, N, cp, bad), fd.SparseMatrixCSC(N, N, cp, bad), colors, "forward")
    badp = cp.copy()
    badp[3] = badp[4] + 1
    with pytest.raises(fd.lib.FdError, match="monotone"):
        fd.make_plan(fd.SparseMatrixCSC(N, N, badp, rv), fd.SparseMatrixCSC(N, N, badp, rv), colors, "forward")
    with pytest.raises((KeyError, ValueError)):                             # fdtype_error (src/epsilons.jl:159-167)
        fd.make_plan(J, J, colors, "hcentral")
    with pytest.raises(fd.lib.FdError, match="colour range"):
        fd.make_plan(J, J, colors, "forward", color_range=(2, 1))
    with pytest.raises(fd.lib.FdError, match="window"):
        fd.make_plan(J, J, colors, "forward", col_window=(5, 50))
    assert L.fd_last_error()  # the message of the last failure stays readable


def test_failing_user_launcher_is_reported_not_swallowed():
    N = 30
    cp, rv = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, cp, rv, _dev(np.zeros(rv.size)))

    def boom(fx, xx):
        raise ArithmeticError("f! failed beyond its bound")      # test/finitedifftests.jl:410: an f that throws

    with pytest.raises(ArithmeticError, match="beyond its bound"):
        fd.finite_difference_jacobian_b(J, fd.TorchF(boom, N, N), _dev(np.ones(N)), "forward", colorvec=P.cyclic_colors(N, 3))
    # the plan / context stay usable afterwards
    fd.finite_difference_jacobian_b(J, fd.BuiltinF("tridiag", N), _dev(np.ones(N)), "forward", colorvec=P.cyclic_colors(N, 3))
    got = J.nzval.cpu().numpy()
    assert np.allclose(got[rv == P.csc_cols(cp)], -2.0, atol=1e-6)


def test_repeated_calls_new_x_same_plan(oracle):
    # cache reuse at a new x (test/cache_reuse_tests.jl:57-62): one plan, many x, results depend only on the current x
    N = 2000
    cp, rv = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    J = fd.SparseMatrixCSC(N, N, cp, rv, _dev(np.zeros(rv.size)))
    cache = fd.JacobianCache(_dev(np.zeros(N)), "central", colorvec=colors, sparsity=J)
    f = fd.BuiltinF("tridiag_nl", N)
    outs = []
    for seed in (1, 2, 1):
        x = np.random.default_rng(seed).random(N)
        fd.finite_difference_jacobian_b(J, f, _dev(x), cache)
        outs.append(J.nzval.cpu().numpy().copy())
    assert np.array_equal(outs[0], outs[2]) and not np.array_equal(outs[0], outs[1])
    ref = oracle.jacobian("central", oracle.Fixture("tridiag_nl", N), np.random.default_rng(1).random(N), colors,
                          kind=oracle.PAT_CSC_COMMON, colptr=cp, rowval=rv)
    assert np.allclose(outs[0], ref["out"], rtol=1e-6, atol=1e-8)


def test_two_plans_from_two_threads():
    # "different plans may be used from different threads" (include/fdjac.h): each thread owns a context + plan
    N = 200_000
    cp, rv = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    results, errors = {}, []

    def work(tag, fdtype):
        try:
            torch.cuda.set_device(0)
            ctx = fd.Context(0, stream=None)      # own non-blocking stream
            J = fd.SparseMatrixCSC(N, N, cp, rv, None)
            plan = fd.make_plan(J, J, colors, fdtype, ctx=ctx)
            f = fd.BuiltinF("tridiag", N, ctx=ctx)
            out = torch.zeros(rv.size, dtype=torch.float64, device="cuda")
            x = _dev(np.random.default_rng(5).random(N))
            for _ in range(20):
                plan.jacobian(f, x, [out])
            results[tag] = out.cpu().numpy()
        except BaseException as e:   # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=work, args=(k, fdt)) for k, fdt in enumerate(["forward", "central", "complex", "forward"])]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    isdiag = rv == P.csc_cols(cp)
    for k, tol in ((0, 5e-7), (1, 1e-9), (2, 1e-14), (3, 5e-7)):
        assert np.max(np.abs(results[k][isdiag] + 2.0)) < tol and np.max(np.abs(results[k][~isdiag] - 1.0)) < tol
    assert np.array_equal(results[0], results[3])


def test_plain_c_client(tmp_path):
    # examples/c_abi_tridiag.c: the ABI used from plain C with Julia-style host arrays (what a ccall shim does)
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_tridiag")
    libdir = os.path.join(root, "finitediff.jl_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_tridiag.c"),
                           "-o", exe, "-L" + libdir, "-lfdjac", "-lm", "-Wl,-rpath," + libdir])
    for n in ("30", "100000"):
        out = subprocess.run([exe, n], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "f!_evaluations=4" in out.stdout


def _build_c_clients(root, tmp_path):
    import subprocess
    exe = str(tmp_path / "c_abi_clients")
    libdir = os.path.join(root, "finitediff.jl_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_clients.c"),
                           "-o", exe, "-L" + libdir, "-lfdjac", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


@pytest.mark.parametrize("client", ["csc", "csc_store_cols", "csc_device", "csc_dense", "coo_dense", "entries", "dense", "tridiagonal", "banded", "blockbanded", "bandedblockbanded",
                                    "csc_f32", "jvp", "solve", "bandsolve", "host", "complex_x", "complex_structured", "out_of_place", "resize", "dropin", "jit", "terms"])
def test_c_clients_every_plan_kind(tmp_path, client):
    # examples/c_abi_clients.c: one plain-C client per method of the Julia shim (finitediff.jl_amd/julia/FiniteDiffMI355X.jl)
    # -- Julia-layout arrays, DEVICE pointers for x / J's storage, the caller's own stream, fd_jacobian_async -- each
    # checking every stored value against the analytic Jacobian and the number of f! evaluations against the reference's
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = _build_c_clients(root, tmp_path)
    out = subprocess.run([exe, client], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all clients ok" in out.stdout and "FAILED" not in out.stdout


def test_c_client_links_a_row_function_given_as_bitcode(tmp_path):
    # fd_f_link_rows_bitcode end to end as a caller without C++ sees it: tests/bitcode_user_tridiag_nl.hip (the C interface only, no
    # library header) compiled OFFLINE to LLVM bitcode by hipcc, handed to the library by a plain-C process that uses /opt/rocm's own
    # runtime and hiprtc (the LLVM that wrote the bitcode reads it), linked into the library's kernels at run time: forward, central and
    # complex step x opaque f! / column store / band store -- every one the built-in family's bits
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    bc = str(tmp_path / "user.bc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fgpu-rdc", "-emit-llvm", "--offload-device-only", "-c",
                    os.path.join(root, "tests", "bitcode_user_tridiag_nl.hip"), "-o", bc], check=True, capture_output=True)
    exe = _build_c_clients(root, tmp_path)
    out = subprocess.run([exe, "bitcode", bc, "300007", "20"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "the built-in family's bits  ok" in out.stdout and "FAILED" not in out.stdout, out.stdout


def test_stage_timings_do_not_serialise_the_stream():
    # fd_plan_enable_timing: level 1 brackets the diff+decompress kernel only, level 2 every stage + the whole call;
    # the calls harvest finished spans with hipEventQuery (never a stream synchronise), results are unchanged
    N = 200_000
    cp, rv = P.tridiag_csc(N)
    pat = fd.SparseMatrixCSC(N, N, cp, rv, None)
    plan = fd.make_plan(pat, pat, P.cyclic_colors(N, 3), "forward")
    f = fd.BuiltinF("tridiag", N)
    x = torch.as_tensor(np.random.default_rng(0).random(N), device="cuda")
    ref = torch.full((rv.size,), float("nan"), dtype=torch.float64, device="cuda")
    plan.jacobian(f, x, [ref])
    out = torch.empty_like(ref)
    plan.enable_timing(1)
    for _ in range(50):
        plan.jacobian(f, x, [out], sync=False)
    tm = plan.timings()
    assert tm["decompress"]["launches"] == 50 and tm["total"]["launches"] == 0 and tm["f"]["launches"] == 0
    assert 0 < tm["decompress"]["ms_sum"] / 50 < 1.0
    plan.enable_timing(2)
    for _ in range(7):
        plan.jacobian(f, x, [out], sync=False)
    tm = plan.timings()
    assert tm["total"]["launches"] == 7 and tm["decompress"]["launches"] == 7 and tm["eps"]["launches"] == 7
    assert tm["total"]["ms_sum"] >= tm["decompress"]["ms_sum"] > 0
    plan.enable_timing(0)
    plan.jacobian(f, x, [out])
    assert plan.timings()["decompress"]["launches"] == 0 and torch.equal(out, ref)


def test_bound_call_matches_and_validates():
    # Plan.bind: arguments resolved once, one foreign call per Jacobian; same bits as Plan.jacobian
    N = 5000
    cp, rv = P.tridiag_csc(N)
    pat = fd.SparseMatrixCSC(N, N, cp, rv, None)
    plan = fd.make_plan(pat, pat, P.cyclic_colors(N, 3), "central")
    f = fd.BuiltinF("tridiag_nl", N)
    x = torch.as_tensor(np.random.default_rng(1).random(N), device="cuda")
    ref = torch.full((rv.size,), float("nan"), dtype=torch.float64, device="cuda")
    plan.jacobian(f, x, [ref])
    out = torch.full_like(ref, float("nan"))
    call = plan.bind(f, x, [out])
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and f.fcalls == 4 * 6
    with pytest.raises(ValueError):
        plan.bind(f, x.cpu().numpy(), [out])
    with pytest.raises(TypeError):
        plan.bind(fd.BuiltinF("tridiag_nl", N, dtype=np.float32), x, [out])


@pytest.mark.parametrize("fdtype", ["forward", "central"])
def test_complex_valued_x_reference_known_answer_dense_arm(fdtype):
    # returntype <: Complex with forward / central differences (src/jacobians.jl:94-128, 537-622), the reference's own fixture
    # test/finitedifftests.jl:480-513 (iipf : C^2 -> C^2, analytic J_ref, err < 1e-4 / 1e-8): dense arm on the device through
    # FD_PLAN_COMPLEX_X, a user f! written in torch on complex tensors; also against the numpy restatement, and f_in
    from oracle import np_oracle as O
    rng = np.random.default_rng(480)
    xh = rng.random(2) + 1j * rng.random(2)
    x = torch.as_tensor(xh, device="cuda")

    def iipf_t(fv, xx):
        fv[0] = (1j * xx[0] + 3) * (xx[1] ** 3 - 7) + 18
        fv[1] = torch.sin(xx[1] * torch.exp(xx[0]) - 1)

    def iipf_n(fv, xx):
        fv[0] = (1j * xx[0] + 3) * (xx[1] ** 3 - 7) + 18
        fv[1] = np.sin(xx[1] * np.exp(xx[0]) - 1)

    J_ref = np.array([[1j * (-7 + xh[1] ** 3), 3 * (3 + 1j * xh[0]) * xh[1] ** 2],
                      [np.exp(xh[0]) * xh[1] * np.cos(1 - np.exp(xh[0]) * xh[1]), np.exp(xh[0]) * np.cos(1 - np.exp(xh[0]) * xh[1])]])
    tol = 1e-4 if fdtype == "forward" else 1e-8
    f = fd.TorchF(iipf_t, 2, 2)
    cache = fd.JacobianCache(x, fdtype, np.complex128)
    J = torch.full((2, 2), complex(float("nan"), float("nan")), dtype=torch.complex128, device="cuda").t()      # column-major
    fd.finite_difference_jacobian_b(J, f, x, cache)
    got = J.cpu().numpy()
    assert np.abs(got - J_ref).max() < tol
    assert f.fcalls == (3 if fdtype == "forward" else 4)
    Jo, _n = O.jacobian_complex_x(iipf_n, xh, np.arange(1, 3), None, fdtype)
    assert np.abs(got - Jo).max() <= 1e-6 * np.abs(Jo).max() + 1e-7
    if fdtype == "forward":          # f_in: one evaluation fewer, same values to rounding
        y = torch.zeros(2, dtype=torch.complex128, device="cuda")
        iipf_t(y, x)
        J2 = torch.zeros((2, 2), dtype=torch.complex128, device="cuda").t()
        f2 = fd.TorchF(iipf_t, 2, 2)
        fd.finite_difference_jacobian_b(J2, f2, x, cache, f_in=y)
        assert f2.fcalls == 2 and np.abs(J2.cpu().numpy() - J_ref).max() < tol
    with pytest.raises(ValueError):     # Val(:complex) with a complex returntype: fdtype_error, as the reference (src/jacobians.jl:106)
        fd.JacobianCache(_dev(np.ones(12)), "complex", np.complex128)


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("kind", ["csc", "dense_sparse", "dense_dense", "builtin_csc"])
def test_complex_valued_x_coloured_paths(fdtype, kind):
    # the coloured sparse arm with complex x: the masked norm is over complex elements, the step real, the real parts are
    # perturbed, J is complex -- CSC nzval (Complex nzval = (re, im) pairs in storage order), dense J with a CSC / dense-matrix
    # pattern, and the built-in tridiagonal fixture evaluated on complex points (is_complex = 1), against the numpy restatement
    from oracle import np_oracle as O
    N = 40_000 if kind == "builtin_csc" else 300
    rng = np.random.default_rng(77)
    xh = rng.random(N) + 1j * (rng.random(N) - 0.5)
    x = torch.as_tensor(xh, device="cuda")
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)

    def f_t(fv, xx):
        z = torch.zeros(1, dtype=xx.dtype, device=xx.device)
        xm, xp = torch.cat([z, xx[:-1]]), torch.cat([xx[1:], z])
        fv.copy_((xm - 2 * xx) + xp + (xx * xx) * xp)

    def f_n(fv, xx):
        xm, xp = np.concatenate([[0], xx[:-1]]), np.concatenate([xx[1:], [0]])
        fv[:] = (xm - 2 * xx) + xp + (xx * xx) * xp

    f = fd.BuiltinF("tridiag_nl", N) if kind == "builtin_csc" else fd.TorchF(f_t, N, N)
    xp_ = np.concatenate([xh[1:], [0]])
    want_dense = None
    if N <= 1000:
        pat = np.abs(np.subtract.outer(np.arange(N), np.arange(N))) <= 1
        want_dense, ncalls = O.jacobian_complex_x(f_n, xh, colors, pat, fdtype)
    tol = 2e-6 if fdtype == "forward" else 2e-9
    if kind in ("csc", "builtin_csc"):
        Jm = fd.SparseMatrixCSC(N, N, colptr, rowval, torch.full((rowval.size,), complex(float("nan"), float("nan")), dtype=torch.complex128, device="cuda"))
        fd.finite_difference_jacobian_b(Jm, f, x, fdtype, np.complex128, colorvec=colors, sparsity=Jm)
        got = Jm.nzval.cpu().numpy()
        col = (np.arange(rowval.size) + 1) // 3
        row = rowval - 1
        analytic = np.where(row == col, -2 + 2 * xh[col] * xp_[col], np.where(row + 1 == col, 1 + xh[row] ** 2, 1.0 + 0j))
        assert np.abs(got - analytic).max() < tol * 4
        if want_dense is not None:
            assert np.abs(got - want_dense[row, col]).max() <= 1e-6 * np.abs(want_dense).max() + 1e-7
    else:
        Jd = torch.full((N, N), complex(float("nan"), float("nan")), dtype=torch.complex128, device="cuda").t()
        sp = fd.SparseMatrixCSC(N, N, colptr, rowval) if kind == "dense_sparse" else pat.astype(float)
        fd.finite_difference_jacobian_b(Jd, f, x, fdtype, np.complex128, colorvec=colors, sparsity=sp)
        got = Jd.cpu().numpy()
        assert np.all(got[~pat] == 0)                                   # fill_matrix!: entries outside the pattern are zero
        assert np.abs(got - want_dense).max() <= 1e-6 * np.abs(want_dense).max() + 1e-7
    assert f.fcalls == (4 if fdtype == "forward" else 6)


@pytest.mark.parametrize("fdtype", ["forward", "central"])
def test_complex_valued_x_structured_storage(fdtype):
    # complex-valued x with J::Tridiagonal / BandedMatrix / BlockBandedMatrix (the reference's loop is generic in the matrix type:
    # src/jacobians.jl:94-128, 537-622 with src/iteration_utils.jl:25-32, ext/FiniteDiffBandedMatricesExt.jl:13-27,
    # ext/FiniteDiffBlockBandedMatricesExt.jl:44-68).  Every stored entry must be the value the dense-J complex path (checked
    # against the numpy restatement in test_complex_valued_x_coloured_paths) puts at that (row, column): same arithmetic, same bits.
    cnan = complex(float("nan"), float("nan"))

    def cfull(*shape):
        return torch.full(shape, cnan, dtype=torch.complex128, device="cuda")

    # --- Tridiagonal and BandedMatrix(1, 1): the tridiagonal fixture
    N = 257
    rng = np.random.default_rng(91)
    xh = rng.random(N) + 1j * (rng.random(N) - 0.5)
    x = torch.as_tensor(xh, device="cuda")
    colors = P.cyclic_colors(N, 3)

    def f_t(fv, xx):
        z = torch.zeros(1, dtype=xx.dtype, device=xx.device)
        xm, xp = torch.cat([z, xx[:-1]]), torch.cat([xx[1:], z])
        fv.copy_((xm - 2 * xx) + xp + (xx * xx) * xp)

    f = fd.TorchF(f_t, N, N)
    pat = (np.abs(np.subtract.outer(np.arange(N), np.arange(N))) <= 1).astype(float)
    Jd = cfull(N, N).t()
    fd.finite_difference_jacobian_b(Jd, f, x, fdtype, np.complex128, colorvec=colors, sparsity=pat)
    ref = Jd.cpu().numpy()
    tri = fd.Tridiagonal(cfull(N - 1), cfull(N), cfull(N - 1))
    n0 = f.fcalls
    fd.finite_difference_jacobian_b(tri, f, x, fdtype, np.complex128, colorvec=colors)
    assert f.fcalls - n0 == (4 if fdtype == "forward" else 6)
    assert np.array_equal(tri.d.cpu().numpy(), np.diag(ref)) and np.array_equal(tri.dl.cpu().numpy(), np.diag(ref, -1))
    assert np.array_equal(tri.du.cpu().numpy(), np.diag(ref, 1))
    band = fd.BandedMatrix(cfull(N, 3).t(), N, 1, 1)                     # (l + u + 1) x N, column-major
    fd.finite_difference_jacobian_b(band, f, x, fdtype, np.complex128, colorvec=colors)
    data = band.data.cpu().numpy()
    for j in range(N):
        for r in range(max(j - 1, 0), min(j + 1, N - 1) + 1):
            assert data[1 + r - j, j] == ref[r, j]
    assert data[0, 0] == 0 and data[2, N - 1] == 0                      # the slots of rows outside the matrix hold 0
    # --- BlockBandedMatrix: dense blocks, block-tridiagonal, f = B x + x .* x with a dense block-tridiagonal B
    lay = P.BlockBandedLayout(np.array([3, 5, 4, 2, 6]), 1, 1)
    Nb = lay.N
    colors_b = lay.colors()
    off = np.concatenate([[0], np.cumsum(lay.blk_sizes)])
    mask = np.zeros((Nb, Nb), bool)
    for K in range(lay.nblk):
        for Jb in range(max(K - 1, 0), min(K + 1, lay.nblk - 1) + 1):
            mask[off[K]:off[K + 1], off[Jb]:off[Jb + 1]] = True
    B = (rng.random((Nb, Nb)) + 1j * rng.random((Nb, Nb))) * mask
    Bt = torch.as_tensor(B, device="cuda")
    xb = torch.as_tensor(rng.random(Nb) + 1j * (rng.random(Nb) - 0.5), device="cuda")
    fb = fd.TorchF(lambda fv, xx: fv.copy_(Bt @ xx + xx * xx), Nb, Nb)
    Jdb = cfull(Nb, Nb).t()
    fd.finite_difference_jacobian_b(Jdb, fb, xb, fdtype, np.complex128, colorvec=colors_b, sparsity=mask.astype(float))
    refb = Jdb.cpu().numpy()
    Jbb = fd.BlockBandedMatrix(cfull(lay.data_len), lay)
    fd.finite_difference_jacobian_b(Jbb, fb, xb, fdtype, np.complex128, colorvec=colors_b)
    got = lay.to_dense(Jbb.data.cpu().numpy()) if hasattr(lay, "to_dense") else None
    if got is not None:
        assert np.array_equal(got[mask], refb[mask])
    assert np.abs(refb[mask] - (B + np.diag(2 * xb.cpu().numpy()))[mask]).max() < (2e-6 if fdtype == "forward" else 2e-9) * 4


def test_hip_error_left_behind_by_a_launcher_is_reported():
    # A launcher that provokes a HIP error and does NOT report it (returns 0): the library's launch checks pick the error up
    # and the call returns FD_ERR_HIP instead of FD_OK.  (A genuine device fault -- an out-of-bounds write inside a user
    # kernel -- is fatal to the process on this platform and cannot be turned into a status code by anyone.)
    import ctypes as C
    N = 5000
    cp, rv = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, cp, rv)
    plan = fd.make_plan(J, J, P.cyclic_colors(N, 3), "forward")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    good = fd.BuiltinF("tridiag", N)

    class Sloppy:
        dtype = np.float64
        fctx = good.fctx
        error = None

        def __init__(self):
            def launch(fctx, fx, x, nbatch, xs, fs, r0, r1, is_complex, stream):
                rc = good.fn(fctx, fx, x, nbatch, xs, fs, r0, r1, is_complex, stream)
                hip.hipMemsetAsync(None, 0, 1 << 40, stream)     # invalid: leaves hipErrorInvalidValue behind
                return rc                                         # ... and does not tell anybody
            self.fn = fd.lib.F_LAUNCH(launch)

    out = _dev(np.full(plan.out_len(0), np.nan))
    with pytest.raises(fd.lib.FdError) as e:
        plan.jacobian(Sloppy(), _dev(np.ones(N)), [out])
    assert e.value.code == 4 and "invalid" in str(e.value).lower()          # FD_ERR_HIP with the HIP error text
    plan.ctx.synchronize()
    plan.jacobian(good, _dev(np.ones(N)), [out])                            # the plan / context stay usable
    assert not torch.isnan(out).any()


def test_user_kernel_stores_the_jacobian_through_the_device_header(tmp_path, oracle):
    # examples/user_f_store.hip: a USER's own HIP f! -- its own shared library, compiled with hipcc apart from libfdjac against
    # include/fdjac.h + include/fdjac_device.h only -- registered with FD_LAZY_CAP_STORE stores the tridiagonal Jacobian of a
    # nonlinear residual itself (fd_band_emit per entry, and fd_band_emit_wave: the dense-store path of the built-in launcher)
    # into CSC nzval, BandedMatrix data and Tridiagonal diagonals.  examples/user_store_client.c (plain C) checks analytic
    # values, bit-identity with the library-decompressed run and the f! evaluation counts; here the CSC values are also
    # compared with the CPU oracle evaluating the same residual (src/jacobians.jl:504-586 + ext/FiniteDiffSparseArraysExt.jl:38-47).
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc, libdir = os.path.join(root, "include"), os.path.join(root, "finitediff.jl_amd", "lib")
    user_so = str(tmp_path / "libuser_f.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-I" + inc,
                           os.path.join(root, "examples", "user_f_store.hip"), "-o", user_so])
    deps = subprocess.run(["ldd", user_so], capture_output=True, text=True).stdout
    assert "libfdjac" not in deps                      # the user's library does not link the product library
    exe = str(tmp_path / "user_store_client")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + inc, os.path.join(root, "examples", "user_store_client.c"), "-o", exe,
                           "-L" + str(tmp_path), "-luser_f", "-L" + libdir, "-lfdjac", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + str(tmp_path), "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    N = 20_011
    dump = str(tmp_path / "dump.bin")
    out = subprocess.run([exe, str(N), dump], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "user_store_client ok" in out.stdout and "FAILED" not in out.stdout
    raw = np.fromfile(dump, dtype=np.float64)
    assert int(np.frombuffer(raw[:1].tobytes(), dtype=np.int64)[0]) == N
    x, got = raw[1:1 + N], raw[1 + N:]
    colptr, rowval = P.tridiag_csc(N)

    def f_py(fx, xx):
        xm = np.concatenate([[0.0], xx[:-1]])
        xp = np.concatenate([xx[1:], [0.0]])
        fx[:] = ((xm - 2.0 * xx) + xp) + (0.25 * xx) * (xp - xm)

    ref = oracle.jacobian("forward", oracle.PyF(f_py, N, N), x, P.cyclic_colors(N, 3), kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    assert got.size == ref["out"].size
    colors = P.cyclic_colors(N, 3)
    eps_min = min(max(1.4901161193847656e-8 * np.sqrt(np.linalg.norm(x * (colors == c))), 1.4901161193847656e-8) for c in (1, 2, 3))
    atol = 16 * 2.220446049250313e-16 * 4.0 / eps_min
    assert np.all(np.abs(got - ref["out"]) <= 1e-6 * np.abs(ref["out"]) + atol)


@pytest.mark.parametrize("shape", [(150, 12), (7, 33), (2, 5)])
def test_user_kernel_stores_a_blockbanded_jacobian_through_the_device_header(tmp_path, shape):
    # examples/user_bb_store.hip: a USER's own HIP f! with a block-tridiagonal Jacobian of dense blocks -- its own shared library,
    # compiled apart from libfdjac -- registered with FD_LAZY_CAP_STORE receives a fd_colrange_store (the plan verified the
    # colouring) and stores imag(f) / eps into BlockBandedMatrix data with fd_colrange_emit (complex step, src/jacobians.jl:624-637,
    # ext/FiniteDiffBlockBandedMatricesExt.jl:44-68).  examples/user_bb_client.c checks the analytic values, agreement with the
    # plan driven through the user's plain launcher + the library's decompression, and the f! evaluation counts.
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc, libdir = os.path.join(root, "include"), os.path.join(root, "finitediff.jl_amd", "lib")
    user_so = str(tmp_path / "libuser_bb.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-I" + inc,
                           os.path.join(root, "examples", "user_bb_store.hip"), "-o", user_so])
    assert "libfdjac" not in subprocess.run(["ldd", user_so], capture_output=True, text=True).stdout
    exe = str(tmp_path / "user_bb_client")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + inc, os.path.join(root, "examples", "user_bb_client.c"), "-o", exe,
                           "-L" + str(tmp_path), "-luser_bb", "-L" + libdir, "-lfdjac", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + str(tmp_path), "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(shape[0]), str(shape[1])], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "user_bb_client ok" in out.stdout and "FAILED" not in out.stdout


@pytest.mark.parametrize("kind", ["csc_store", "csc_handover", "blockbanded_complex"])
def test_jacobian_call_captures_into_a_hip_graph(monkeypatch, kind):
    # fd_jacobian_async enqueues kernels only -- no allocation, no synchronisation, no host read-back once the plan has run once --
    # so a time-stepping loop may capture it into a HIP graph: the replay gives the bits of the stream launches, also for new
    # contents of x (scripts/graph_probe.py has the timings: replay is no faster than three stream launches on this runtime)
    monkeypatch.setenv("FDJAC_LAZY_STORE", "0" if kind == "csc_handover" else "1")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx = fd.Context(0)
        if kind == "blockbanded_complex":
            nb, bs = 24, 16
            N = nb * bs
            lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
            J = fd.BlockBandedMatrix(None, lay)
            plan = fd.make_plan(J, J, lay.colors(), "complex", ctx=ctx)
            f = fd.BuiltinF("blockcoupled", nb, bs, ctx=ctx)
        else:
            N = 30_011
            cp, rv = P.tridiag_csc(N)
            J = fd.SparseMatrixCSC(N, N, cp, rv)
            plan = fd.make_plan(J, J, P.cyclic_colors(N, 3), "forward", ctx=ctx)
            f = fd.BuiltinF("tridiag_nl", N, ctx=ctx)
        plan.set_lazy(f)
        x = torch.rand(N, dtype=torch.float64, device="cuda")
        out = torch.full((plan.out_len(0),), float("nan"), dtype=torch.float64, device="cuda")
        enq = plan.bind(f, x, [out])
        enq()
        torch.cuda.synchronize()
        ref = out.clone()
        out.fill_(float("nan"))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            enq()
        torch.cuda.synchronize()
        for scale in (1.0, 0.5, 3.0):
            x.mul_(scale)
            out.fill_(float("nan"))
            g.replay()
            torch.cuda.synchronize()
            got = out.clone()
            out.fill_(float("nan"))
            enq()
            torch.cuda.synchronize()
            assert not torch.isnan(got).any() and torch.equal(got.view(torch.int64), out.view(torch.int64))
            if scale == 1.0:
                assert torch.equal(got.view(torch.int64), ref.view(torch.int64))


# ---- the drop-in call's cache -> plan lookup (round 4): identity key + invalidate, or the library's content check -------------
def _tridiag_dropin(N, device_pattern=False):
    cp, rv = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    nz = _dev(np.full(rv.size, np.nan))
    if device_pattern:
        J = fd.DevicePatternCSC(N, N, torch.as_tensor(cp.astype(np.int32), device="cuda"), torch.as_tensor(rv.astype(np.int32), device="cuda"), nz)
        colors = torch.as_tensor(colors.astype(np.int32), device="cuda")
    else:
        J = fd.SparseMatrixCSC(N, N, cp, rv, nz)
    return J, colors, cp


@pytest.mark.parametrize("device_pattern", [False, True])
@pytest.mark.parametrize("check", ["identity", "content"])
def test_dropin_plan_lookup_sees_an_in_place_edit_of_colorvec(check, device_pattern):
    # The reference re-reads colorvec on every call (src/jacobians.jl:512): removing a column's colour IN PLACE zeroes its stored
    # values on the next call.  The drop-in lookup is O(1) on identities: with pattern_check = "identity" the compiled plan stands
    # until cache.invalidate(); with "content" the library compares the arrays with the plan's fingerprints (fd_plan_matches) and
    # the edit takes effect by itself -- on host arrays (host threads) and on device arrays (kernels) alike.
    N = 20011
    J, colors, cp = _tridiag_dropin(N, device_pattern)
    x = _dev(np.random.default_rng(3).random(N))
    f = fd.BuiltinF("tridiag_nl", N)
    cache = fd.JacobianCache(x, "forward", colorvec=colors, sparsity=J)
    cache.pattern_check = check
    fd.finite_difference_jacobian_b(J, f, x, cache)
    p0 = cache.last_plan
    fd.finite_difference_jacobian_b(J, f, x, cache)
    assert cache.last_plan is p0                                   # the second call found the plan
    first = J.nzval.clone()
    jz = N // 2
    d = int(cp[jz] - 1 + 1)                                        # the diagonal entry of column jz
    assert float(first[d]) != 0.0
    colors[jz] = 0                                                 # in-place edit (numpy array / CUDA tensor)
    fd.finite_difference_jacobian_b(J, f, x, cache)
    if check == "content":
        assert cache.last_plan is not p0 and float(J.nzval[d]) == 0.0
    else:
        assert cache.last_plan is p0 and torch.equal(J.nzval, first)   # the snapshot
        cache.invalidate()
        fd.finite_difference_jacobian_b(J, f, x, cache)
        assert cache.last_plan is not p0 and float(J.nzval[d]) == 0.0
    # every other stored value is what it was, up to the step size of column jz's colour (its masked norm lost one element)
    keep = torch.ones_like(first, dtype=torch.bool)
    keep[int(cp[jz] - 1):int(cp[jz + 1] - 1)] = False
    assert torch.allclose(J.nzval[keep], first[keep], rtol=1e-6, atol=1e-7)
    # a NEW colour array (same content as the edited one) is a new identity: new plan without being told
    p1 = cache.last_plan
    colors2 = colors.clone() if torch.is_tensor(colors) else colors.copy()
    fd.finite_difference_jacobian_b(J, f, x, cache, colorvec=colors2)
    assert cache.last_plan is not p1 and float(J.nzval[d]) == 0.0


def test_dropin_deferred_content_check_reports_an_edit_one_call_late():
    # pattern_check = "content_async" (fd_plan_matches_async): ONE fused kernel ahead of every Jacobian compares the device arrays with
    # the plan's fingerprints and raises a sticky status on the device -- nothing is copied back, the stream is never stopped.  The
    # verdict is deferred: the call right after an in-place edit still runs on the old plan; Context.synchronize() reports FD_ERR_STALE,
    # Plan.stale() turns true, and the next call recompiles and computes with the edited colours.
    N = 20011
    J, colors, cp = _tridiag_dropin(N, True)
    x = _dev(np.random.default_rng(3).random(N))
    f = fd.BuiltinF("tridiag_nl", N)
    cache = fd.JacobianCache(x, "forward", colorvec=colors, sparsity=J)
    cache.pattern_check = "content_async"
    ctx = fd.Context.default()
    for _ in range(3):
        fd.finite_difference_jacobian_b(J, f, x, cache)
    ctx.synchronize()
    p0 = cache.last_plan
    assert not p0.stale()
    first = J.nzval.clone()
    jz = N // 2
    d = int(cp[jz] - 1 + 1)
    colors[jz] = 0                                                 # in-place edit of the CUDA tensor
    fd.finite_difference_jacobian_b(J, f, x, cache)                # check enqueued, then the Jacobian of the OLD plan
    assert cache.last_plan is p0
    with pytest.raises(fd.lib.FdError) as e:
        ctx.synchronize()                                          # the deferred verdict
    assert e.value.code == 9 and p0.stale()                        # FD_ERR_STALE
    assert torch.equal(J.nzval, first)                             # (what the stale plan computed)
    with pytest.raises(fd.lib.FdError) as e2:                      # the stale plan refuses further calls
        p0.jacobian(f, x, [J.nzval])
    assert e2.value.code == 9
    fd.finite_difference_jacobian_b(J, f, x, cache)                # the lookup sees the verdict: new plan, edited colours
    ctx.synchronize()
    assert cache.last_plan is not p0 and float(J.nzval[d]) == 0.0 and not cache.last_plan.stale()
    # unedited arrays: many calls, never stale; and a pattern edit (rowval) is seen as well
    for _ in range(20):
        fd.finite_difference_jacobian_b(J, f, x, cache)
    ctx.synchronize()
    p1 = cache.last_plan
    assert not p1.stale()
    J.rowval[5] = J.rowval[5]                                      # a write that changes nothing
    fd.finite_difference_jacobian_b(J, f, x, cache)
    ctx.synchronize()
    assert cache.last_plan is p1 and not p1.stale()


@pytest.mark.parametrize("seed", range(12))
def test_deferred_and_blocking_content_checks_agree_randomised(seed):
    # fd_plan_matches_async (one fused kernel, per-workgroup slots, group tickets) against fd_plan_matches (three kernels + copy) and the
    # truth: random sizes, index widths, column windows, and ONE edited element at a random place of a random array (or none)
    rng = np.random.default_rng(3100 + seed)
    N = int(rng.choice([7, 300, 5000, 70001, 400003]))
    per_col = int(rng.integers(1, 6))
    rows = np.sort(np.minimum(np.arange(N)[:, None] + rng.integers(0, 9, size=(N, per_col)), N - 1), axis=1)
    keep = np.ones_like(rows, dtype=bool)
    keep[:, 1:] = rows[:, 1:] != rows[:, :-1]
    cnt = keep.sum(axis=1)
    cp = np.empty(N + 1, np.int64)
    cp[0] = 1
    np.cumsum(cnt, out=cp[1:])
    cp[1:] += 1
    rv = (rows[keep] + 1).astype(np.int64)
    pat = fd.SparseMatrixCSC(N, N, cp, rv, None)
    colors = fd.matrix_colors(pat)
    kw = {}
    if rng.random() < 0.4 and N > 20:
        a = int(rng.integers(0, N // 2))
        kw["col_window"] = (a, int(rng.integers(a + 1, N + 1)))
    plan = fd.make_plan(pat, pat, colors, "forward", fingerprint=True, **kw)
    it = torch.int32 if rng.random() < 0.5 else torch.int64
    dcp, drv = torch.as_tensor(cp, device="cuda").to(it), torch.as_tensor(rv, device="cuda").to(it)
    dcv = torch.as_tensor(np.asarray(colors), device="cuda").to(torch.int32 if rng.random() < 0.5 else torch.int64)
    ctx = fd.Context.default()
    assert plan.matches(dcp, drv, dcv)
    plan.matches(dcp, drv, dcv, deferred=True)
    ctx.synchronize()
    assert not plan.stale()
    which = int(rng.integers(0, 4))                                 # 0: nothing edited
    c0, c1 = kw.get("col_window", (0, N))
    inside = True
    if which == 1:
        j = int(rng.integers(0, N))
        dcv[j] = dcv[j] % int(colors.max()) + 1 if int(colors.max()) > 1 else 0
    elif which == 2:
        q = int(rng.integers(0, rv.size))
        drv[q] = drv[q] % N + 1 if N > 1 else 1
        inside = int(cp[c0]) - 1 <= q < int(cp[c1]) - 1             # (a window plan compares its own slice of rowval)
        if N == 1:
            which = 0
    elif which == 3 and N > 1:
        j = int(rng.integers(c0, c1 + 1))
        dcp[j] = dcp[j] + 1
    elif which == 3:
        which = 0
    want_same = which == 0 or not inside
    assert bool(plan.matches(dcp, drv, dcv)) == want_same
    plan.matches(dcp, drv, dcv, deferred=True)
    if want_same:
        ctx.synchronize()
        assert not plan.stale()
    else:
        with pytest.raises(fd.lib.FdError) as e:
            ctx.synchronize()
        assert e.value.code == 9 and plan.stale()


def test_plan_matches_compares_content_not_identity():
    # fd_plan_matches on host arrays, device arrays, either index width; column windows compare their own slice only
    N = 50021
    cp, rv = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    pat = fd.SparseMatrixCSC(N, N, cp, rv, None)
    plan = fd.make_plan(pat, pat, colors, "forward", fingerprint=True)
    assert plan.matches(cp, rv, colors)
    assert plan.matches(cp.copy(), rv.copy(), colors.copy())                    # other objects, same content
    assert plan.matches(cp.astype(np.int32), rv.astype(np.int32), colors.astype(np.int32))     # other width, same values
    dcp, drv, dcv = (torch.as_tensor(a.astype(np.int32), device="cuda") for a in (cp, rv, colors))
    assert plan.matches(dcp, drv, dcv)                                          # the same content on the device (kernels)
    assert plan.matches(colorvec=colors) and plan.matches(cp, rv)               # members left out are not compared
    c2 = colors.copy(); c2[N // 3] = 1 + c2[N // 3] % 3
    assert not plan.matches(cp, rv, c2) and not plan.matches(colorvec=c2)
    r2 = rv.copy(); r2[[7, 8]] = r2[[8, 7]]
    assert not plan.matches(cp, r2, colors)
    drv2 = drv.clone(); drv2[1000] += 1
    assert not plan.matches(dcp, drv2, dcv)
    assert not plan.matches(cp[:-1], rv, colors)                                # another length is a mismatch, nothing is read
    with pytest.raises(fd.lib.FdError):
        fd.make_plan(pat, pat, colors, "forward").matches(cp, rv, colors)       # no FD_PLAN_FINGERPRINT
    # a column window fingerprints its own slice of colptr / rowval
    w = fd.make_plan(pat, pat, colors, "forward", col_window=(1000, 30000), fingerprint=True)
    r3 = rv.copy(); r3[5] += 1                                                  # outside the window's entries
    assert w.matches(cp, r3, colors)
    r3[int(cp[2000])] += 1                                                      # inside
    assert not w.matches(cp, r3, colors)
    # the device-built plan of a device-resident pattern
    pd = fd.make_plan_csc_device(N, N, dcp, drv, dcv, "forward", fingerprint=True)
    assert pd.matches(dcp, drv, dcv) and pd.matches(cp, rv, colors) and not pd.matches(dcp, drv2, dcv)


def test_plan_matches_structural_and_list_plans():
    N = 3001
    colors = P.cyclic_colors(N, 3)
    tri = fd.Tridiagonal(_dev(np.zeros(N - 1)), _dev(np.zeros(N)), _dev(np.zeros(N - 1)))
    pt = fd.make_plan(tri, tri, colors, "central", fingerprint=True)
    c2 = colors.copy(); c2[5] = 0
    assert pt.matches(colorvec=colors) and not pt.matches(colorvec=c2)
    assert pt.matches(np.arange(4), np.arange(4), colors)                       # index arrays are ignored by structural plans
    # dense J with a dense-matrix pattern: the (I, J) lists of _findstructralnz
    A = (np.random.default_rng(1).random((40, 30)) < 0.2).astype(float)
    cv = np.arange(1, 31, dtype=np.int64)
    Jd = torch.zeros((30, 40), dtype=torch.float64, device="cuda").t()
    pl = fd.make_plan(Jd, A, cv, "forward", fingerprint=True)
    cols, rows = np.nonzero(A.T)
    assert pl.matches(rows + 1, cols + 1, cv)
    assert not pl.matches(rows + 1, np.roll(cols + 1, 1), cv)
    # Float32 instantiation and the lowered complex-valued x (fingerprints in the CALLER's units)
    cp, rv = P.tridiag_csc(N)
    pat = fd.SparseMatrixCSC(N, N, cp, rv, None)
    p32 = fd.make_plan(pat, pat, colors, "forward", dtype=np.float32, fingerprint=True)
    assert p32.matches(cp, rv, colors) and not p32.matches(cp, rv, c2)
    pcx = fd.make_plan(pat, pat, colors, "forward", complex_x=True, fingerprint=True)
    assert pcx.matches(cp, rv, colors) and not pcx.matches(cp, rv, c2)


def test_unknown_plan_flags_are_rejected():
    # fd_plan_opts.flags: only the documented bits (round-3 advisor: an internal marker lived in this field, unvalidated)
    import ctypes as C
    N = 64
    cp, rv = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    ctx = fd.Context.default()
    for bad in (1 << 16, 64, 1 << 30):
        o = fd.lib.PlanOpts()
        o.fdtype = 0
        o.flags = bad
        h = C.c_void_p()
        rc = ctx.L.fd_plan_create_csc(ctx.handle, N, N, cp.ctypes.data, rv.ctypes.data, 8, 1, colors.ctypes.data, 8, C.byref(o), C.byref(h))
        assert rc == 1 and b"flags" in ctx.L.fd_last_error() and not h.value       # FD_ERR_ARG
        rc = ctx.L.fd_plan_create_tridiagonal(ctx.handle, N, colors.ctypes.data, 8, C.byref(o), C.byref(h))
        assert rc == 1 and not h.value
        o.flags = bad | fd.lib.PLAN_COMPLEX_X
        rc = ctx.L.fd_plan_create_csc(ctx.handle, N, N, cp.ctypes.data, rv.ctypes.data, 8, 1, colors.ctypes.data, 8, C.byref(o), C.byref(h))
        assert rc == 1 and not h.value


@pytest.mark.parametrize("fdtype", FDTYPES)
def test_dropin_installs_the_lazy_launcher_like_the_shim(fdtype):
    # install_lazy!(plan, f) of the shim: the drop-in call of a built-in family runs the lazy / storing launches; cache.lazy = False
    # keeps the materialised points.  Same bits, same number of f! evaluations.
    N = 40009
    res = []
    for lazy in (True, False):
        J, colors, _cp = _tridiag_dropin(N)
        x = _dev(np.random.default_rng(8).random(N))
        f = fd.BuiltinF("tridiag_nl", N)
        cache = fd.JacobianCache(x, fdtype, colorvec=colors, sparsity=J)
        cache.lazy = lazy
        fd.finite_difference_jacobian_b(J, f, x, cache)
        res.append((J.nzval.clone(), f.fcalls, cache.last_plan.info(fd.lib.INFO_LAZY_STORE)))
    assert torch.equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    assert res[0][2] == (0 if fdtype == "complex" else 1) and res[1][2] == 0


def test_dropin_default_checks_device_arrays_beside_the_call_and_tracked_host_arrays_by_generation():
    # round 6 defaults (pattern_check "auto"): DEVICE pattern / colours -> the fused fingerprint kernel on the context's side stream,
    # verdict one call late; HOST arrays in Tracked holders -> generations (no re-hash until an edit); plain host arrays -> content.
    N = 60_000
    cp, rv = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    x = torch.as_tensor(np.random.default_rng(3).random(N), device="cuda")
    f = fd.BuiltinF("tridiag_nl", N)
    # ---- device arrays ----
    cpd, rvd = torch.as_tensor(cp.astype(np.int32), device="cuda"), torch.as_tensor(rv.astype(np.int32), device="cuda")
    cvd = torch.as_tensor(colors.astype(np.int32), device="cuda")
    Jd = fd.DevicePatternCSC(N, N, cpd, rvd, torch.full((rv.size,), float("nan"), dtype=torch.float64, device="cuda"))
    cache = fd.JacobianCache(x, "forward", colorvec=cvd, sparsity=Jd)
    assert cache.pattern_check == "auto"
    for _ in range(3):
        fd.finite_difference_jacobian_b(Jd, f, x, cache)
    torch.cuda.synchronize()
    ref = Jd.nzval.clone()
    plan0 = cache.last_plan
    # an in-place edit of the colours (a valid recolouring: shifted by one): found by the check that runs beside the NEXT call, acted on
    # by the call after it -- a new plan, the Jacobian of the new colours
    cvd.copy_(torch.as_tensor(((np.arange(N) + 1) % 3 + 1).astype(np.int32), device="cuda"))
    fd.finite_difference_jacobian_b(Jd, f, x, cache)          # (still the old plan; its check sees the edit)
    cache.last_plan.ctx.synchronize() if False else torch.cuda.synchronize()
    import time
    t0 = time.time()
    while not plan0.stale() and time.time() - t0 < 5:
        time.sleep(0.01)
    assert plan0.stale()
    with pytest.raises(fd.lib.FdError):          # the context's own sticky word: reported (and cleared) by the next fd_ctx_synchronize
        f.ctx.synchronize()
    f.ctx.synchronize()
    fd.finite_difference_jacobian_b(Jd, f, x, cache)
    torch.cuda.synchronize()
    assert cache.last_plan is not plan0
    assert torch.allclose(Jd.nzval, ref, rtol=1e-5, atol=1e-6)      # (a relabelling of the same three column sets: the same Jacobian)
    # ---- tracked host arrays ----
    A = fd.TrackedCSC(N, N, cp, rv, torch.full((rv.size,), float("nan"), dtype=torch.float64, device="cuda"))
    cv = fd.TrackedVector(colors)
    cache2 = fd.JacobianCache(x, "forward", colorvec=cv, sparsity=A)
    fd.finite_difference_jacobian_b(A, f, x, cache2)
    p1 = cache2.last_plan
    calls = {"n": 0}
    real = fd.JacobianCache._content_matches

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    fd.JacobianCache._content_matches = staticmethod(counting)
    try:
        for _ in range(5):
            fd.finite_difference_jacobian_b(A, f, x, cache2)
        assert calls["n"] == 0 and cache2.last_plan is p1            # unedited: generations only
        with cv.edit() as a:
            a[:] = (np.arange(N) + 1) % 3 + 1
        fd.finite_difference_jacobian_b(A, f, x, cache2)
        assert calls["n"] == 1 and cache2.last_plan is not p1        # edited: the content was compared, the plan rebuilt
        fd.finite_difference_jacobian_b(A, f, x, cache2)
        assert calls["n"] == 1
    finally:
        fd.JacobianCache._content_matches = staticmethod(real)
    torch.cuda.synchronize()
    assert torch.allclose(A.nzval, ref, rtol=1e-5, atol=1e-6)
