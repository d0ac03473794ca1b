"""The device plan builder (fdjac_planbuild.hip) against the host builder, which stays in the library as its checker:
every plan is built both ways (FDJAC_PLAN_DEVICE=0 / 1) and the compiled arrays are compared through fd_plan_checksum,
the Jacobians bit for bit.  Replaces the per-call pattern work of the reference (src/jacobians.jl:524-535,547;
ext/FiniteDiffSparseArraysExt.jl:38-47,51-52)."""
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


def _case(name, N):
    win = None
    if name == "band5":
        colptr, rowval = P.banded_csc(N, N, 2, 2)
        colors = P.cyclic_colors(N, 5)
    elif name == "bidiag":
        colptr, rowval = P.banded_csc(N, N, 1, 0)
        colors = P.cyclic_colors(N, 2)
    else:
        colptr, rowval = P.tridiag_csc(N)
        colors = P.cyclic_colors(N, 3)
    if name == "tridiag_none":
        colors = colors.copy()
        colors[[0, 5, 4096, N // 2, N - 1]] = 0
    if name == "tridiag_shifted":
        colors = ((np.arange(N) + 2) % 3 + 1).astype(np.int64)
    if name == "tridiag_irregular":        # not cyclic, tiles with and without periodic codes (a plan does not care
        colors = colors.copy()             # whether the colouring is valid)
        colors[::1001] = colors[::1001] % 3 + 1
    if name == "tridiag_window":
        win = (N // 5 + 1, 4 * N // 5)
    return colptr, rowval, colors, win


@pytest.mark.parametrize("name", ["tridiag", "tridiag_none", "tridiag_shifted", "tridiag_irregular", "tridiag_window", "band5", "bidiag"])
@pytest.mark.parametrize("fdtype", ["forward", "complex"])
def test_device_built_plan_equals_host_built_plan(monkeypatch, name, fdtype):
    N = 300_011
    colptr, rowval, colors, win = _case(name, N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    x = _dev(np.random.default_rng(3).random(N))
    plans, outs = {}, {}
    for dev in ("0", "1"):
        monkeypatch.setenv("FDJAC_PLAN_DEVICE", dev)
        plan = fd.make_plan(J, J, colors, fdtype, col_window=win, x_window=(win[0] - 3, win[1] + 3) if win else None)
        assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == int(dev)
        assert plan.info(fd.lib.INFO_WINDOW) == 1
        out = _dev(np.full(plan.out_len(0), np.nan))
        f = fd.BuiltinF("tridiag_nl", N)
        plan.set_lazy(f)
        plan.jacobian(f, x, [out])
        plans[dev], outs[dev] = plan, out
    assert plans["0"].checksum() == plans["1"].checksum()
    for key in (fd.lib.INFO_WIN_PERIOD, fd.lib.INFO_WIN_OVERREAD_X100, fd.lib.INFO_EPS_CYCLIC, fd.lib.INFO_ROW_BEGIN, fd.lib.INFO_ROW_END,
                fd.lib.INFO_NNZ_LOCAL, fd.lib.INFO_ENTRY_BEGIN, fd.lib.INFO_NCOLORS):
        assert plans["0"].info(key) == plans["1"].info(key), key
    assert not torch.isnan(outs["0"]).any() and torch.equal(outs["0"], outs["1"])


@pytest.mark.parametrize("tile", ["2048", "1024", "512"])
@pytest.mark.parametrize("periodic", ["1", "0"])
def test_device_builder_follows_the_tuning_switches(monkeypatch, tile, periodic):
    N = 200_003
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    colors = P.cyclic_colors(N, 3)
    monkeypatch.setenv("FDJAC_WIN_TILE", tile)
    monkeypatch.setenv("FDJAC_WIN_PERIODIC", periodic)
    sums = []
    for dev in ("0", "1"):
        monkeypatch.setenv("FDJAC_PLAN_DEVICE", dev)
        plan = fd.make_plan(J, J, colors, "central")
        assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == int(dev)
        assert (plan.info(fd.lib.INFO_WIN_PERIOD) > 0) == (periodic == "1")
        sums.append(plan.checksum())
    assert sums[0] == sums[1]


@pytest.mark.parametrize("idx", ["int32_0based", "int64_1based"])
def test_plan_from_a_device_resident_pattern(monkeypatch, idx):
    # fd_plan_create_csc_device: colptr / rowval / colorvec are device arrays (e.g. a device sparse matrix's colPtr / rowVal)
    N = 400_009
    colptr, rowval = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    monkeypatch.setenv("FDJAC_PLAN_DEVICE", "0")
    ref = fd.make_plan(J, J, colors, "forward")
    monkeypatch.delenv("FDJAC_PLAN_DEVICE")
    if idx == "int32_0based":
        cp, rv, cv, base = (torch.as_tensor((colptr - 1).astype(np.int32), device="cuda"), torch.as_tensor((rowval - 1).astype(np.int32), device="cuda"),
                            torch.as_tensor(colors.astype(np.int32), device="cuda"), 0)
    else:
        cp, rv, cv, base = torch.as_tensor(colptr, device="cuda"), torch.as_tensor(rowval, device="cuda"), torch.as_tensor(colors, device="cuda"), 1
    plan = fd.make_plan_csc_device(N, N, cp, rv, cv, "forward", idx_base=base)
    assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == 1 and plan.checksum() == ref.checksum()
    x = _dev(np.random.default_rng(4).random(N))
    a, b = _dev(np.full(rowval.size, np.nan)), _dev(np.full(rowval.size, np.nan))
    f = fd.BuiltinF("tridiag_nl", N)
    ref.jacobian(f, x, [a])
    plan.jacobian(f, x, [b])
    assert torch.equal(a, b)


def test_patterns_the_device_builder_declines_go_to_the_host_builder(monkeypatch):
    monkeypatch.setenv("FDJAC_PLAN_DEVICE", "1")
    nx, ny = 48, 3000                              # 5-point stencil on a narrow grid (stride 48 < 64: no 2-D tiles): tiles
    colptr, rowval = P.lap5_csc(nx, ny)            # need several row windows, clustered by the host builder
    J = fd.SparseMatrixCSC(nx * ny, nx * ny, colptr, rowval)
    plan = fd.make_plan(J, J, P.lap5_colors(nx, ny), "central")
    assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == 0 and plan.info(fd.lib.INFO_WINDOW) == 1
    N = 200_000                                    # a BandedMatrix with more colours than a window tile holds: host builder
    plan = fd.make_plan(fd.BandedMatrix(None, N, 6, 6), None, P.cyclic_colors(N, 13), "forward")
    assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == 0


@pytest.mark.parametrize("fdtype", FDTYPES)
@pytest.mark.parametrize("case", ["band13", "band13_none", "band13_window", "band5_c11", "lap7_c12"])
def test_many_colours_built_on_the_device(monkeypatch, case, fdtype):
    # More than 8 colours (round 4): the entry lists come from the device builder, the per-colour column lists of the step-size
    # reduction from the colours copied back (N bytes).  Same plan as the host builder's, same Jacobian bits as the plain gather kernels.
    win = None
    if case.startswith("band13"):
        N = 200_000
        colptr, rowval = P.banded_csc(N, N, 6, 6)
        colors = P.cyclic_colors(N, 13)
        if case == "band13_none":
            colors = colors.copy()
            colors[[3, 777, N // 2, N - 2]] = 0
        if case == "band13_window":
            win = (N // 5 + 1, 4 * N // 5)
    elif case == "band5_c11":
        N = 260_003
        colptr, rowval = P.banded_csc(N, N, 2, 2)
        colors = ((np.arange(N) * 7) % 11 + 1).astype(np.int64)
    else:
        n1, n2, n3 = 83, 47, 41
        colptr, rowval, _ = _lap7_csc(n1, n2, n3)
        N = n1 * n2 * n3
        k = np.arange(N)
        colors = (((k % n1) + 2 * ((k // n1) % n2) + 3 * (k // (n1 * n2))) % 12 + 1).astype(np.int64)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    x = _dev(np.random.default_rng(9).random(N) + 0.1)

    def fn(fx, xx):
        fx.copy_(xx.roll(1) ** 2 + 3 * xx + xx.roll(-5) * xx)

    plans, outs = {}, {}
    for dev in ("0", "1", "ref"):
        monkeypatch.setenv("FDJAC_PLAN_DEVICE", "0" if dev == "ref" else dev)
        if dev == "ref":
            monkeypatch.setenv("FDJAC_WINDOW", "0")
            monkeypatch.setenv("FDJAC_SORTED", "0")
        plan = fd.make_plan(J, J, colors, fdtype, col_window=win)
        if dev != "ref":
            assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == int(dev), case
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(fd.TorchF(fn, N, N), x, [out])
        plans[dev], outs[dev] = plan, out
    assert plans["0"].checksum() == plans["1"].checksum(), case
    for key in (fd.lib.INFO_WINDOW, fd.lib.INFO_SORTED_GATHER, fd.lib.INFO_EPS_CYCLIC, fd.lib.INFO_ROW_BEGIN, fd.lib.INFO_ROW_END, fd.lib.INFO_NNZ_LOCAL,
                fd.lib.INFO_NCOLORS, fd.lib.INFO_LAZY_STORE):
        assert plans["0"].info(key) == plans["1"].info(key), key
    assert np.array_equal(plans["0"].epsilons(), plans["1"].epsilons())
    assert not torch.isnan(outs["ref"]).any()
    for dev in ("0", "1"):
        assert torch.equal(outs[dev], outs["ref"]), (dev, case)
    if case == "band13":    # the same through the device-pointer entry point: nothing crosses PCIe but the N colour bytes
        monkeypatch.setenv("FDJAC_PLAN_DEVICE", "1")
        monkeypatch.delenv("FDJAC_WINDOW")
        monkeypatch.delenv("FDJAC_SORTED")
        cp, rv = torch.as_tensor(colptr, device="cuda"), torch.as_tensor(rowval, device="cuda")
        plan2 = fd.make_plan_csc_device(N, N, cp, rv, torch.as_tensor(colors, device="cuda"), fdtype)
        assert plan2.info(fd.lib.INFO_BUILT_ON_DEVICE) == 1 and plan2.checksum() == plans["0"].checksum()


def test_device_builder_reports_an_inconsistent_pattern(monkeypatch):
    monkeypatch.setenv("FDJAC_PLAN_DEVICE", "1")
    N = 150_000
    colptr, rowval = P.tridiag_csc(N)
    bad = rowval.copy()
    bad[12345] = N + 7                              # row outside 1..M
    with pytest.raises(fd.lib.FdError) as e:
        fd.make_plan(fd.SparseMatrixCSC(N, N, colptr, bad), fd.SparseMatrixCSC(N, N, colptr, bad), P.cyclic_colors(N, 3), "forward")
    assert e.value.code == 2                        # FD_ERR_SHAPE


def _stencil_csc(nx, ny, offsets):
    """CSC pattern of a 2-D stencil in natural ordering: column k = i + nx*j couples to rows (i+di) + nx*(j+dj)."""
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    cols, rows = [], []
    for di, dj in offsets:
        ok = (i + di >= 0) & (i + di < nx) & (j + dj >= 0) & (j + dj < ny)
        cols.append((i + nx * j)[ok])
        rows.append(((i + di) + nx * (j + dj))[ok])
    cols, rows = np.concatenate(cols), np.concatenate(rows)
    order = np.lexsort((rows, cols))
    cols, rows = cols[order], rows[order]
    colptr = np.zeros(nx * ny + 1, dtype=np.int64)
    np.add.at(colptr, cols + 1, 1)
    return np.cumsum(colptr) + 1, rows.astype(np.int64) + 1


@pytest.mark.parametrize("case", ["lap5", "lap5_odd", "lap5_none", "lap5_window", "lap5_wide_rows", "nine_point", "lap5_skewed"])
@pytest.mark.parametrize("fdtype", ["central", "forward", "complex"])
def test_device_built_2d_tiles_equal_host_built(monkeypatch, case, fdtype):
    # 2-D stencil patterns in natural ordering (BASELINE config 3's shape): the strided tiles of k_decompress_window2d are
    # compiled by k_pb2_* on the device; the host builder (try_window2d_plan) is the checker -- same arrays, same Jacobian bits
    nx, ny = {"lap5_odd": (331, 257), "lap5_wide_rows": (5000, 40), "nine_point": (300, 260)}.get(case, (400, 300))
    N = nx * ny
    win = None
    if case == "nine_point":
        colptr, rowval = _stencil_csc(nx, ny, [(di, dj) for di in (-1, 0, 1) for dj in (-1, 0, 1)])
        ii, jj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
        colors = ((ii % 4) + 4 * (jj % 2) + 1).T.reshape(-1).astype(np.int64)      # 8 colours (not a valid colouring: a plan does not care)
    else:
        colptr, rowval = P.lap5_csc(nx, ny)
        colors = P.lap5_colors(nx, ny)
    if case == "lap5_none":
        colors = colors.copy()
        colors[[0, 7, nx + 3, N // 2, N - 1]] = 0
    if case == "lap5_skewed":
        colors = colors.copy()                      # (a plan does not care whether the colouring is valid)
        colors[::997] = colors[::997] % 5 + 1
    if case == "lap5_window":
        win = (N // 7 + 1, 6 * N // 7)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    x = _dev(np.random.default_rng(5).random(N))
    plans, outs = {}, {}
    for dev in ("0", "1"):
        monkeypatch.setenv("FDJAC_PLAN_DEVICE", dev)
        plan = fd.make_plan(J, J, colors, fdtype, col_window=win, x_window=(win[0] - nx - 2, win[1] + nx + 2) if win else None)
        assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == int(dev)
        assert plan.info(fd.lib.INFO_WINDOW) == 1 and plan.info(fd.lib.INFO_WINDOW2D) == 1
        out = _dev(np.full(plan.out_len(0), np.nan))
        f = fd.BuiltinF("lap5_nl" if case != "nine_point" else "lap5", nx, ny)
        if getattr(f, "lazy_fn", None) is not None:     # (the built-in lazy launcher wants an even nx)
            plan.set_lazy(f)
        plan.jacobian(f, x, [out])
        plans[dev], outs[dev] = plan, out
    assert plans["0"].checksum() == plans["1"].checksum()
    for key in (fd.lib.INFO_WIN_OVERREAD_X100, fd.lib.INFO_EPS_CYCLIC, fd.lib.INFO_ROW_BEGIN, fd.lib.INFO_ROW_END, fd.lib.INFO_NNZ_LOCAL,
                fd.lib.INFO_ENTRY_BEGIN, fd.lib.INFO_NCOLORS, fd.lib.INFO_LINES_DIRECT_X100, fd.lib.INFO_LINES_SORTED_X100):
        assert plans["0"].info(key) == plans["1"].info(key), key
    # entries the colouring leaves out are zero-filled, everything else is overwritten: no NaN survives
    assert not torch.isnan(outs["0"]).any() and torch.equal(outs["0"], outs["1"])


def test_2d_plan_from_a_device_resident_pattern_matches_oracle(oracle):
    nx, ny = 420, 310
    N = nx * ny
    colptr, rowval = P.lap5_csc(nx, ny)
    colors = P.lap5_colors(nx, ny)
    cp, rv = torch.as_tensor((colptr - 1).astype(np.int32), device="cuda"), torch.as_tensor((rowval - 1).astype(np.int32), device="cuda")
    cv = torch.as_tensor(colors.astype(np.int32), device="cuda")
    plan = fd.make_plan_csc_device(N, N, cp, rv, cv, "central", idx_base=0)
    assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == 1 and plan.info(fd.lib.INFO_WINDOW2D) == 1
    xh = np.random.default_rng(6).random(N)
    out = _dev(np.full(rowval.size, np.nan))
    f = fd.BuiltinF("lap5_nl", nx, ny)
    plan.set_lazy(f)
    plan.jacobian(f, _dev(xh), [out])
    ref = oracle.jacobian("central", oracle.Fixture("lap5_nl", nx, ny), xh, colors, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
    got = out.cpu().numpy()
    assert np.all(np.isfinite(got))
    np.testing.assert_allclose(got, ref["out"], rtol=1e-6, atol=1e-4)     # central differences at eps ~ 6e-6: ~1e-10/eps absolute floor


@pytest.mark.parametrize("l,u,M,N,special", [(1, 1, 200_003, 200_003, ""), (2, 3, 150_000, 150_000, "none"), (0, 2, 180_001, 180_001, ""),
                                              (3, 0, 120_050, 120_000, "window"), (1, 2, 99_990, 100_000, "irregular")])
@pytest.mark.parametrize("fdtype", ["forward", "complex"])
def test_device_built_banded_plan_equals_host_built(monkeypatch, l, u, M, N, special, fdtype):
    # BandedMatrix J (ext/FiniteDiffBandedMatricesExt.jl:13-27): the band's column-major storage is an entry list with
    # implicit indices; k_pb_tiles<BAND> compiles it without any index array, the host loops are its checker
    w = l + u + 1
    colors = P.cyclic_colors(N, w)
    if special == "none":
        colors[[0, 11, N // 3, N - 1]] = 0
    if special == "irregular":
        colors[::1013] = colors[::1013] % w + 1
    win = (N // 6 + 1, 5 * N // 6) if special == "window" else None
    x = _dev(np.random.default_rng(8).random(N))
    A = torch.as_tensor(np.random.default_rng(9).random((M, w)), device="cuda")

    def fn(fx, xx):   # f_i = sum_k A[i,k] * x[i - l + k]^2 (clamped): rows i depend on columns i-l .. i+u
        idx = torch.arange(M, device="cuda")
        acc = torch.zeros(M, dtype=xx.dtype, device="cuda")
        for k in range(w):
            acc = acc + A[:, k].to(xx.dtype) * xx[torch.clamp(idx - l + k, 0, N - 1)] ** 2
        fx.copy_(acc)

    plans, outs = {}, {}
    for dev in ("0", "1"):
        monkeypatch.setenv("FDJAC_PLAN_DEVICE", dev)
        plan = fd.make_plan(fd.BandedMatrix(None, M, l, u), None, colors, fdtype, col_window=win)
        assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == int(dev) and plan.info(fd.lib.INFO_WINDOW) == 1
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(fd.TorchF(fn, M, N), x, [out])
        plans[dev], outs[dev] = plan, out
    assert plans["0"].checksum() == plans["1"].checksum()
    for key in (fd.lib.INFO_WIN_PERIOD, fd.lib.INFO_WIN_OVERREAD_X100, fd.lib.INFO_EPS_CYCLIC, fd.lib.INFO_ROW_BEGIN, fd.lib.INFO_ROW_END,
                fd.lib.INFO_NNZ_LOCAL, fd.lib.INFO_NCOLORS):
        assert plans["0"].info(key) == plans["1"].info(key), key
    assert not torch.isnan(outs["0"]).any() and torch.equal(outs["0"], outs["1"])


@pytest.mark.parametrize("special", ["", "none", "irregular", "four"])
@pytest.mark.parametrize("fdtype", ["forward", "central", "complex"])
def test_tridiagonal_colours_converted_on_the_device(monkeypatch, special, fdtype):
    # Tridiagonal J (src/iteration_utils.jl:25-32 through its three diagonals): a large plan converts / tests its colours with
    # k_pb_colmax / k_pb_colors; the host loops are the checker
    N = 262_147
    colors = P.cyclic_colors(N, 4 if special == "four" else 3)
    if special == "none":
        colors[[0, 9, N // 2, N - 1]] = 0
    if special == "irregular":
        colors[::1009] = colors[::1009] % 3 + 1
    x = _dev(np.random.default_rng(10).random(N))
    f = fd.BuiltinF("tridiag_nl", N)
    plans, outs = {}, {}
    for dev in ("0", "1"):
        monkeypatch.setenv("FDJAC_PLAN_DEVICE", dev)
        plan = fd.make_plan(fd.Tridiagonal(None, np.empty(N), None), None, colors, fdtype)
        assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == int(dev)
        o = [_dev(np.full(plan.out_len(k), np.nan)) for k in range(3)]
        plan.set_lazy(f)
        plan.jacobian(f, x, o)
        plans[dev], outs[dev] = plan, o
    assert plans["0"].checksum() == plans["1"].checksum()
    cyc = 0 if special in ("none", "irregular") or fdtype == "complex" else (4 if special == "four" else 3)     # (reports C)
    assert plans["0"].info(fd.lib.INFO_EPS_CYCLIC) == plans["1"].info(fd.lib.INFO_EPS_CYCLIC) == cyc
    for a, b in zip(outs["0"], outs["1"]):
        assert not torch.isnan(a).any() and torch.equal(a, b)


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_patterns_device_vs_host_builder(monkeypatch, seed):
    # Randomised sweep over what the device builders take: bands of random width / shape / colouring (cyclic, shifted,
    # irregular, with uncoloured columns), 2-D stencils of random offsets on random grids, random column windows, CSC and
    # BandedMatrix storage.  Whatever each builder decides (row windows, 2-D tiles, computed descriptors, host fallback),
    # the device-built plan must be the host-built plan (checksum) and give the same Jacobian bits.
    rng = np.random.default_rng(int(os.environ.get("FDJAC_TEST_SEED_BASE", "1000")) + seed)     # (exploratory runs shift the base)
    kind = ["band", "banded_matrix", "stencil"][seed % 3]
    fdtype = FDTYPES[int(rng.integers(0, 3))]
    win = None
    if kind == "stencil":
        nx, ny = int(rng.integers(70, 700)), int(rng.integers(40, 500))
        N = M = nx * ny
        offs = {(0, 0), (-1, 0), (1, 0), (0, -1), (0, 1)}
        for cand in [(-1, -1), (1, 1), (-1, 1), (1, -1), (-2, 0), (2, 0)]:
            if rng.random() < 0.25:
                offs.add(cand)
        colptr, rowval = _stencil_csc(nx, ny, sorted(offs))
        C = int(rng.integers(3, 9))
        ii, jj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
        colors = ((ii * int(rng.integers(1, 4)) + jj * int(rng.integers(1, 4))) % C + 1).T.reshape(-1).astype(np.int64)
        l = u = 0
    else:
        N = int(rng.integers(140_000, 400_000))
        l, u = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        M = N + int(rng.integers(-30, 31)) if rng.random() < 0.3 else N
        w = l + u + 1
        C = w if rng.random() < 0.7 else int(rng.integers(1, 9))
        colors = ((np.arange(N) + int(rng.integers(0, C))) % C + 1).astype(np.int64)
        if kind == "band":
            colptr, rowval = P.banded_csc(M, N, l, u)
    style = rng.random()
    if style < 0.2:
        colors[rng.integers(0, N, size=5)] = 0
    elif style < 0.4:
        idx = rng.integers(0, N, size=50)
        colors[idx] = colors[idx] % C + 1
    if rng.random() < 0.3:
        a = int(rng.integers(0, N // 3))
        win = (a + 1, int(rng.integers(a + N // 3, N)))
    x = _dev(rng.random(N))
    wband = l + u + 1
    A = torch.as_tensor(rng.random((M, max(wband, 1))), device="cuda")

    def fn(fx, xx):     # some f! (the plan does not care whether pattern and colouring fit it: same arithmetic both ways)
        idx = torch.arange(M, device="cuda")
        acc = torch.zeros(M, dtype=xx.dtype, device="cuda")
        for k in range(max(wband, 1)):
            acc = acc + A[:, k].to(xx.dtype) * xx[torch.clamp(idx - l + k, 0, N - 1)] ** 2
        fx.copy_(acc)

    plans, outs = {}, {}
    # "0" / "1": host- / device-built plan with every fast path the library chooses; "ref": the plain gather kernels on a
    # host-built plan (no row windows, no periodic codes, no computed descriptors) -- an independent path to the same bits
    for dev in ("0", "1", "ref"):
        monkeypatch.setenv("FDJAC_PLAN_DEVICE", "0" if dev == "ref" else dev)
        if dev == "ref":
            monkeypatch.setenv("FDJAC_WINDOW", "0")
            monkeypatch.setenv("FDJAC_SORTED", "0")
        if kind == "banded_matrix":
            plan = fd.make_plan(fd.BandedMatrix(None, M, l, u), None, colors, fdtype, col_window=win)
        else:
            J = fd.SparseMatrixCSC(M, N, colptr, rowval)
            plan = fd.make_plan(J, J, colors, fdtype, col_window=win)
        if dev == "ref":
            assert plan.info(fd.lib.INFO_WINDOW) == 0 and plan.info(fd.lib.INFO_SORTED_GATHER) == 0
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(fd.TorchF(fn, M, N), x, [out])
        plans[dev], outs[dev] = plan, out
    if plans["0"].checksum() != plans["1"].checksum():
        monkeypatch.setenv("FDJAC_CHECKSUM_TRACE", "1")
        plans["0"].checksum(), plans["1"].checksum()
        print("PLANS", kind, fdtype, N, M, l, u, C, win, [[plans[d].info(k) for k in (fd.lib.INFO_BUILT_ON_DEVICE, fd.lib.INFO_WINDOW, fd.lib.INFO_SORTED_GATHER, fd.lib.INFO_LINES_DIRECT_X100, fd.lib.INFO_LINES_SORTED_X100, fd.lib.INFO_NNZ_LOCAL, fd.lib.INFO_ROW_BEGIN, fd.lib.INFO_ROW_END, fd.lib.INFO_EPS_CYCLIC)] for d in ("0", "1")])
    assert plans["0"].checksum() == plans["1"].checksum(), (kind, fdtype, N, l, u, C, win, [
        (plans[d].info(fd.lib.INFO_BUILT_ON_DEVICE), plans[d].info(fd.lib.INFO_WINDOW), plans[d].info(fd.lib.INFO_SORTED_GATHER),
         plans[d].info(fd.lib.INFO_LINES_DIRECT_X100), plans[d].info(fd.lib.INFO_LINES_SORTED_X100)) for d in ("0", "1")])
    for key in (fd.lib.INFO_WINDOW, fd.lib.INFO_WINDOW2D, fd.lib.INFO_SORTED_GATHER, fd.lib.INFO_BAND_DESC, fd.lib.INFO_WIN_PERIOD,
                fd.lib.INFO_EPS_CYCLIC, fd.lib.INFO_ROW_BEGIN, fd.lib.INFO_ROW_END, fd.lib.INFO_NNZ_LOCAL, fd.lib.INFO_NCOLORS):
        assert plans["0"].info(key) == plans["1"].info(key), key
    assert not torch.isnan(outs["ref"]).any()
    for dev in ("0", "1"):
        bad = torch.nonzero(outs[dev] != outs["ref"]).flatten()
        assert bad.numel() == 0, (dev, kind, fdtype, N, M, l, u, C, win, int(bad.numel()), bad[:8].tolist(),
                                  outs[dev][bad[:8]].tolist(), outs["ref"][bad[:8]].tolist())


def _lap7_csc(n1, n2, n3):
    N = n1 * n2 * n3
    k = np.arange(N, dtype=np.int64)
    i, j, l = k % n1, (k // n1) % n2, k // (n1 * n2)
    has = np.stack([l > 0, j > 0, i > 0, np.ones(N, bool), i < n1 - 1, j < n2 - 1, l < n3 - 1], axis=1)
    rows = np.stack([k - n1 * n2, k - n1, k - 1, k, k + 1, k + n1, k + n1 * n2], axis=1)
    colptr = np.concatenate([[0], np.cumsum(has.sum(axis=1))]).astype(np.int64) + 1
    return colptr, (rows[has] + 1).astype(np.int64), ((i + 2 * j + 3 * l) % 7 + 1).astype(np.int64)


def _offsets_csc(N, offsets, rng=None, jitter=0):
    cols, rows = [], []
    j = np.arange(N, dtype=np.int64)
    per = []
    for o in offsets:
        r = j + o + (rng.integers(-jitter, jitter + 1, size=N) if jitter else 0)
        per.append(np.where((r >= 0) & (r < N), r, -1))
    R = np.sort(np.stack(per, axis=1), axis=1)
    keep = R >= 0
    keep[:, 1:] &= R[:, 1:] != R[:, :-1]                 # (a column lists a row once)
    colptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.int64) + 1
    return colptr, (R[keep] + 1).astype(np.int64)


@pytest.mark.parametrize("case", ["lap7", "lap7_flat", "lap7_none", "lap7_window", "far_offsets", "far_jitter", "random_rows", "lap7_int32_device"])
@pytest.mark.parametrize("fdtype", FDTYPES)
def test_index_list_plans_built_on_the_device_equal_host_built(monkeypatch, case, fdtype):
    # Patterns no row-window form describes -- 3-D stencils, far offsets, random rows: the device builder compiles the index lists
    # (tiles sorted by colour and row, their output positions, the f(x) runs of the forward kernel, the far-band tile order) with
    # kernels; the host builder's loops are the checker: same plan arrays (checksum), same Jacobian bits as the plain gather kernels.
    rng = np.random.default_rng(77)
    win = None
    if case.startswith("lap7"):
        n1, n2, n3 = (97, 31, 53) if case == "lap7_flat" else (83, 47, 41)
        colptr, rowval, colors = _lap7_csc(n1, n2, n3)
        N = n1 * n2 * n3
        if case == "lap7_none":
            colors = colors.copy()
            colors[rng.integers(0, N, size=40)] = 0
        if case == "lap7_window":
            win = (N // 7 + 1, 5 * N // 7)
    elif case == "random_rows":
        N = 150_000
        rows = np.sort(rng.integers(0, N, size=(N, 5)), axis=1)
        keep = np.ones_like(rows, bool)
        keep[:, 1:] = rows[:, 1:] != rows[:, :-1]
        colptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.int64) + 1
        rowval = (rows[keep] + 1).astype(np.int64)
        colors = rng.integers(1, 7, size=N).astype(np.int64)
    else:
        N = 180_007
        colptr, rowval = _offsets_csc(N, [-N // 3, -4099, -257, -1, 0, 1, 257, 4099, N // 3], rng, 3 if case == "far_jitter" else 0)
        colors = ((np.arange(N) * 3) % 8 + 1).astype(np.int64)
    x = _dev(rng.random(N))

    def fn(fx, xx):
        fx.copy_(xx.roll(1) ** 2 + 3 * xx + xx.roll(-5) * xx)

    plans, outs = {}, {}
    for dev in ("0", "1", "ref"):
        monkeypatch.setenv("FDJAC_PLAN_DEVICE", "0" if dev == "ref" else dev)
        if dev == "ref":
            monkeypatch.setenv("FDJAC_WINDOW", "0")
            monkeypatch.setenv("FDJAC_SORTED", "0")
        if case == "lap7_int32_device" and dev == "1":
            d = [torch.as_tensor((a - 1).astype(np.int32), device="cuda") for a in (colptr, rowval)] + [torch.as_tensor(colors.astype(np.int32), device="cuda")]
            plan = fd.make_plan_csc_device(N, N, d[0], d[1], d[2], fdtype, idx_base=0)
        else:
            J = fd.SparseMatrixCSC(N, N, colptr, rowval)
            plan = fd.make_plan(J, J, colors, fdtype, col_window=win)
        if dev != "ref":
            assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == int(dev), case
            assert plan.info(fd.lib.INFO_WINDOW) == 0
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(fd.TorchF(fn, N, N), x, [out])
        plans[dev], outs[dev] = plan, out
    assert plans["0"].checksum() == plans["1"].checksum(), case
    for key in (fd.lib.INFO_SORTED_GATHER, fd.lib.INFO_EPS_CYCLIC, fd.lib.INFO_ROW_BEGIN, fd.lib.INFO_ROW_END, fd.lib.INFO_NNZ_LOCAL,
                fd.lib.INFO_NCOLORS, fd.lib.INFO_LINES_DIRECT_X100):
        assert plans["0"].info(key) == plans["1"].info(key), key
    if case != "random_rows":
        assert plans["1"].info(fd.lib.INFO_SORTED_GATHER) == 1
    assert not torch.isnan(outs["ref"]).any()
    for dev in ("0", "1"):
        assert torch.equal(outs[dev], outs["ref"]), (dev, case)


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_scattered_patterns_device_vs_host_builder(monkeypatch, seed):
    # Randomised: 3-D stencils on random grids, random far-offset sets (with and without jitter), random rows; random colourings
    # (cyclic, affine in the grid indices, irregular, with uncoloured columns), random column windows, every fdtype.  Whatever the
    # builders decide -- row windows after all, sorted lists, plain lists -- the device-built plan must be the host-built plan.
    rng = np.random.default_rng(int(os.environ.get("FDJAC_TEST_SEED_BASE", "5000")) + seed)
    kind = ["lap7", "offsets", "jitter", "rows"][seed % 4]
    fdtype = FDTYPES[int(rng.integers(0, 3))]
    C = int(rng.integers(2, 9))
    if kind == "lap7":
        n1, n2, n3 = int(rng.integers(20, 130)), int(rng.integers(8, 70)), int(rng.integers(8, 60))
        colptr, rowval, _ = _lap7_csc(n1, n2, n3)
        N = n1 * n2 * n3
        k = np.arange(N)
        a, b, c = (int(v) for v in rng.integers(1, 4, size=3))
        colors = ((a * (k % n1) + b * ((k // n1) % n2) + c * (k // (n1 * n2))) % C + 1).astype(np.int64)
    elif kind == "rows":
        N = int(rng.integers(60_000, 200_000))
        per = int(rng.integers(2, 7))
        rows = np.sort(rng.integers(0, N, size=(N, per)), axis=1)
        keep = np.ones_like(rows, bool)
        keep[:, 1:] = rows[:, 1:] != rows[:, :-1]
        colptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.int64) + 1
        rowval = (rows[keep] + 1).astype(np.int64)
        colors = rng.integers(1, C + 1, size=N).astype(np.int64)
    else:
        N = int(rng.integers(100_000, 300_000))
        offs = {0}
        for cand in (1, 2, int(rng.integers(20, 100)), int(rng.integers(100, 2000)), int(rng.integers(2000, N // 4)), N // int(rng.integers(2, 5))):
            if rng.random() < 0.7:
                offs.add(cand)
            if rng.random() < 0.7:
                offs.add(-cand)
        colptr, rowval = _offsets_csc(N, sorted(offs), rng, int(rng.integers(1, 6)) if kind == "jitter" else 0)
        colors = ((np.arange(N) * int(rng.integers(1, 4)) + int(rng.integers(0, C))) % C + 1).astype(np.int64)
    style = rng.random()
    if style < 0.25:
        colors[rng.integers(0, N, size=7)] = 0
    elif style < 0.5:
        idx = rng.integers(0, N, size=60)
        colors[idx] = colors[idx] % C + 1
    win = None
    if rng.random() < 0.35:
        a = int(rng.integers(0, N // 3))
        win = (a + 1, int(rng.integers(a + N // 3, N)))
    x = _dev(rng.random(N))

    def fn(fx, xx):
        fx.copy_(xx.roll(1) ** 2 + 3 * xx + xx.roll(-5) * xx)

    plans, outs = {}, {}
    for dev in ("0", "1", "ref"):
        monkeypatch.setenv("FDJAC_PLAN_DEVICE", "0" if dev == "ref" else dev)
        if dev == "ref":
            monkeypatch.setenv("FDJAC_WINDOW", "0")
            monkeypatch.setenv("FDJAC_SORTED", "0")
        J = fd.SparseMatrixCSC(N, N, colptr, rowval)
        plan = fd.make_plan(J, J, colors, fdtype, col_window=win)
        out = _dev(np.full(plan.out_len(0), np.nan))
        plan.jacobian(fd.TorchF(fn, N, N), x, [out])
        plans[dev], outs[dev] = plan, out
    what = (kind, fdtype, N, C, win, [[plans[d].info(k) for k in (fd.lib.INFO_BUILT_ON_DEVICE, fd.lib.INFO_WINDOW, fd.lib.INFO_SORTED_GATHER)] for d in ("0", "1")])
    assert plans["0"].checksum() == plans["1"].checksum(), what
    for key in (fd.lib.INFO_WINDOW, fd.lib.INFO_WINDOW2D, fd.lib.INFO_SORTED_GATHER, fd.lib.INFO_EPS_CYCLIC, fd.lib.INFO_ROW_BEGIN, fd.lib.INFO_ROW_END,
                fd.lib.INFO_NNZ_LOCAL, fd.lib.INFO_NCOLORS):
        assert plans["0"].info(key) == plans["1"].info(key), (key, what)
    assert not torch.isnan(outs["ref"]).any()
    for dev in ("0", "1"):
        assert torch.equal(outs[dev], outs["ref"]), (dev, what)
