"""The device plan builder (fdjac_planbuild.hip) against the host builder, which stays in the library as its checker:
every plan is built both ways (FDJAC_PLAN_DEVICE=0 / 1) and the compiled arrays are compared through fd_plan_checksum,
the Jacobians bit for bit.  Replaces the per-call pattern work of the reference (src/jacobians.jl:524-535,547;
ext/FiniteDiffSparseArraysExt.jl:38-47,51-52)."""
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
    nx, ny = 400, 300                              # 5-point stencil: tiles need several row windows
    colptr, rowval = P.lap5_csc(nx, ny)
    J = fd.SparseMatrixCSC(nx * ny, nx * ny, colptr, rowval)
    plan = fd.make_plan(J, J, P.lap5_colors(nx, ny), "central")
    assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == 0 and plan.info(fd.lib.INFO_WINDOW) == 1
    N = 200_000                                    # many colours: the segmented reduction lists are built on the host
    colptr, rowval = P.banded_csc(N, N, 6, 6)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    plan = fd.make_plan(J, J, P.cyclic_colors(N, 13), "forward")
    assert plan.info(fd.lib.INFO_BUILT_ON_DEVICE) == 0
    # the same through the device-pointer entry point: copied back once, built on the host, same plan
    cp, rv = torch.as_tensor(colptr, device="cuda"), torch.as_tensor(rowval, device="cuda")
    cv = torch.as_tensor(P.cyclic_colors(N, 13), device="cuda")
    plan2 = fd.make_plan_csc_device(N, N, cp, rv, cv, "forward")
    assert plan2.info(fd.lib.INFO_BUILT_ON_DEVICE) == 0 and plan2.checksum() == plan.checksum()


def test_device_builder_reports_an_inconsistent_pattern(monkeypatch):
    monkeypatch.setenv("FDJAC_PLAN_DEVICE", "1")
    N = 150_000
    colptr, rowval = P.tridiag_csc(N)
    bad = rowval.copy()
    bad[12345] = N + 7                              # row outside 1..M
    with pytest.raises(fd.lib.FdError) as e:
        fd.make_plan(fd.SparseMatrixCSC(N, N, colptr, bad), fd.SparseMatrixCSC(N, N, colptr, bad), P.cyclic_colors(N, 3), "forward")
    assert e.value.code == 2                        # FD_ERR_SHAPE
