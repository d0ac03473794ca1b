"""The fused step (round 6): the whole Jacobian in ONE launch -- the step-size reduction's workgroups, its finisher and the storing
wavefronts in one grid, values handed over as their own flags (csrc/fdjac_eps_dev.h).  It must give the bits of the two-launch call
(step sizes and every stored value), call after call, on every storage an exact tridiagonal band has; and one rank's share of a
sharded step run alone through a loop-back mailbox must give the bits of the unsharded call's slice."""
import numpy as np
import pytest

import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P
from finitediff_jl_amd import sharded as S

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _tdt(dtype):
    return torch.float64 if dtype == np.float64 else torch.float32


def _storage(kind, N, dtype):
    t = _tdt(dtype)
    if kind == "csc":
        colptr, rowval = P.tridiag_csc(N)
        return fd.SparseMatrixCSC(N, N, colptr, rowval, torch.full((rowval.size,), float("nan"), dtype=t, device="cuda"))
    if kind == "banded":
        return fd.BandedMatrix(torch.full((3 * N,), float("nan"), dtype=t, device="cuda"), N, 1, 1)
    return fd.Tridiagonal(*(torch.full((n,), float("nan"), dtype=t, device="cuda") for n in (N - 1, N, N - 1)))


def _outs(J):
    if isinstance(J, fd.SparseMatrixCSC):
        return [J.nzval]
    if isinstance(J, fd.BandedMatrix):
        return [J.data]
    return [J.dl, J.d, J.du]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("kind", ["csc", "banded", "tridiagonal"])
@pytest.mark.parametrize("family,C,N", [("tridiag_nl", 3, 16385), ("tridiag", 3, 10 ** 6), ("tridiag_nl", 4, 300001), ("tridiag_nl", 7, 2 ** 21)])
def test_fused_step_has_the_bits_of_the_two_launch_call(dtype, fdtype, kind, family, C, N):
    colors = P.cyclic_colors(N, C)
    f = fd.BuiltinF(family, N, dtype=dtype)
    res = {}
    rng = np.random.default_rng(N % 1000 + C)
    xs = [torch.as_tensor(((rng.random(N) * 2 - 0.5) * (1 + 3 * it)).astype(dtype), device="cuda") for it in range(3)]
    for fused in (False, True):
        J = _storage(kind, N, dtype)
        plan = fd.make_plan(J, J, colors, fdtype, dtype=dtype)
        plan.set_lazy(f, fused=fused)
        assert plan.info(fd.lib.INFO_LAZY_STORE) == 1
        got = []
        for it in range(5):       # (five calls: both parities of the hand-over slots, a new x every time)
            x = xs[it % 3]
            for o in _outs(J):
                o.fill_(float("nan"))
            plan.enable_timing(2)
            plan.jacobian(f, x, _outs(J))
            tm = plan.timings()
            plan.enable_timing(0)
            # the fused call has NO launch of its own for the step sizes
            assert tm["eps"]["launches"] == (0 if fused else 1), (fused, tm)
            got.append(([o.clone() for o in _outs(J)], plan.epsilons()))
        res[fused] = got
    for (oa, ea), (ob, eb) in zip(res[False], res[True]):
        assert np.array_equal(ea, eb)
        for a, b in zip(oa, ob):
            assert not torch.isnan(b).any()
            assert torch.equal(a, b)


def test_fused_step_matches_the_oracle(oracle):
    N = 40000
    x = np.random.default_rng(11).random(N)
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    for fdtype in ("forward", "central"):
        J = fd.SparseMatrixCSC(N, N, colptr, rowval, torch.zeros(rowval.size, dtype=torch.float64, device="cuda"))
        f = fd.BuiltinF("tridiag_nl", N)
        plan = fd.make_plan(J, J, colors, fdtype)
        plan.set_lazy(f)
        plan.enable_timing(2)
        plan.jacobian(f, torch.as_tensor(x, device="cuda"), [J.nzval])
        assert plan.timings()["eps"]["launches"] == 0          # the fused step ran
        ref = oracle.jacobian(fdtype, oracle.Fixture("tridiag_nl", N), x, colors, kind=oracle.PAT_CSC_COMMON, colptr=colptr, rowval=rowval)
        got = J.nzval.cpu().numpy()
        err = np.max(np.abs(got - ref["out"]) / (1e-6 * np.abs(ref["out"]) + 1e-7))
        assert err <= 1.0, (fdtype, err)
        rel = np.sqrt(np.finfo(np.float64).eps) if fdtype == "forward" else np.cbrt(np.finfo(np.float64).eps)
        want = np.array([max(rel * np.sqrt(np.sqrt(np.sum(x[colors == c + 1] ** 2))), rel) for c in range(3)])
        assert np.allclose(plan.epsilons(), want, rtol=1e-12, atol=0)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("W", [2, 3, 8])
def test_rank_share_through_a_loopback_mailbox_has_the_bits_of_the_unsharded_slice(dtype, W, fused):
    # rank r of W alone on this GPU: its groups of the reduction, ONE exchange launch against a mailbox whose peers' slots hold the
    # true group sums and halos (fd_p2p_create_loopback), the storing launch on its columns -- every rank's slice of the unsharded call.
    # fused: all of that in ONE launch (the finishers store the rank's group sums into the peers' cells and poll their own)
    N = 700001
    t = _tdt(dtype)
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    pattern = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    f = fd.BuiltinF("tridiag_nl", N, dtype=dtype)
    x_full = torch.as_tensor(np.random.default_rng(5).random(N).astype(dtype), device="cuda")
    plan1 = fd.make_plan(pattern, pattern, colors, "forward", dtype=dtype)
    plan1.set_lazy(f)
    out1 = torch.empty(rowval.size, dtype=t, device="cuda")
    plan1.jacobian(f, x_full, [out1])
    eps1 = plan1.epsilons()
    for b in range(W):
        pptr, slot = plan1.eps_partials(x_full, b, W)
    torch.cuda.synchronize()

    class _Raw:
        __cuda_array_interface__ = {"shape": (W * slot,), "typestr": "<f8", "data": (pptr, False), "version": 2}
    gsum = torch.as_tensor(_Raw(), device="cuda").clone()
    cuts = S.eps_shard_cuts(N, W)
    halo = 2
    ctx = fd.Context.default()
    for r in range(W):
        c0, c1 = int(cuts[r]), int(cuts[r + 1])
        e0 = 0 if c0 <= 0 else 3 * c0 - 1
        e1 = 3 * N - 2 if c1 >= N else 3 * c1 - 1
        plan = fd.make_plan(pattern, pattern, colors, "forward", col_window=(c0, c1), x_window=S.x_window(cuts, r, N, 1, 1, 1), dtype=dtype)
        plan.set_lazy(f, fused=fused)
        mb = fd.P2P.loopback(ctx, W, r, 1 << 16)
        for b in range(W):
            if b == r:
                continue
            mb.fill(b, 0, gsum[b * slot:(b + 1) * slot])
            if b == r - 1:
                mb.fill(b, slot * 8, x_full[c0 - halo:c0].contiguous())
            if b == r + 1:
                mb.fill(b, slot * 8, x_full[c1:c1 + halo].contiguous())
        mb.fill_fused(gsum[:512].contiguous(), x_full[c0 - halo:c0].contiguous() if r > 0 else None,
                      x_full[c1:c1 + halo].contiguous() if r + 1 < W else None)
        plan.set_p2p(mb)
        plan.set_halo(c0, c1, halo)
        x = torch.full_like(x_full, float("nan"))
        x[c0:c1] = x_full[c0:c1]
        out = torch.full((e1 - e0,), float("nan"), dtype=t, device="cuda")
        for _ in range(4):
            out.fill_(float("nan"))
            x[:c0] = float("nan")
            x[c1:] = float("nan")
            plan.enable_timing(2)
            plan.jacobian(f, x, [out])
            tm = plan.timings()
            plan.enable_timing(0)
            assert tm["eps"]["launches"] == (0 if fused else 1) and tm["exchange"]["launches"] == (0 if fused else 1), tm
            assert mb.status() == 0
            assert np.array_equal(plan.epsilons(), eps1), (W, r)
            if not torch.equal(out, out1[e0:e1]):      # (say what differs: a missing store, a wrong halo, a wrong step size ...)
                d = torch.nonzero((out != out1[e0:e1]) | torch.isnan(out)).flatten()
                raise AssertionError((W, r, _, int(d.numel()), d[:4].tolist(), d[-2:].tolist(), int(torch.isnan(out).sum()), out[d[:4]].tolist(),
                                      out1[e0:e1][d[:4]].tolist(), x[c0 - halo:c0 + 1].tolist() if r > 0 else None))
        # the halo cells of x arrived with the call
        if r > 0:
            assert torch.equal(x[c0 - halo:c0], x_full[c0 - halo:c0])
        if r + 1 < W:
            assert torch.equal(x[c1:c1 + halo], x_full[c1:c1 + halo])
