"""The multi-GPU pieces that one GPU can exercise: the RCCL communicator behind the C ABI (a single-rank communicator
goes through the same RCCL calls), the sharded step-size reduction (shards run one after the other on one device must
give the bits of the unsharded call), and the column-range + gatherv assembly.  The world_size-2 tests of the host logic
run on CPU with gloo (tests/test_sharded_gloo.py)."""
import numpy as np
import pytest

import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P
from finitediff_jl_amd import sharded as S

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


def _nan(n):
    return torch.full((int(n),), float("nan"), dtype=torch.float64, device="cuda")


@pytest.fixture(scope="module")
def comm1():
    ctx = fd.Context.default()
    return fd.Comm(ctx, 1, 0, fd.Comm.unique_id())


def test_comm_single_rank_goes_through_rccl(comm1):
    info = comm1.info()
    assert info["nranks"] == 1 and info["rank"] == 0 and info["rccl_version"] > 20000 and "rccl" in info["library"]
    buf = _dev(np.arange(1000.0))
    want = buf.clone()
    comm1.allgather(buf, 1000)           # in place, one slot
    comm1.allreduce_sum(buf)             # sum over one rank
    comm1.broadcast(buf, root=0)
    send = _dev(np.arange(77.0) + 0.5)
    recv = _nan(77)
    comm1.gatherv(send, recv, [77], root=0)
    comm1.ctx.synchronize()
    assert torch.equal(buf, want) and torch.equal(recv, send)
    f32 = torch.arange(64, dtype=torch.float32, device="cuda")
    comm1.allreduce_sum(f32)
    comm1.ctx.synchronize()
    assert torch.equal(f32, torch.arange(64, dtype=torch.float32, device="cuda"))
    with pytest.raises(fd.lib.FdError):
        comm1.gatherv(send, recv, [76], root=0)      # counts[rank] must match the send length
    with pytest.raises(fd.lib.FdError):
        fd.Comm(comm1.ctx, 2, 5, fd.Comm.unique_id())  # rank outside the communicator


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("C,N", [(3, 10 ** 6 + 7), (5, 300001), (8, 123456)])
def test_sharded_step_size_reduction_bit_identical(fdtype, C, N):
    # fd_plan_eps_partials shard by shard (what each rank of a W-rank job runs: its groups of the two-level sum), fd_plan_eps_finalize,
    # FD_EPS_PRECOMPUTED: the step sizes and the Jacobian have the bits of the plain call, whatever W is
    colors = P.cyclic_colors(N, C)
    if C == 5:
        colors = colors.copy()
        colors[[10, N // 3]] = 0          # not cyclic any more: the reduction reads the colours
    xh = np.random.default_rng(7 + C).random(N) * 2 - 0.5
    x = _dev(xh)
    colptr, rowval = P.banded_csc(N, N, C // 2, C - 1 - C // 2)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    fn = fd.TorchF(lambda fx, xx: fx.copy_(torch.sin(xx) + torch.roll(xx, 1) * xx), N, N)
    ref_plan = fd.make_plan(J, J, colors, fdtype)
    ref = _nan(ref_plan.out_len(0))
    ref_plan.jacobian(fn, x, [ref])
    eps_ref = ref_plan.epsilons()
    for W in (1, 2, 3, 8, 64):
        plan = fd.make_plan(J, J, colors, fdtype)
        for r in reversed(range(W)):       # any order: the slots are disjoint
            ptr, slot = plan.eps_partials(x, r, W)
            assert ptr and slot % 8 == 0
        plan.eps_finalize()
        plan.set_eps_mode(True)
        out = _nan(plan.out_len(0))
        plan.jacobian(fn, x, [out])
        assert np.array_equal(plan.epsilons(), eps_ref), W
        assert torch.equal(out, ref), W
        plan.set_eps_mode(False)
        out.fill_(float("nan"))
        plan.jacobian(fn, x, [out])
        assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("C,N", [(3, 10 ** 6 + 7), (5, 300001), (8, 123456), (3, 16385), (2, 3 * 10 ** 6)])
def test_step_sizes_have_the_bits_of_the_defined_summation_order(dtype, fdtype, C, N):
    # the step sizes are a DEFINED function of x: tests/eps_order.py restates the two-level summation order in numpy, IEEE operation
    # for operation -- the device must give its bits (one launch, every level inside it: the last block of a group adds the group,
    # the last group adds the groups), call after call with a new x each time (a stale block / group sum would show)
    import eps_order
    colors = P.cyclic_colors(N, C)
    if C == 5:
        colors = colors.copy()
        colors[[10, N // 3]] = 0          # not cyclic: the reduction reads the colours; two columns without colour
    colptr, rowval = P.banded_csc(N, N, C // 2, C - 1 - C // 2)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    t_dt = torch.float64 if dtype == np.float64 else torch.float32
    fn = fd.TorchF(lambda fx, xx: fx.copy_(torch.sin(xx)), N, N, dtype=dtype)
    plan = fd.make_plan(J, J, colors, fdtype, dtype=dtype)
    out = torch.empty(plan.out_len(0), dtype=t_dt, device="cuda")
    rng = np.random.default_rng(100 + C)
    for it in range(4):
        xh = ((rng.random(N) * 2 - 0.5) * (1 + 10 * it)).astype(dtype)
        x = torch.as_tensor(xh, device="cuda")
        plan.jacobian(fn, x, [out])
        want = eps_order.epsilons(xh, np.asarray(colors, dtype=np.int64) - 1, C, fdtype, dtype=dtype)
        got = plan.epsilons()
        assert np.array_equal(got, want.astype(np.float64)), (it, got, want)


def test_plan_with_communicator_matches_plain_call(comm1):
    # fd_plan_set_comm: group sums of this rank's groups -> in-place all-gather -> level 2 of the same sum
    N = 2 * 10 ** 6 + 1
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    x = _dev(np.random.default_rng(3).random(N))
    f = fd.BuiltinF("tridiag_nl", N)
    outs = []
    for with_comm in (False, True):
        plan = fd.make_plan(J, J, colors, "forward")
        plan.set_lazy(f)
        if with_comm:
            plan.set_comm(comm1)
        out = _nan(plan.out_len(0))
        plan.jacobian(f, x, [out])
        outs.append((out, plan.epsilons()))
        if with_comm:
            plan.set_comm(None)
    assert torch.equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    # plans whose reduction cannot be sharded say so
    small = fd.make_plan(fd.SparseMatrixCSC(30, 30, *P.tridiag_csc(30)), fd.SparseMatrixCSC(30, 30, *P.tridiag_csc(30)),
                         P.cyclic_colors(30, 3), "forward")
    with pytest.raises(fd.lib.FdError) as e:
        small.eps_partials(_dev(np.ones(30)), 0, 2)
    assert e.value.code == 3   # FD_ERR_UNSUPPORTED


def test_column_ranges_and_gatherv_assemble_the_jacobian(comm1):
    # the data path of `bench.py --gpus W`, all W column ranges on one device: each range's plan fills its slice, the
    # slices are assembled with fd_comm_gatherv (counts / displs as the W-rank job computes them); bits of the full call
    N, W = 300007, 4
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    pat = fd.SparseMatrixCSC(N, N, colptr, rowval)
    x = _dev(np.random.default_rng(5).random(N))
    f = fd.BuiltinF("tridiag_nl", N)
    full = _nan(rowval.size)
    fd.make_plan(pat, pat, colors, "forward").jacobian(f, x, [full])
    cuts = S.partition_columns(colptr, W)
    ranges = S.entry_ranges(colptr, cuts)
    assembled = _nan(rowval.size)
    for r in range(W):
        plan = fd.make_plan(pat, pat, colors, "forward", col_window=(int(cuts[r]), int(cuts[r + 1])),
                            x_window=S.x_window(cuts, r, N, 1, 1, 1))
        a, b = ranges[r]
        piece = _nan(b - a)
        plan.jacobian(f, x, [piece])
        comm1.gatherv(piece, assembled[a:b], [b - a], root=0)    # this rank's slice lands at its displacement
    comm1.ctx.synchronize()
    assert torch.equal(assembled, full)


@pytest.mark.parametrize("fdtype", ["forward", "central"])
@pytest.mark.parametrize("N", [10 ** 6 + 7, 300001])
def test_contiguous_step_size_reduction_reads_only_the_shard(monkeypatch, fdtype, N):
    # the reduction sums contiguous GROUPS of x (64 of them; a shard = whole groups), so shard r of W reads only
    # x[fd_plan_eps_shard_range(r, W)) -- proven by poisoning everything else with NaN -- and the step sizes / the Jacobian of
    # the sharded reduction have the bits of the same plan's unsharded call
    C = 3
    colors = P.cyclic_colors(N, C)
    xh = np.random.default_rng(11).random(N) * 2 - 0.5
    x = _dev(xh)
    colptr, rowval = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval)
    f = fd.BuiltinF("tridiag_nl", N)
    ref_plan = fd.make_plan(J, J, colors, fdtype)
    ref_plan.set_lazy(f)
    ref = _nan(ref_plan.out_len(0))
    ref_plan.jacobian(f, x, [ref])
    eps_ref = ref_plan.epsilons()
    for W in (1, 2, 3, 8, 64):
        plan = fd.make_plan(J, J, colors, fdtype)
        plan.set_lazy(f)
        ranges = [plan.eps_shard_range(r, W) for r in range(W)]
        assert S.eps_shard_cuts(N, W).tolist() == [a for a, _b in ranges] + [N]      # the host-side formula bench.py cuts with
        assert ranges[0][0] == 0 and ranges[-1][1] == N and all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
        for r in reversed(range(W)):
            a, b = ranges[r]
            xs = torch.full_like(x, float("nan"))
            xs[a:b] = x[a:b]                                   # this "rank" holds only its own part of x
            plan.eps_partials(xs, r, W)
            torch.cuda.synchronize()
        plan.eps_finalize()
        plan.set_eps_mode(True)
        out = _nan(plan.out_len(0))
        plan.jacobian(f, x, [out])
        assert np.array_equal(plan.epsilons(), eps_ref), W
        assert torch.equal(out, ref), W
        # column cuts at the shard boundaries: windowed plans (all of them with the contiguous map) concatenate to the full result
        cuts = S.partition_columns_at(ranges, N)
        pieces = []
        if W > 8:
            continue
        for r in range(W):
            wp = fd.make_plan(J, J, colors, fdtype, col_window=(int(cuts[r]), int(cuts[r + 1])))
            wp.set_lazy(f)
            o = _nan(wp.out_len(0))
            wp.jacobian(f, x, [o])
            pieces.append(o)
        assert torch.equal(torch.cat(pieces), ref), W


@pytest.mark.timeout(300)
def test_p2p_mailbox_two_processes_one_gpu():
    # fd_p2p_*: small-message exchange by direct stores into the peers' mailboxes (csrc/fdjac_p2p.hip).  Two processes share the one
    # GPU of the box: the IPC mapping, epochs / parity over many exchanges, the halo channel, the sharded step-size reduction with
    # its partial sums exchanged through the mailbox (bit-identical), a wait that times out instead of hanging.  (What two GPUs
    # would add -- the ordering of remote stores over xGMI -- cannot be executed here.)
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(root, "tests", "p2p_two_ranks.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-5000:]
    assert out.stdout.count("p2p rank") == 2
