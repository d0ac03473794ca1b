"""Two processes, ONE GPU: the peer-to-peer mailbox (fd_p2p_*, csrc/fdjac_p2p.hip) end to end -- IPC mapping of the other
process's mailbox, epochs / parity double-buffering over many exchanges, the halo channel, the sharded step-size reduction with
its partial sums exchanged through the mailbox (bit-identical step sizes), and the timeout of a wait whose peer never arrives.
Launched by tests/test_gpu_multigpu.py::test_p2p_mailbox_two_processes_one_gpu through torch.distributed.run (gloo is only the
out-of-band channel for the 64-byte handles and the barriers)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ["FDJAC_TEST_SWITCHES"] = "1"      # (read once, by the first plan: ranks sharing a device take the fused sharded step on request only)
    os.environ["FDJAC_FUSED_SHARED"] = "1"
    import torch
    import torch.distributed as dist
    import finitediff_jl_amd as fd
    from finitediff_jl_amd import patterns as P
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = fd.Context(0)
    p2p = fd.P2P.from_torch_distributed(ctx, dist, slot_bytes=1 << 16)
    info = p2p.info()
    assert info["nranks"] == world and info["rank"] == rank and info["slot_bytes"] >= 1 << 16

    # (1) all-gather channel: many exchanges back to back, new payload every time, nothing synchronises in between
    S = 1000
    bufs = []
    for it in range(64):
        buf = torch.zeros(world * S, dtype=torch.float64, device=dev)
        buf[rank * S:(rank + 1) * S] = torch.arange(S, dtype=torch.float64, device=dev) + 1e6 * rank + it
        p2p.allgather(buf, S)
        bufs.append(buf)
    torch.cuda.synchronize()
    for it, buf in enumerate(bufs):
        want = torch.cat([torch.arange(S, dtype=torch.float64, device=dev) + 1e6 * r + it for r in range(world)])
        assert torch.equal(buf, want), ("allgather", it)
    assert p2p.status() == 0

    # (1b) what one exchange costs here (two processes sharing ONE GPU -- launch + flag polling, no xGMI hop): printed, not asserted
    t_us = {}
    for nS in (8, 1000, 7750):                          # 64 B, 8 KB, 62 KB per rank (the c4 step-size partials are 62 KB)
        buf = torch.zeros(world * nS, dtype=torch.float64, device=dev)
        for _ in range(20):
            p2p.allgather(buf, nS)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            p2p.allgather(buf, nS)
        e1.record()
        torch.cuda.synchronize()
        t_us[nS * 8] = e0.elapsed_time(e1) * 1e3 / 200
    assert p2p.status() == 0
    if rank == 0:
        print("p2p allgather per exchange (2 processes, 1 GPU): " + ", ".join("%d B: %.1f us" % kv for kv in t_us.items()))

    # (2) halo channel, interleaved with all-gathers (independent epochs per channel)
    N, halo = 4096, 2
    cuts = [0, N // 2, N] if world == 2 else list(np.linspace(0, N, world + 1).astype(int))
    a, b = int(cuts[rank]), int(cuts[rank + 1])
    for it in range(10):
        x = torch.full((N,), float("nan"), dtype=torch.float64, device=dev)
        x[a:b] = torch.arange(a, b, dtype=torch.float64, device=dev) * (it + 1)
        p2p.halo_exchange(x, a, b, halo)
        g = torch.zeros(world * 8, dtype=torch.float64, device=dev)
        g[rank * 8:(rank + 1) * 8] = rank + it
        p2p.allgather(g, 8)
        torch.cuda.synchronize()
        lo, hi = max(a - halo, 0), min(b + halo, N)
        assert torch.equal(x[lo:hi], torch.arange(lo, hi, dtype=torch.float64, device=dev) * (it + 1)), ("halo", it)
        assert torch.isnan(x[:lo]).all() and torch.isnan(x[hi:]).all()
        assert torch.equal(g, torch.arange(world, dtype=torch.float64, device=dev).repeat_interleave(8) + it)

    # (3) the sharded step-size reduction with its partial sums exchanged through the mailbox: the bits of the unsharded call
    Nn = 300_001
    cp, rv = P.tridiag_csc(Nn)
    colors = P.cyclic_colors(Nn, 3)
    pat = fd.SparseMatrixCSC(Nn, Nn, cp, rv, None)
    xs = torch.as_tensor(np.random.default_rng(5).random(Nn), device=dev)
    f = fd.BuiltinF("tridiag_nl", Nn, ctx=ctx)
    ref_plan = fd.make_plan(pat, pat, colors, "forward", ctx=ctx)
    out_ref = torch.empty(rv.size, dtype=torch.float64, device=dev)
    ref_plan.jacobian(f, xs, [out_ref])
    eps_ref = ref_plan.epsilons()
    plan = fd.make_plan(pat, pat, colors, "forward", ctx=ctx)
    pptr, slot = plan.eps_partials(xs, rank, world)

    class _Raw:
        __cuda_array_interface__ = {"shape": (world * slot,), "typestr": "<f8", "data": (pptr, False), "version": 2}
    part = torch.as_tensor(_Raw(), device=dev)
    p2p.allgather(part, slot)
    plan.eps_finalize()
    plan.set_eps_mode(True)
    out = torch.empty_like(out_ref)
    plan.jacobian(f, xs, [out])
    assert np.array_equal(plan.epsilons(), eps_ref) and torch.equal(out, out_ref)
    assert p2p.status() == 0
    dist.barrier()

    # (3b) the per-step exchange of a sharded call as ONE launch (fd_plan_set_p2p + fd_plan_set_halo): each rank holds only its own part
    # of x (NaN elsewhere), cut where the reduction's groups are cut; the call exchanges the halo and the group sums and finishes the
    # step sizes itself.  Many steps back to back with a new x each time and unequal work between the ranks (rank 1 does extra launches:
    # a stale group sum or halo from the step before would show); the bits of the unsharded call on the full x.
    from finitediff_jl_amd import sharded as S
    ranges = [ref_plan.eps_shard_range(r, world) for r in range(world)]
    cuts3 = S.partition_columns_at(ranges, Nn)
    c0, c1 = int(cuts3[rank]), int(cuts3[rank + 1])
    ent = S.entry_ranges(cp, cuts3)
    e0, e1 = ent[rank]
    # ... in BOTH forms: three launches (reduction, ONE exchange launch, store) and the FUSED step -- one launch per rank whose finishers
    # store the group sums straight into the peer's mailbox cells and poll their own, every cell its own flag, three buffers by epoch.
    # (Ranks sharing a device take the fused step only on request, FDJAC_FUSED_SHARED=1: this problem is small enough for both ranks'
    # launches to be resident together.)
    wplan = fd.make_plan(pat, pat, colors, "forward", ctx=ctx, col_window=(c0, c1), x_window=S.x_window(cuts3, rank, Nn, 1, 1, 1))
    wplan.set_p2p(p2p)
    wplan.set_halo(c0, c1, 2)
    ref_plan.set_lazy(f, fused=False)
    rng = np.random.default_rng(77)
    lo, hi = max(c0 - 2, 0), min(c1 + 2, Nn)
    junk = torch.zeros(1 << 22, dtype=torch.float64, device=dev)
    for it in range(80):
        fused = it >= 40 or it % 7 == 3          # (and a few fused steps in between the others: the two forms keep separate epochs)
        wplan.set_lazy(f, fused=fused)
        xfull = torch.as_tensor(rng.random(Nn) * (1.0 + it), device=dev)
        xmine = torch.full((Nn,), float("nan"), dtype=torch.float64, device=dev)
        xmine[c0:c1] = xfull[c0:c1]
        if rank == 1 and it % 3 == 0:
            for _ in range(5):
                junk.add_(1.0)                       # unequal load: this rank arrives late
        piece = torch.full((e1 - e0,), float("nan"), dtype=torch.float64, device=dev)
        wplan.enable_timing(2)
        wplan.jacobian(f, xmine, [piece], sync=False)
        ref_plan.jacobian(f, xfull, [out_ref])
        torch.cuda.synchronize()
        tm = wplan.timings()
        wplan.enable_timing(0)
        assert tm["exchange"]["launches"] == (0 if fused else 1), (it, fused, tm)         # the fused step really is ONE launch
        assert np.array_equal(wplan.epsilons(), ref_plan.epsilons()), ("step eps", it)
        assert torch.equal(piece, out_ref[e0:e1]), ("step", it)
        assert torch.equal(xmine[lo:hi], xfull[lo:hi]) and torch.isnan(xmine[:lo]).all() and torch.isnan(xmine[hi:]).all(), ("step halo", it)
    assert p2p.status() == 0
    xmine[c0:c1] = xfull[c0:c1]
    dist.barrier()
    for fused in (False, True):
        wplan.set_lazy(f, fused=fused)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        call = wplan.bind(f, xmine, [piece])
        for _ in range(10):
            call()
        torch.cuda.synchronize()
        dist.barrier()
        ev0.record()
        for _ in range(200):
            call()
        ev1.record()
        torch.cuda.synchronize()
        assert p2p.status() == 0
        if rank == 0:
            print("sharded call, %s (2 processes, 1 GPU, N = %d): %.1f us per call" % ("fused: ONE launch" if fused else "three launches", Nn, ev0.elapsed_time(ev1) * 1e3 / 200))
        del call
        dist.barrier()
    wplan.set_p2p(None)
    dist.barrier()

    # (4) a peer that never arrives: the wait gives up after FDJAC_P2P_TIMEOUT_MS and raises the status word -- no hang
    os.environ["FDJAC_P2P_TIMEOUT_MS"] = "150"
    if rank == 0:
        g = torch.zeros(world * 8, dtype=torch.float64, device=dev)
        p2p.allgather(g, 8)
        torch.cuda.synchronize()
        assert p2p.status() == 2, p2p.status()          # 1 + (rank 1)
    dist.barrier()
    print("p2p rank %d ok (uncached mailbox: %s)" % (rank, info["uncached"]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
