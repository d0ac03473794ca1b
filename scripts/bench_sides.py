"""Side measurements of bench.py that are not part of the contract line's timed region.

rank_share: ONE rank's share of the W-rank sharded step of the tridiagonal headline problem, run alone on one GPU through a
loop-back mailbox (fd_p2p_create_loopback): the step-size reduction over the rank's own 64 / W groups, the ONE exchange launch
(group sums + halo of x stored into the "peers'" mailbox -- a local sink -- and the peers' contributions, pre-filled with the true
values, copied out of the rank's own mailbox, step sizes formed), the storing launch on the rank's N / W columns -- the launches,
stores and copies of the real W-GPU job, minus the xGMI hop and minus any waiting for a slower peer.  The result is checked against
the unsharded Jacobian's slice bit for bit.  What it gives is a measured per-rank floor, not a scaling curve.
"""
import time

import numpy as np


def _median(v):
    return float(np.median(v)) if len(v) else None


def rank_share(fd, torch, ctx, N, seed, W, ranks=None, steps=200, np_dt=np.float64, family="tridiag"):
    from finitediff_jl_amd import patterns as P
    from finitediff_jl_amd import sharded as S
    dev = torch.device("cuda", torch.cuda.current_device())
    t_dt = torch.float32 if np_dt == np.float32 else torch.float64
    x_host = np.random.default_rng(seed).random(N).astype(np_dt)
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    pattern = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    f = fd.BuiltinF(family, N, ctx=ctx, dtype=np_dt)
    x_full = torch.as_tensor(x_host, device=dev)

    def timed(call, reps):
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    def stages(plan, call, reps):
        plan.enable_timing(4)
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        st = {k: _median(plan.timing_samples(k)) for k in ("eps", "exchange", "decompress")}
        plan.enable_timing(0)
        return {k: (v * 1e3 if v is not None else None) for k, v in st.items()}

    # ---- the unsharded call: T1 and the reference bits ----
    plan1 = fd.make_plan(pattern, pattern, colors, "forward", ctx=ctx, dtype=np_dt)
    plan1.set_lazy(f)
    out1 = torch.empty(rowval.size, dtype=t_dt, device=dev)
    call1 = plan1.bind(f, x_full, [out1])
    t1_us = timed(call1, steps)
    st1 = stages(plan1, call1, min(steps, 50))
    eps1 = plan1.epsilons()
    # every shard's group sums (the slots the peers would deliver)
    slot = 0
    for b in range(W):
        pptr, slot = plan1.eps_partials(x_full, b, W)
    torch.cuda.synchronize()

    class _Raw:
        __cuda_array_interface__ = {"shape": (W * slot,), "typestr": "<f8", "data": (pptr, False), "version": 2}
    gsum = torch.as_tensor(_Raw(), device=dev).clone()
    gs_bytes = slot * 8
    cuts = S.eps_shard_cuts(N, W)
    vs = 4 if np_dt == np.float32 else 8
    halo = 2 if vs == 8 else 2
    res = {"W": W, "N": N, "T1_us": t1_us, "T1_stages_us": st1, "ranks": []}
    if ranks is None:
        ranks = sorted(set([0, W // 2, W - 1]))
    for r in ranks:
        c0, c1 = int(cuts[r]), int(cuts[r + 1])
        e0 = 0 if c0 <= 0 else 3 * c0 - 1
        e1 = 3 * N - 2 if c1 >= N else 3 * c1 - 1
        plan = fd.make_plan(pattern, pattern, colors, "forward", ctx=ctx, col_window=(c0, c1), x_window=S.x_window(cuts, r, N, 1, 1, 1), dtype=np_dt)
        plan.set_lazy(f)
        mb = fd.P2P.loopback(ctx, W, r, 1 << 17)
        for b in range(W):
            if b == r:
                continue
            mb.fill(b, 0, gsum[b * slot:(b + 1) * slot])
            if b == r - 1:
                mb.fill(b, gs_bytes, x_full[c0 - halo:c0].contiguous())
            if b == r + 1:
                mb.fill(b, gs_bytes, x_full[c1:c1 + halo].contiguous())
        mb.fill_fused(gsum[:512].contiguous(), x_full[c0 - halo:c0].contiguous() if r > 0 else None,
                      x_full[c1:c1 + halo].contiguous() if r + 1 < W else None)
        plan.set_p2p(mb)
        plan.set_halo(c0, c1, halo)
        x = torch.full_like(x_full, float("nan"))
        x[c0:c1] = x_full[c0:c1]
        out = torch.full((e1 - e0,), float("nan"), dtype=t_dt, device=dev)
        entry = {"rank": r, "columns": c1 - c0}
        same = True
        for form in ("three_launches", "one_launch"):
            plan.set_lazy(f, fused=(form == "one_launch"))
            call = plan.bind(f, x, [out])
            out.fill_(float("nan"))
            call()
            torch.cuda.synchronize()
            ok = bool(torch.equal(out, out1[e0:e1])) and bool(np.array_equal(plan.epsilons(), eps1)) and mb.status() == 0
            same = same and ok
            step_us = timed(call, steps)
            st = stages(plan, call, min(steps, 50))
            if form == "three_launches":
                eps_only = (st["eps"] - st["exchange"]) if (st["eps"] is not None and st["exchange"] is not None) else None
                entry[form] = {"eps_us": eps_only, "exchange_us": st["exchange"], "store_us": st["decompress"], "step_us": step_us, "bit_identical": ok}
            else:
                entry[form] = {"launch_us": st["decompress"], "step_us": step_us, "bit_identical": ok}
            del call
        entry["step_us"] = min(entry["three_launches"]["step_us"], entry["one_launch"]["step_us"])
        entry["bit_identical_to_unsharded_slice"] = same
        res["ranks"].append(entry)
        del plan, mb, x, out
    worst = max(q["step_us"] for q in res["ranks"])
    res["step_us"] = worst
    res["implied_speedup"] = t1_us / worst
    res["all_bit_identical"] = all(q["bit_identical_to_unsharded_slice"] for q in res["ranks"])
    res["what"] = ("rank r of W alone on one GPU, loop-back mailbox.  three_launches: eps over its own 64/W groups -> ONE exchange launch (stores "
                   "into a local sink, peers' group sums + halo pre-filled) -> storing launch on N/W columns; one_launch: the fused step (the "
                   "finishers of the reduction store the group sums into the peers' cells and poll their own, the storing wavefronts wait for "
                   "the step sizes); step_us = wall clock of %d back-to-back calls / %d "
                   "(worst of the sampled ranks), stage times = medians of HIP-event spans; implied_speedup = T1 / step -- a per-rank FLOOR "
                   "(no xGMI hop, no waiting for a slower peer), not a measured scaling curve" % (steps, steps))
    return res


if __name__ == "__main__":
    import argparse
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import finitediff_jl_amd as fd
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10 ** 7)
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--ranks", type=int, nargs="*", default=[2, 4, 8])
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    torch.cuda.set_stream(torch.cuda.Stream())
    ctx = fd.Context(0)
    for W in a.ranks:
        print(json.dumps(rank_share(fd, torch, ctx, a.n, a.seed, W, steps=a.steps)))
