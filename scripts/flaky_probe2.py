"""Reproduce the W = 8 loop-back failure (second test of a process) and show WHICH group sums are wrong."""
import os, sys
os.environ["FDJAC_TEST_SWITCHES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P, sharded as S

def run(dtype, W, fused, tag):
    t = torch.float32 if dtype == np.float32 else torch.float64
    N = 700001
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
    ctx = fd.Context.default()
    nbad = 0
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
                mb.fill(b, slot * 8, x_full[c0 - 2:c0].contiguous())
            if b == r + 1:
                mb.fill(b, slot * 8, x_full[c1:c1 + 2].contiguous())
        mb.fill_fused(gsum[:512].contiguous(), x_full[c0 - 2:c0].contiguous() if r > 0 else None, x_full[c1:c1 + 2].contiguous() if r + 1 < W else None)
        plan.set_p2p(mb)
        plan.set_halo(c0, c1, 2)
        x = torch.full_like(x_full, float("nan"))
        x[c0:c1] = x_full[c0:c1]
        gp, gslot = plan.eps_partials(x, r, W)        # (the address of this plan's group-sum buffer)
        torch.cuda.synchronize()
        class _R2:
            __cuda_array_interface__ = {"shape": (512,), "typestr": "<f8", "data": (gp, False), "version": 2}
        pg = torch.as_tensor(_R2(), device="cuda")
        out = torch.full((e1 - e0,), float("nan"), dtype=t, device="cuda")
        for it in range(4):
            out.fill_(float("nan"))
            x[:c0] = float("nan"); x[c1:] = float("nan")
            plan.jacobian(f, x, [out])
            e = plan.epsilons()
            if not np.array_equal(e, eps1) or not torch.equal(out, out1[e0:e1]):
                nbad += 1
                d = torch.nonzero(pg != gsum[:512]).flatten().tolist()
                print(tag, "W", W, "rank", r, "it", it, "eps", e, "want", eps1, "out_equal", bool(torch.equal(out, out1[e0:e1])), "st", mb.status())
                print("   group-sum entries that differ (g, c):", [(i // 8, i % 8) for i in d][:24], "got", pg[d[:6]].tolist(), "want", gsum[d[:6]].tolist())
    print(tag, "W", W, "dtype", np.dtype(dtype).name, "fused", fused, "bad calls:", nbad)

run(np.float64, 8, False, "first")
run(np.float64, 8, False, "second")
run(np.float32, 8, False, "third")
run(np.float64, 8, True, "fourth")
run(np.float64, 8, True, "fifth")
