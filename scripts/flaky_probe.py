"""Which side of test_rank_share...[8-float32] is flaky?  (a) fused vs unfused call, (b) the loop-back rank share vs the unfused call."""
import os, sys
os.environ["FDJAC_TEST_SWITCHES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P, sharded as S
dtype = np.float32 if (len(sys.argv) < 2 or sys.argv[1] == "f32") else np.float64
t = torch.float32 if dtype == np.float32 else torch.float64
N, W = 700001, 8
colors = P.cyclic_colors(N, 3)
colptr, rowval = P.tridiag_csc(N)
pattern = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
f = fd.BuiltinF("tridiag_nl", N, dtype=dtype)
x_full = torch.as_tensor(np.random.default_rng(5).random(N).astype(dtype), device="cuda")
ref_plan = fd.make_plan(pattern, pattern, colors, "forward", dtype=dtype)
ref_plan.set_lazy(f, fused=False)
ref = torch.empty(rowval.size, dtype=t, device="cuda")
ref_plan.jacobian(f, x_full, [ref])
fz_plan = fd.make_plan(pattern, pattern, colors, "forward", dtype=dtype)
fz_plan.set_lazy(f)
out = torch.empty_like(ref)
bad = 0
for it in range(300):
    out.fill_(float("nan"))
    fz_plan.jacobian(f, x_full, [out])
    if not torch.equal(out, ref):
        bad += 1
        d = (out != ref) | torch.isnan(out)
        idx = torch.nonzero(d).flatten()
        print("fused mismatch it", it, "count", idx.numel(), "first", idx[:6].tolist(), "nan", int(torch.isnan(out).sum()), out[idx[:3]].tolist(), ref[idx[:3]].tolist())
print("fused: %d / 300 calls differ" % bad)
for b in range(W):
    pptr, slot = ref_plan.eps_partials(x_full, b, W)
torch.cuda.synchronize()
class _Raw:
    __cuda_array_interface__ = {"shape": (W * slot,), "typestr": "<f8", "data": (pptr, False), "version": 2}
gsum = torch.as_tensor(_Raw(), device="cuda").clone()
cuts = S.eps_shard_cuts(N, W)
ctx = fd.Context.default()
for r in range(W):
    c0, c1 = int(cuts[r]), int(cuts[r + 1])
    e0 = 0 if c0 <= 0 else 3 * c0 - 1
    e1 = 3 * N - 2 if c1 >= N else 3 * c1 - 1
    plan = fd.make_plan(pattern, pattern, colors, "forward", col_window=(c0, c1), x_window=S.x_window(cuts, r, N, 1, 1, 1), dtype=dtype)
    plan.set_lazy(f)
    mb = fd.P2P.loopback(ctx, W, r, 1 << 16)
    for b in range(W):
        if b == r:
            continue
        mb.fill(b, 0, gsum[b * slot:(b + 1) * slot])
        if b == r - 1:
            mb.fill(b, slot * 8, x_full[c0 - 2:c0].contiguous())
        if b == r + 1:
            mb.fill(b, slot * 8, x_full[c1:c1 + 2].contiguous())
    plan.set_p2p(mb)
    plan.set_halo(c0, c1, 2)
    x = torch.full_like(x_full, float("nan"))
    x[c0:c1] = x_full[c0:c1]
    o = torch.full((e1 - e0,), float("nan"), dtype=t, device="cuda")
    bad = 0
    for it in range(100):
        o.fill_(float("nan"))
        if it % 10 == 0:      # (fresh halo cells now and then: the first call of a rank is the one that brings them in)
            x[:c0] = float("nan"); x[c1:] = float("nan")
        plan.jacobian(f, x, [o])
        if not torch.equal(o, ref[e0:e1]):
            bad += 1
            d = (o != ref[e0:e1]) | torch.isnan(o)
            idx = torch.nonzero(d).flatten()
            if bad <= 3:
                print("rank", r, "it", it, "count", idx.numel(), "first", idx[:6].tolist(), "last", idx[-3:].tolist(), "of", o.numel(), "nan", int(torch.isnan(o).sum()), o[idx[:3]].tolist(), ref[e0:e1][idx[:3]].tolist())
    print("rank %d: %d / 100 calls differ" % (r, bad))
