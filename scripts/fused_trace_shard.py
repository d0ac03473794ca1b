"""Timeline of ONE rank's fused sharded launch (loop-back mailbox): device-clock marks.  FDJAC_TEST_SWITCHES=1 FDJAC_FUSED_TRACE=1."""
import os, sys
os.environ["FDJAC_TEST_SWITCHES"] = "1"
os.environ["FDJAC_FUSED_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P, sharded as S
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 7
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.cuda.set_stream(torch.cuda.Stream())
ctx = fd.Context(0)
colors = P.cyclic_colors(N, 3)
colptr, rowval = P.tridiag_csc(N)
pattern = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
f = fd.BuiltinF("tridiag", N, ctx=ctx)
x_full = torch.as_tensor(np.random.default_rng(4).random(N), device="cuda")
plan1 = fd.make_plan(pattern, pattern, colors, "forward", ctx=ctx)
for b in range(W):
    pptr, slot = plan1.eps_partials(x_full, b, W)
torch.cuda.synchronize()
class _Raw:
    __cuda_array_interface__ = {"shape": (W * slot,), "typestr": "<f8", "data": (pptr, False), "version": 2}
gsum = torch.as_tensor(_Raw(), device="cuda").clone()
cuts = S.eps_shard_cuts(N, W)
c0, c1 = int(cuts[r]), int(cuts[r + 1])
e0 = 0 if c0 <= 0 else 3 * c0 - 1
e1 = 3 * N - 2 if c1 >= N else 3 * c1 - 1
plan = fd.make_plan(pattern, pattern, colors, "forward", ctx=ctx, col_window=(c0, c1), x_window=S.x_window(cuts, r, N, 1, 1, 1))
plan.set_lazy(f)
mb = fd.P2P.loopback(ctx, W, r, 1 << 16)
mb.fill_fused(gsum[:512].contiguous(), x_full[c0 - 2:c0].contiguous() if r > 0 else None, x_full[c1:c1 + 2].contiguous() if r + 1 < W else None)
plan.set_p2p(mb)
plan.set_halo(c0, c1, 2)
x = torch.full_like(x_full, float("nan"))
x[c0:c1] = x_full[c0:c1]
out = torch.empty(e1 - e0, dtype=torch.float64, device="cuda")
call = plan.bind(f, x, [out])
rows, evs = [], []
for _ in range(8):
    for _ in range(20):
        call()
    plan.enable_timing(3)
    call()
    torch.cuda.synchronize()
    evs.append(plan.timing_samples("total")[-1] * 1e3)
    plan.enable_timing(0)
    rows.append(plan.fused_trace())
print("rank %d of %d, columns %d" % (r, W, c1 - c0))
print("%-22s" % "event_us(total call)", " ".join("%7.2f" % e for e in evs))
for k in rows[0]:
    print("%-22s" % k, " ".join("%7.2f" % q[k] for q in rows))
