import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P
N = 10**7
ctx = fd.Context.default()
cp, rv = P.tridiag_csc(N)
pat = fd.SparseMatrixCSC(N, N, cp, rv, None)
colors = P.cyclic_colors(N, 3)
plan = fd.make_plan(pat, pat, colors, "forward", ctx=ctx)
f = fd.BuiltinF("tridiag", N, ctx=ctx)
plan.set_lazy(f)
x = torch.as_tensor(np.random.default_rng(4).random(N), device="cuda")
out = torch.empty(rv.size, dtype=torch.float64, device="cuda")
call = plan.bind(f, x, [out])
for _ in range(5): call()
torch.cuda.synchronize()
plan.enable_timing(2)
for _ in range(50): call()
torch.cuda.synchronize()
tm = plan.timings()
print({k: v["ms_sum"] / max(v["launches"], 1) for k, v in tm.items()})
plan.enable_timing(3)
for _ in range(50): call()
torch.cuda.synchronize()
print("median total", float(np.median(plan.timing_samples("total"))))
