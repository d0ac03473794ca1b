"""Where a fused launch's time goes: device-clock marks (fd_plan_fused_trace).  FDJAC_TEST_SWITCHES=1 FDJAC_FUSED_TRACE=1."""
import os, sys, json
os.environ["FDJAC_TEST_SWITCHES"] = "1"
os.environ["FDJAC_FUSED_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 6
torch.cuda.set_stream(torch.cuda.Stream())
colors = P.cyclic_colors(N, 3)
colptr, rowval = P.tridiag_csc(N)
J = fd.SparseMatrixCSC(N, N, colptr, rowval, torch.zeros(rowval.size, dtype=torch.float64, device="cuda"))
f = fd.BuiltinF("tridiag", N)
plan = fd.make_plan(J, J, colors, "forward")
plan.set_lazy(f)
x = torch.as_tensor(np.random.default_rng(2).random(N), device="cuda")
call = plan.bind(f, x, [J.nzval])
for _ in range(10):
    call()
torch.cuda.synchronize()
rows = []
evs = []
for _ in range(8):
    for _ in range(20):
        call()
    plan.enable_timing(3)
    call()
    torch.cuda.synchronize()
    evs.append(plan.timing_samples("total")[-1] * 1e3)
    plan.enable_timing(0)
    rows.append(plan.fused_trace())
print("%-22s" % "event_us(total call)", " ".join("%7.2f" % e for e in evs))
for k in rows[0]:
    print("%-22s" % k, " ".join("%7.2f" % r[k] for r in rows))
