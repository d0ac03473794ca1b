"""Per-call cost of the bound Jacobian call at N = 10^6: host enqueue time vs steady-state wall time per call, fused / unfused."""
import os, sys, time, json
os.environ["FDJAC_TEST_SWITCHES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 6
torch.cuda.set_stream(torch.cuda.Stream())
colors = P.cyclic_colors(N, 3)
colptr, rowval = P.tridiag_csc(N)
x = torch.as_tensor(np.random.default_rng(2).random(N), device="cuda")
for fused in (False, True):
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, torch.zeros(rowval.size, dtype=torch.float64, device="cuda"))
    f = fd.BuiltinF("tridiag", N)
    plan = fd.make_plan(J, J, colors, "forward")
    plan.set_lazy(f, fused=fused)
    call = plan.bind(f, x, [J.nzval])
    for _ in range(50):
        call()
    torch.cuda.synchronize()
    for timing in (0, 1):
        plan.enable_timing(timing)
        for reps in (20, 2000):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(json.dumps({"fused": fused, "timing": timing, "reps": reps, "host_enqueue_us": (t1 - t0) / reps * 1e6, "wall_us": (t2 - t0) / reps * 1e6}))
        plan.enable_timing(0)
