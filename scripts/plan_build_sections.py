import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P
N = 10 ** 7
cp, rv = P.tridiag_csc(N); colors = P.cyclic_colors(N, 3)
d = [torch.as_tensor(a, device="cuda") for a in (cp, rv, colors)]
torch.cuda.synchronize()
fd.make_plan_csc_device(N, N, d[0], d[1], d[2], "forward")
os.environ["FDJAC_PLAN_TIMING"] = "1"
t = time.perf_counter(); pl = fd.make_plan_csc_device(N, N, d[0], d[1], d[2], "forward"); print("total ms", (time.perf_counter() - t) * 1e3)
t = time.perf_counter(); del pl; print("destroy ms", (time.perf_counter() - t) * 1e3)
