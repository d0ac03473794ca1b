#!/usr/bin/env python3
"""Per-call latency of one coloured Jacobian at small N (the regime of an implicit ODE step): wall time of
fd_jacobian (blocking) and of a stream of fd_jacobian_async calls, built-in lazy f! vs materialised points."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import finitediff_jl_amd as fd  # noqa: E402
from finitediff_jl_amd import patterns as P  # noqa: E402

print("| N | mode | blocking call us | async stream us/call |")
print("|---|---|---|---|")
for N in (100, 1000, 10_000, 100_000, 1_000_000):
    cp, rv = P.tridiag_csc(N)
    J = fd.SparseMatrixCSC(N, N, cp, rv)
    x = torch.rand(N, dtype=torch.float64, device="cuda")
    out = torch.empty(rv.size, dtype=torch.float64, device="cuda")
    for mode in ("lazy", "materialised"):
        plan = fd.make_plan(J, J, P.cyclic_colors(N, 3), "forward")
        f = fd.BuiltinF("tridiag", N)
        if mode == "lazy":
            plan.set_lazy(f)
        for _ in range(20):
            plan.jacobian(f, x, [out])
        reps = 300
        t0 = time.perf_counter()
        for _ in range(reps):
            plan.jacobian(f, x, [out])
        tb = (time.perf_counter() - t0) / reps * 1e6
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            plan.jacobian(f, x, [out], sync=False)
        torch.cuda.synchronize()
        ta = (time.perf_counter() - t0) / reps * 1e6
        print("| %d | %s | %.1f | %.1f |" % (N, mode, tb, ta))
