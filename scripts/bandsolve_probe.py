"""Timing of the banded consumer solve (fd_banded_solve_async) against the tridiagonal one: N = 10^7."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import finitediff_jl_amd as fd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 7
res = {}
for (l, u) in [(1, 1), (2, 2), (3, 3), (4, 4)]:
    w = l + u + 1
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    data = torch.randn((N, w), generator=g, device="cuda", dtype=torch.float64)
    b = torch.randn(N, generator=g, device="cuda", dtype=torch.float64)
    y = torch.empty(N, dtype=torch.float64, device="cuda")
    s = fd.BandedSolver(N, l, u)
    for _ in range(3):
        s.solve(data.view(-1), b, y, alpha=1.0, beta=-0.02)
    assert s.status() == 0
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx = fd.Context.default()
    reps = 10
    import time
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        s.solve(data.view(-1), b, y, alpha=1.0, beta=-0.02)
    ctx.synchronize(); t1 = time.perf_counter()
    res["%d,%d" % (l, u)] = {"ms": (t1 - t0) / reps * 1e3, "bytes_in": (w + 1) * 8 * N, "nlev": None}
    del s, data
print(json.dumps(res))
