"""Timing of the block-tridiagonal consumer solve at BASELINE's config 5: 10^4 blocks of 32 x 32."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 4
res = {}
for b in (32, 16, 8):
    lay = P.BlockBandedLayout([b] * nb, 1, 1)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    data = torch.randn(lay.data_len, generator=g, device="cuda", dtype=torch.float64)
    rhs = torch.randn(nb * b, generator=g, device="cuda", dtype=torch.float64)
    y = torch.empty(nb * b, dtype=torch.float64, device="cuda")
    s = fd.BlockTridiagSolver(nb, b)
    gamma = 0.2 / (3 * b)
    for _ in range(3):
        s.solve(data, rhs, y, alpha=1.0, beta=-gamma)
    assert s.status() == 0
    ctx = fd.Context.default()
    ctx.synchronize(); t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        s.solve(data, rhs, y, alpha=1.0, beta=-gamma)
    ctx.synchronize(); t1 = time.perf_counter()
    res["b=%d" % b] = {"ms": (t1 - t0) / reps * 1e3, "matrix_MB": lay.data_len * 8 / 1e6}
print(json.dumps(res))
