"""The deferred content check (fd_plan_matches_async -> k_fingerprint3_check) at N = 10^7 with a device-resident Int32 pattern: the drop-in
call with pattern_check = content_async, 30 times; run under rocprofv3 --kernel-trace --stats for the kernel's time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 7
dev = torch.device("cuda", 0)
cp, rv = P.tridiag_csc(N)
colors = P.cyclic_colors(N, 3)
x = torch.as_tensor(np.random.default_rng(0).random(N), device=dev)
f = fd.BuiltinF("tridiag_nl", N)
out = torch.empty(rv.size, dtype=torch.float64, device=dev)
J = fd.DevicePatternCSC(N, N, torch.as_tensor(cp.astype(np.int32), device=dev), torch.as_tensor(rv.astype(np.int32), device=dev), out)
cv = torch.as_tensor(np.asarray(colors).astype(np.int32), device=dev)
for mode in ("identity", "content_async"):
    cache = fd.JacobianCache(x, "forward", colorvec=cv, sparsity=J)
    cache.pattern_check = mode
    for _ in range(3):
        fd.finite_difference_jacobian_b(J, f, x, cache)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30):
        fd.finite_difference_jacobian_b(J, f, x, cache)
    torch.cuda.synchronize(); print(mode, "%.1f us per call" % ((time.perf_counter() - t) / 30 * 1e6))
