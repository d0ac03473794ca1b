"""Time fd_tridiag_solve_async at N = 10^7 (both layouts); run under rocprofv3 --kernel-trace --stats for per-kernel times."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finitediff_jl_amd as fd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 7
rng = np.random.default_rng(0)
dl, d, du, b = (torch.as_tensor(rng.random(n) - 0.5, device="cuda") for n in (N - 1, N, N - 1, N))
nz = torch.empty(3 * N - 2, dtype=torch.float64, device="cuda")
j = torch.arange(N, device="cuda")
nz[torch.where(j > 0, 3 * j, 0)] = d; nz[3 * j[1:] - 1] = du; nz[3 * j[:-1] + 1] = dl
y = torch.empty(N, dtype=torch.float64, device="cuda")
for layout, J in (("diagonals", fd.Tridiagonal(dl, d, du)), ("csc", [nz])):
    s = fd.TridiagSolver(N, layout)
    for _ in range(3): s.solve(J, b, y, 3.0, -0.7)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): s.solve(J, b, y, 3.0, -0.7)
    torch.cuda.synchronize(); print(layout, "%.1f us per solve" % ((time.perf_counter() - t) / 10 * 1e6))
