"""Decompression time of the storage-order gather (k_decompress_list) vs the colour-sorted gather (k_decompress_sorted, f(x) in LDS)
on a random rectangular pattern (1.2e6 x 1.0e6, ~6 per column within +-3000 rows + 3 % far entries), forward differences."""
import os, sys, time
os.environ.setdefault("FDJAC_TEST_SWITCHES", "1")   # (the library honours its variant switches only on request)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finitediff_jl_amd as fd

M, N = 1_200_000, 1_000_000
rng = np.random.default_rng(1234)
centre = (np.arange(N) * (M / N)).astype(np.int64)
rws = centre[:, None] + rng.integers(-3000, 3001, size=(N, 6))
far = rng.random(N) < 0.03
rws[far, 0] = rng.integers(0, M, size=int(far.sum()))
rws = np.abs(rws)
rws = np.sort(np.where(rws > M - 1, 2 * (M - 1) - rws, rws), axis=1)
keep = np.ones_like(rws, bool); keep[:, 1:] = rws[:, 1:] != rws[:, :-1]
colptr = np.empty(N + 1, np.int64); colptr[0] = 1
np.cumsum(keep.sum(axis=1), out=colptr[1:]); colptr[1:] += 1
rowval = (rws[keep] + 1).astype(np.int64)
J = fd.SparseMatrixCSC(M, N, colptr, rowval)
colors = fd.matrix_colors(J)
x = torch.rand(N, dtype=torch.float64, device="cuda")
f = fd.TorchF(lambda fv, xx: fv.copy_(torch.cat([xx, xx[: M - N]])), M, N)
for fdtype in ("forward", "central"):
    for env in ({}, {"FDJAC_SORTED": "1"}):
        for k in ("FDJAC_SORTED",):
            os.environ.pop(k, None)
        os.environ.update(env)
        plan = fd.make_plan(J, J, colors, fdtype)
        out = torch.empty(rowval.size, dtype=torch.float64, device="cuda")
        for _ in range(2):
            plan.jacobian(f, x, [out], sync=False)
        torch.cuda.synchronize()
        plan.enable_timing(2)
        for _ in range(5):
            plan.jacobian(f, x, [out], sync=False)
        torch.cuda.synchronize()
        tm = plan.timings()
        us = tm["decompress"]["ms_sum"] / max(tm["decompress"]["launches"], 1) * 1e3
        kern = "sorted" if plan.info(fd.lib.INFO_SORTED_GATHER) else "window" if plan.info(fd.lib.INFO_WINDOW) else "list"
        print("%-8s %-40s kernel %-7s decompress %.1f us (nnz %d, %d colours)" % (fdtype, env, kern, us, rowval.size, int(colors.max())))
