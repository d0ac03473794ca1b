#!/usr/bin/env python3
"""Per-kernel table on one MI355X: every storage type of the path at benchmark scale, all three fdtypes.

For each case: HIP-event time of the fused difference+decompression kernel (fd_plan_enable_timing(2)),
its SURVEY 8(d) algorithmic bytes, the implied GB/s, and the other stages.  Output: markdown on stdout.
    python scripts/kernel_sweep.py [--n 10000000] [--reps 10]
"""
import argparse
import os
os.environ.setdefault("FDJAC_TEST_SWITCHES", "1")   # (the library honours its variant switches only on request)
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10 ** 7)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    import torch
    import finitediff_jl_amd as fd
    from finitediff_jl_amd import patterns as P
    dev = torch.device("cuda", 0)
    N = a.n
    rows = []

    def run(name, plan, f, x, outs, alg_bytes, lazy):
        if lazy and getattr(f, "lazy_fn", None) is not None:
            plan.set_lazy(f)
        for _ in range(3):
            plan.jacobian(f, x, outs, sync=False)
        torch.cuda.synchronize()
        plan.enable_timing(2)
        for _ in range(a.reps):
            plan.jacobian(f, x, outs, sync=False)
        torch.cuda.synchronize()
        tm = plan.timings()
        plan.enable_timing(0)
        ms = {k: v["ms_sum"] / max(v["launches"], 1) for k, v in tm.items()}
        gbs = alg_bytes / (ms["decompress"] * 1e-3) / 1e9 if ms["decompress"] > 0 else 0.0
        kern = "window" if plan.info(fd.lib.INFO_WINDOW) else "sorted" if plan.info(fd.lib.INFO_SORTED_GATHER) else "-"
        rows.append((name, kern, ms["eps"] * 1e3, ms["perturb"] * 1e3, ms["f"] * 1e3, ms["decompress"] * 1e3,
                     ms["total"] * 1e3, alg_bytes / 1e6, gbs, 100 * gbs / 8000.0))

    x = torch.as_tensor(np.random.default_rng(4).random(N), device=dev)
    colors3 = P.cyclic_colors(N, 3)
    want = set(a.only.split(",")) if a.only else None

    def on(tag):
        return want is None or tag in want

    for fdtype in ("forward", "central", "complex"):
        pts = 2 if fdtype == "central" else 1
        sz = 16 if fdtype == "complex" else 8
        fxb = 3 * pts * N * sz + (N * 8 if fdtype == "forward" else 0)     # f! values streamed once per colour (+ fx)
        if on("csc"):
            colptr, rowval = P.tridiag_csc(N)
            pat = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
            plan = fd.make_plan(pat, pat, colors3, fdtype)
            out = torch.empty(rowval.size, dtype=torch.float64, device=dev)
            alg = (2 * 3 * N * 8 if fdtype != "complex" else 3 * N * 16) + rowval.size * 12 + 4 * (N + 1) + N
            run("tridiagonal CSC " + fdtype, plan, fd.BuiltinF("tridiag", N), x, [out], alg, True)
            del plan, out, pat, colptr, rowval
        if on("tridiag"):
            J = fd.Tridiagonal(torch.empty(N - 1, dtype=torch.float64, device=dev), torch.empty(N, dtype=torch.float64, device=dev),
                               torch.empty(N - 1, dtype=torch.float64, device=dev))
            plan = fd.make_plan(J, None, colors3, fdtype)
            alg = (2 * 3 * N * 8 if fdtype != "complex" else 3 * N * 16) + (3 * N - 2) * 8 + N
            run("Tridiagonal " + fdtype, plan, fd.BuiltinF("tridiag", N), x, [J.dl, J.d, J.du], alg, True)
            del plan, J
        if on("banded"):
            data = torch.empty(3 * N, dtype=torch.float64, device=dev)
            J = fd.BandedMatrix(data.view(N, 3).t(), N, 1, 1)
            plan = fd.make_plan(J, None, colors3, fdtype)
            alg = (2 * 3 * N * 8 if fdtype != "complex" else 3 * N * 16) + 3 * N * 8 + N
            run("BandedMatrix(1,1) " + fdtype, plan, fd.BuiltinF("tridiag", N), x, [data], alg, True)
            del plan, J, data
    if on("lap5"):
        nx, ny = 4000, N // 4000
        n5 = nx * ny
        x5 = torch.as_tensor(np.random.default_rng(3).random(n5), device=dev)
        colptr, rowval = P.lap5_csc(nx, ny)
        nnz = rowval.size
        pat = fd.SparseMatrixCSC(n5, n5, colptr, rowval, None)
        del rowval
        for fdtype in ("forward", "central", "complex"):
            plan = fd.make_plan(pat, pat, P.lap5_colors(nx, ny), fdtype)
            out = torch.empty(nnz, dtype=torch.float64, device=dev)
            alg = (2 * 5 * n5 * 8 if fdtype != "complex" else 5 * n5 * 16) + nnz * 12 + 4 * (n5 + 1) + n5
            run("5-point CSC %dx%d %s" % (nx, ny, fdtype), plan, fd.BuiltinF("lap5", nx, ny), x5, [out], alg, True)
            del plan, out
    if on("bb"):
        nb, bs = max(N // 1000, 3), 32
        lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
        xb = torch.as_tensor(np.random.default_rng(5).random(nb * bs), device=dev)
        for fdtype in ("forward", "complex"):
            Jbb = fd.BlockBandedMatrix(None, lay)
            plan = fd.make_plan(Jbb, Jbb, lay.colors(), fdtype)
            out = torch.empty(lay.data_len, dtype=torch.float64, device=dev)
            C = 96
            alg = (2 * C * nb * bs * 8 if fdtype == "forward" else C * nb * bs * 16) + lay.data_len * 8
            run("BlockBanded %dx32x32 %s" % (nb, fdtype), plan, fd.BuiltinF("blockcoupled", nb, bs), xb, [out], alg, True)
            del plan, out

    if on("jvp"):
        import time
        v = torch.as_tensor(np.random.default_rng(6).random(N) - 0.5, device=dev)
        for fdtype in ("forward", "central"):
            out = torch.empty(N, dtype=torch.float64, device=dev)
            cache = fd.JVPCache(x, fdtype)
            f = fd.BuiltinF("tridiag", N)
            for _ in range(3):
                fd.finite_difference_jvp_b(out, f, x, v, cache, sync=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                fd.finite_difference_jvp_b(out, f, x, v, cache, sync=False)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / a.reps * 1e6
            # materialised: dot (x, v) 16 B + points (x, v -> X) 24/32 B + two f! evaluations 2 x 16 B + difference 24 B
            # per state; with the lazy-point launcher (built-in families): dot 16 + f! (x, v -> two outputs) 32 + difference 24
            # launcher that writes the finished quotient (FD_LAZY_JVP_CAP_QUOTIENT): dot 16 + f! (x, v -> jvp) 24
            lazy = cache.lazy and f.lazy_jvp_fn is not None
            quot = lazy and cache.quotient and f.lazy_jvp_caps and os.environ.get("FDJAC_LAZY_DIFF", "1") != "0"
            mb = ((16 + 24) if quot else (16 + 32 + 24) if lazy else (16 + (24 if fdtype == "forward" else 32) + 32 + 24)) * N / 1e6
            rows.append(("finite_difference_jvp! " + fdtype + (" (lazy f! writes the quotient," if quot else " (lazy f!," if lazy else " (") + " whole call, wall)", "-", 0.0, 0.0, 0.0, 0.0, us, mb,
                         mb / us * 1e3, 100 * (mb / us * 1e3) / 8000.0))

    print("| case | kernel variant | eps us | perturb us | f! us | diff+decompress us | whole call us | algorithmic MB | GB/s | % of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %s | %.1f | %.1f | %.1f | %.1f | %.1f | %.0f | %.0f | %.1f |" % r)


if __name__ == "__main__":
    main()
