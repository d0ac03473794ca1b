#!/usr/bin/env python3
"""The random band (2*10^6 columns, +-300, 6 per column, greedy colouring) through the row-wise store: the built-in sparse family
(k_f_sparse_store_rows) against a SEPARABLE USER functor compiled from its term (fd_f_compile_terms -> fd_csc_store_rows), and the
same functor through the column store.  Prints a markdown table (whole call, median of individually timed calls; storing launch).
    python scripts/terms_probe.py [--n 2000000] [--reps 20] [--only builtin|terms|cols]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TERMS = """
struct SparseTerms {
    template <class T> __device__ T term(long long r, long long j, T v) const
    {
        return ((real_t)1 + (real_t)0.125 * (real_t)(int)((r + 3 * j) & 7)) * (v + ((real_t)0.25 * v) * v);
    }
};
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--fdtype", default="forward")
    a = ap.parse_args()
    import torch
    import finitediff_jl_amd as fd
    dev = torch.device("cuda", 0)
    fd.Context.default().stream_copy_gbps(1 << 30, 4)      # (the PMC calibration kernel of the same run)
    N = a.n
    rng = np.random.default_rng(1)
    offs = np.sort(rng.integers(-300, 301, size=(N, 6)), axis=1)
    rows = np.arange(N)[:, None] + offs
    keep = (rows >= 0) & (rows < N)
    keep[:, 1:] &= rows[:, 1:] != rows[:, :-1]
    cnt = keep.sum(axis=1)
    colptr = np.empty(N + 1, np.int64)
    colptr[0] = 1
    np.cumsum(cnt, out=colptr[1:])
    colptr[1:] += 1
    rowval = (rows[keep] + 1).astype(np.int64)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    colors = fd.matrix_colors(J)
    x = torch.rand(N, dtype=torch.float64, device=dev) + 0.1
    outs = {}
    print("| route | N | nnz | colours | whole call us (median) | storing launch us | eps us | row-wise launches |")
    print("|---|---|---|---|---|---|---|---|")
    for route in ("builtin", "builtin_own", "terms", "cols"):
        if a.only and a.only != route:
            continue
        plan = fd.make_plan(J, J, colors, a.fdtype, store_csc=True, store_rows=(route in ("terms", "builtin")))
        if route in ("builtin", "builtin_own"):      # (builtin_own: a plan without row lists -- the family's own kernel, k_f_sparse_store_rows)
            f = fd.BuiltinF.sparse(N, N, colptr, rowval)
        else:
            src = fd.make_plan(J, J, colors, a.fdtype, store_rows=True) if route == "cols" else plan      # (cols: lists of ANOTHER plan -> column store)
            f = fd.JitTerms(TERMS, "SparseTerms", src)
        plan.set_lazy(f)
        out = torch.full((rowval.size,), float("nan"), dtype=torch.float64, device=dev)
        call = plan.bind(f, x, [out])
        for _ in range(4):
            call()
            torch.cuda.synchronize()
        plan.enable_timing(2)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        st = {k: v["ms_sum"] / max(v["launches"], 1) * 1e3 for k, v in plan.timings().items()}
        plan.enable_timing(3)
        for _ in range(a.reps):
            call()
        torch.cuda.synchronize()
        tot = plan.timing_samples("total")
        plan.enable_timing(0)
        nrow = f.row_stores() if route.startswith("builtin") else f.row_stores
        outs[route] = out
        print("| %s | %d | %d | %d | %.1f | %.1f | %.1f | %d |" % (route, N, rowval.size, int(colors.max()), float(np.median(tot)) * 1e3, st["decompress"], st["eps"], nrow))
    ks = list(outs)
    for k in ks[1:]:
        print("same bits %s == %s: %s" % (ks[0], k, bool(torch.equal(outs[ks[0]].view(torch.int64), outs[k].view(torch.int64)))))


if __name__ == "__main__":
    main()
