#!/usr/bin/env python3
"""General patterns through the destination-table store (round 4) vs the hand-over path, same f!, same plan otherwise:
3-D 7-point stencil n^3 (built-in lap7, 7 colours) and a random band (built-in sparse family, greedy colouring).
    python scripts/pattern_probe_store.py [--n 8000000] [--only stencil|band]
Prints one markdown table row per (pattern, fdtype, path): whole call (median of individually timed calls, HIP events), the
storing / decompression launch, the other stages, plan build time, bit-identity of the two paths."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.pattern_probe import stencil7_csc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8_000_000)
    ap.add_argument("--only", choices=["stencil", "band"], default=None)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--store-only", action="store_true", help="skip the hand-over path (kernel experiments)")
    a = ap.parse_args()
    import torch
    import finitediff_jl_amd as fd
    dev = torch.device("cuda", 0)
    rows_out = []
    fd.Context.default().stream_copy_gbps(1 << 30, 4)      # (the PMC calibration kernel of the same run)

    def run(name, colptr, rowval, colors, M, N, f, fdtype):
        J = fd.SparseMatrixCSC(M, N, colptr, rowval, None)
        x = torch.rand(N, dtype=torch.float64, device=dev) + 0.1
        res = {}
        outs = {}
        paths = ("store",) if a.store_only else ("store", "handover")
        for path in paths:
            t0 = time.perf_counter()
            plan = fd.make_plan(J, J, colors, fdtype, store_csc=(path == "store"))
            torch.cuda.synchronize()
            build_ms = (time.perf_counter() - t0) * 1e3
            if path == "store":
                plan.set_lazy(f)
            out = torch.full((rowval.size,), float("nan"), dtype=torch.float64, device=dev)
            call = plan.bind(f, x, [out])
            for _ in range(2):
                call()
            torch.cuda.synchronize()
            plan.enable_timing(2)
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            st = {k: v["ms_sum"] / max(v["launches"], 1) * 1e3 for k, v in plan.timings().items()}
            nl = {k: v["launches"] // 5 for k, v in plan.timings().items()}
            plan.enable_timing(3)
            for _ in range(a.reps):
                call()
            torch.cuda.synchronize()
            tot = plan.timing_samples("total")
            plan.enable_timing(0)
            res[path] = (float(np.median(tot)) * 1e3, st, nl, build_ms, int(plan.info(fd.lib.INFO_LAZY_STORE)), int(plan.info(fd.lib.INFO_STORE_CSC)))
            outs[path] = out
            del plan, call
        same = a.store_only or bool(torch.equal(outs["store"].view(torch.int64), outs["handover"].view(torch.int64)))
        C = int(colors.max())
        nnz = rowval.size
        for path in paths:
            us, st, nl, build_ms, active, table = res[path]
            if path == "store":      # x, colours, the table (4 B rowptr per row + 5 B per entry), every value out
                model = N * 8 + N + (M + 1) * 4 + nnz * 5 + nnz * 8
            else:                    # perturbed points written + read, f! outputs written + read, codes / indices, values
                pts = 2 if fdtype == "central" else 1
                model = N * 8 + C * pts * N * 8 * 2 + (C * pts + (1 if fdtype == "forward" else 0)) * M * 8 * 2 + nnz * (8 + 5)
            rows_out.append("| %s | %s | %s | %d | %d | %d | %.1f | %.1f x%d | %.1f | %.1f x%d | %.1f | %.0f | %.0f | %.0f | %s |" % (
                name, fdtype, path + (" (table %d)" % table if path == "store" else ""), N, nnz, C, us, st["decompress"], nl["decompress"], st["eps"],
                st["f"], nl["f"], st["perturb"], model / 1e6, model / us / 1e3, build_ms, "yes" if same else "NO"))

    n3 = int(round(a.n ** (1 / 3)))
    if a.only in (None, "stencil"):
        cp, rv, col = stencil7_csc(n3, n3, n3)
        f = fd.BuiltinF("lap7", n3, n3, n3)
        for fdtype in ("forward", "central"):
            run("3-D 7-point %d^3" % n3, cp, rv, col, n3 ** 3, n3 ** 3, f, fdtype)
        del cp, rv, col, f
    if a.only in (None, "band"):
        N = min(a.n, 2_000_000)
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
        f = fd.BuiltinF.sparse(N, N, colptr, rowval)
        run("random band (+-300), 6 per column", colptr, rowval, colors, N, N, f, "forward")
    print("| pattern | fdtype | path | N | nnz | colours | whole call us (median) | store / decompress us x launches | eps us | f! us x launches | perturb us | model MB | GB/s (model / call) | plan build ms | same bits |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows_out:
        print(r)


if __name__ == "__main__":
    main()
