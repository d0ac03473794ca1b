#!/usr/bin/env python3
"""Decompression-kernel time on patterns WITHOUT a built-in f!: 3-D 7-point stencil and a random banded pattern.
f! is a trivial torch launcher (fx <- x), only the fused difference + decompression stage is of interest.
    python scripts/pattern_probe.py [--n 8000000]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stencil7_csc(nx, ny, nz):
    N = nx * ny * nz
    k = np.arange(N, dtype=np.int64)
    i, j, l = k % nx, (k // nx) % ny, k // (nx * ny)
    has = np.stack([l > 0, j > 0, i > 0, np.ones(N, bool), i < nx - 1, j < ny - 1, l < nz - 1], axis=1)
    rows = np.stack([k - nx * ny, k - nx, k - 1, k, k + 1, k + nx, k + nx * ny], axis=1)
    cnt = has.sum(axis=1)
    colptr = np.empty(N + 1, np.int64)
    colptr[0] = 1
    np.cumsum(cnt, out=colptr[1:])
    colptr[1:] += 1
    return colptr, (rows[has] + 1).astype(np.int64), ((i + 2 * j + 3 * l) % 7 + 1).astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8_000_000)
    ap.add_argument("--only", choices=["stencil", "band"], default=None, help="one pattern only (PMC passes: one kernel population per run)")
    a = ap.parse_args()
    import torch
    import finitediff_jl_amd as fd
    from finitediff_jl_amd import patterns as P
    dev = torch.device("cuda", 0)
    out_rows = []

    def run(name, colptr, rowval, colors, N, fdtype="forward"):
        J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
        import time
        t0 = time.perf_counter()
        plan = fd.make_plan(J, J, colors, fdtype)
        print("plan build (%s): %.0f ms" % (name, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
        f = fd.TorchF(lambda fx, xx: fx.copy_(xx), N, N)
        x = torch.rand(N, dtype=torch.float64, device=dev)
        out = torch.empty(rowval.size, dtype=torch.float64, device=dev)
        for _ in range(2):
            plan.jacobian(f, x, [out], sync=False)
        torch.cuda.synchronize()
        plan.enable_timing(2)
        for _ in range(5):
            plan.jacobian(f, x, [out], sync=False)
        torch.cuda.synchronize()
        tm = plan.timings()
        us = tm["decompress"]["ms_sum"] / max(tm["decompress"]["launches"], 1) * 1e3
        print("stages (%s): " % name + ", ".join("%s %.1f us" % (k, v["ms_sum"] / max(v["launches"], 1) * 1e3) for k, v in tm.items()), file=sys.stderr)
        C = int(colors.max())
        kern = ("window2d" if plan.info(fd.lib.INFO_WINDOW2D) else "window" if plan.info(fd.lib.INFO_WINDOW)
                else "sorted" if plan.info(fd.lib.INFO_SORTED_GATHER) else "list")
        idx = {"window2d": 2, "window": 2, "sorted": 7, "list": 5}[kern]
        if kern.startswith("window"):   # every f! array streamed once (HBM), plus codes and values
            real = (C + 1) * N * 8 + rowval.size * (8 + idx)
        else:                           # gathers: one perturbed value per entry, fx once per row, index, value
            real = rowval.size * (8 + 8 + idx) + N * 8
        out_rows.append((name, N, rowval.size, C, kern, plan.info(fd.lib.INFO_WIN_OVERREAD_X100) / 100.0, us, real / 1e6,
                         real / us / 1e3))

    fd.Context.default().stream_copy_gbps(1 << 30, 4)     # k_stream_copy: the PMC calibration kernel of the same run (1 GiB read + written)
    n3 = int(round(a.n ** (1 / 3)))
    if a.only in (None, "stencil"):
        cp, rv, col = stencil7_csc(n3, n3, n3)
        run("3-D 7-point stencil %d^3" % n3, cp, rv, col, n3 ** 3)
        del cp, rv, col
    if a.only == "stencil":
        return report(out_rows)
    # random banded: 6 entries per column at random rows within +-300 of the diagonal, greedy colouring
    N = min(a.n, 2_000_000)
    rng = np.random.default_rng(1)
    offs = np.sort(rng.integers(-300, 301, size=(N, 6)), axis=1)
    rows = np.clip(np.arange(N)[:, None] + offs, 0, N - 1)
    rows = np.sort(rows, axis=1)
    keep = np.ones_like(rows, bool)
    keep[:, 1:] = rows[:, 1:] != rows[:, :-1]
    cnt = keep.sum(axis=1)
    colptr = np.empty(N + 1, np.int64)
    colptr[0] = 1
    np.cumsum(cnt, out=colptr[1:])
    colptr[1:] += 1
    rowval = (rows[keep] + 1).astype(np.int64)
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    colors = fd.matrix_colors(J)
    run("random band (+-300), 6 per column", colptr, rowval, colors, N)
    report(out_rows)


def report(out_rows):
    print("| pattern | N | nnz | colours | kernel | over-read | diff+decompress us | real MB | GB/s (real traffic) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in out_rows:
        print("| %s | %d | %d | %d | %s | %.2f | %.1f | %.0f | %.0f |" % r)


if __name__ == "__main__":
    main()
