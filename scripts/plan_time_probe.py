#!/usr/bin/env python3
"""Where a plan's build time goes (FDJAC_PLAN_TIMING=1: the constructors' sections on stderr), for the two patterns the round-3 review
quoted: the BlockBandedMatrix configuration (10^4 blocks of 32 x 32, complex step) and the 3-D 7-point stencil 200^3.
    python scripts/plan_time_probe.py [--n 200]"""
import argparse
import os
os.environ.setdefault("FDJAC_TEST_SWITCHES", "1")   # (the library honours its variant switches only on request)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.pattern_probe import stencil7_csc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    a = ap.parse_args()
    import torch
    import finitediff_jl_amd as fd
    from finitediff_jl_amd import patterns as P
    fd.Context.default().stream_copy_gbps(1 << 28, 2)

    def timed(name, make, reps=3):
        for i in range(reps):
            os.environ["FDJAC_PLAN_TIMING"] = "1" if i == reps - 1 else "0"
            t = time.perf_counter()
            plan = make()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) * 1e3
            print("%-64s build %d: %9.2f ms" % (name, i, dt), flush=True)
            sys.stderr.flush()
            del plan
        os.environ["FDJAC_PLAN_TIMING"] = "0"

    lay = P.BlockBandedLayout(np.full(10 ** 4, 32), 1, 1)
    colors = lay.colors()
    Jb = fd.BlockBandedMatrix(None, lay)
    timed("BlockBanded 1e4 x 32^2, complex step", lambda: fd.make_plan(Jb, Jb, colors, "complex"))
    n = a.n
    colptr, rowval, col7 = stencil7_csc(n, n, n)
    N = n ** 3
    J = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    for store in (True, False):
        timed("7-point %d^3 host pattern (Int64), store_csc=%d" % (n, store), lambda: fd.make_plan(J, J, col7, "forward", store_csc=store))
    d = [torch.as_tensor(v, device="cuda") for v in (colptr, rowval, col7)]
    torch.cuda.synchronize()
    for store in (True, False):
        timed("7-point %d^3 device pattern (Int64), store_csc=%d" % (n, store),
              lambda: fd.make_plan_csc_device(N, N, d[0], d[1], d[2], "forward", store_csc=store))


if __name__ == "__main__":
    main()
