#!/usr/bin/env python3
"""A runtime-compiled row functor on BlockBandedMatrix data (block-tridiagonal, dense bs x bs blocks): the functor's own storing launch
(fd_colrange_store_cols: one wavefront per column) against the same functor as an opaque f! (perturb, batched rows, decompression),
and the built-in block-coupled family's storing kernel for scale.  python scripts/colrange_probe.py [--nb 2000] [--bs 32]"""
import argparse
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=2000)
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    import torch
    import finitediff_jl_amd as fd
    from finitediff_jl_amd import patterns as P
    from test_gpu_jit import BLOCK_COUPLED
    nb, bs = a.nb, a.bs
    N = nb * bs
    lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
    colors = lay.colors()
    x = torch.rand(N, dtype=torch.float64, device="cuda") - 0.3
    Jb = fd.BlockBandedMatrix(None, lay)
    fj = fd.JitF(BLOCK_COUPLED, "BlockCoupled", N, N, params=struct.pack("qq", nb, bs))
    fb = fd.BuiltinF("blockcoupled", nb, bs)
    print("| fdtype | route | N | stored values | whole call us (median) |")
    print("|---|---|---|---|---|")
    for fdtype in ("complex", "forward"):
        outs = {}
        for route in ("compiled functor, its own storing launch", "compiled functor, opaque f!", "built-in family"):
            f = fb if route.startswith("built-in") else fj
            plan = fd.make_plan(Jb, Jb, colors, fdtype)
            if "opaque" not in route:
                plan.set_lazy(f)
            out = torch.full((plan.out_len(0),), float("nan"), dtype=torch.float64, device="cuda")
            call = plan.bind(f, x, [out])
            for _ in range(2):
                call()
            torch.cuda.synchronize()
            plan.enable_timing(3)
            for _ in range(a.reps):
                call()
            torch.cuda.synchronize()
            tot = plan.timing_samples("total")
            plan.enable_timing(0)
            outs[route] = out
            print("| %s | %s | %d | %d | %.1f |" % (fdtype, route, N, out.numel(), float(np.median(tot)) * 1e3))
        k = list(outs)
        print("same bits, storing launch == opaque: %s" % bool(torch.equal(outs[k[0]].view(torch.int64), outs[k[1]].view(torch.int64))))


if __name__ == "__main__":
    main()
