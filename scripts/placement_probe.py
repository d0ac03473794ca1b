#!/usr/bin/env python3
"""Does the graded kernel's time depend on where its buffers land?  Re-creates the plan / the output array several
times in one process (old ones kept alive, so every round gets fresh addresses) and prints the HIP-event time of the
fused difference+decompression kernel per round.  python scripts/placement_probe.py [--n 10000000] [--rounds 8]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10 ** 7)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--warm", type=int, default=5, help="untimed calls before the timed ones (clock ramp)")
    ap.add_argument("--timed", type=int, default=20)
    a = ap.parse_args()
    import torch
    import finitediff_jl_amd as fd
    from finitediff_jl_amd import patterns as P
    N = a.n
    dev = torch.device("cuda", 0)
    x = torch.as_tensor(np.random.default_rng(4).random(N), device=dev)
    colors = P.cyclic_colors(N, 3)
    colptr, rowval = P.tridiag_csc(N)
    pat = fd.SparseMatrixCSC(N, N, colptr, rowval, None)
    f = fd.BuiltinF("tridiag", N)
    keep = []
    print("| round | what changed | out address mod 2 MiB | decompress us | f! us | eps us |")
    print("|---|---|---|---|---|---|")
    plan = None
    for r in range(a.rounds):
        what = []
        if r % 2 == 0 or plan is None:
            plan = fd.make_plan(pat, pat, colors, "forward")
            plan.set_lazy(f)
            keep.append(plan)
            what.append("plan")
        pad = torch.empty((r * 4099 + 1) * 16, dtype=torch.uint8, device=dev)      # shifts the next allocation
        out = torch.empty(rowval.size, dtype=torch.float64, device=dev)
        keep += [pad, out]
        what.append("out")
        call = plan.bind(f, x, [out])
        for _ in range(a.warm):
            call()
        torch.cuda.synchronize()
        plan.enable_timing(2)
        for _ in range(a.timed):
            call()
        tm = plan.timings()
        plan.enable_timing(0)
        us = {k: v["ms_sum"] / max(v["launches"], 1) * 1e3 for k, v in tm.items()}
        print("| %d | %s | %d | %.1f | %.1f | %.1f |" % (r, "+".join(what), out.data_ptr() % (2 << 20), us["decompress"], us["f"], us["eps"]))


if __name__ == "__main__":
    main()
