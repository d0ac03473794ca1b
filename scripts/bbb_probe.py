"""Time the BandedBlockBandedMatrix path (structural plan, k_decompress_bbb) on the reference's 2-D fixture at a size that fills the GPU:
ny blocks of nx columns, bandwidths (1, 1) / (1, 1), 9 colours.  usage: bbb_probe.py [nx ny]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P
nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40, 50000)
lay = P.BandedBlockBandedLayout(np.full(ny, nx), 1, 1, 1, 1)
N = lay.N
colors = lay.colors()
x = torch.as_tensor(np.random.default_rng(0).random(N), device="cuda")
f = fd.BuiltinF("clamp5", nx, ny)
for fdtype in ("forward", "central"):
    J = fd.BandedBlockBandedMatrix(None, lay)
    plan = fd.make_plan(J, J, colors, fdtype)
    out = torch.empty(lay.data_len, dtype=torch.float64, device="cuda")
    for _ in range(3): plan.jacobian(f, x, [out])
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t = time.perf_counter()
        plan.jacobian(f, x, [out]); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    print("bbb %dx%d N=%d slots=%d colours=%d %s: %.1f us per Jacobian (median of 10), checksum %.17g" % (nx, ny, N, lay.data_len, int(colors.max()), fdtype, np.median(ts) * 1e6, float(out.double().sum())))
