"""Host cost of building a plan (pattern ingestion, colour lookup, tile descriptors, uploads) at BASELINE sizes."""
import os
os.environ.setdefault("FDJAC_TEST_SWITCHES", "1")   # (the library honours its variant switches only on request)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finitediff_jl_amd as fd  # noqa: E402
from finitediff_jl_amd import patterns as P  # noqa: E402


def timed(name, make):
    t = time.perf_counter()
    plan = make()
    dt = time.perf_counter() - t
    print("%-44s plan build %8.1f ms" % (name, dt * 1e3), flush=True)
    return plan


for N in (10 ** 6, 10 ** 7):
    cp, rv = P.tridiag_csc(N)
    colors = P.cyclic_colors(N, 3)
    J = fd.SparseMatrixCSC(N, N, cp, rv, None)
    timed("tridiagonal CSC N=%d" % N, lambda: fd.make_plan(J, J, colors, "forward"))
    timed("tridiagonal CSC N=%d (again)" % N, lambda: fd.make_plan(J, J, colors, "forward"))
    os.environ["FDJAC_PLAN_DEVICE"] = "0"
    timed("tridiagonal CSC N=%d, host builder" % N, lambda: fd.make_plan(J, J, colors, "forward"))
    del os.environ["FDJAC_PLAN_DEVICE"]
    import torch
    dcp, drv, dcv = torch.as_tensor(cp, device="cuda"), torch.as_tensor(rv, device="cuda"), torch.as_tensor(colors, device="cuda")
    torch.cuda.synchronize()
    timed("tridiagonal CSC N=%d, pattern on the device (Int64)" % N, lambda: fd.make_plan_csc_device(N, N, dcp, drv, dcv, "forward"))
    timed("tridiagonal CSC N=%d, pattern on the device (again)" % N, lambda: fd.make_plan_csc_device(N, N, dcp, drv, dcv, "forward"))
    d32 = [torch.as_tensor((a - 1).astype(np.int32), device="cuda") for a in (cp, rv)] + [torch.as_tensor(colors.astype(np.int32), device="cuda")]
    torch.cuda.synchronize()
    timed("tridiagonal CSC N=%d, pattern on the device (Int32)" % N, lambda: fd.make_plan_csc_device(N, N, d32[0], d32[1], d32[2], "forward", idx_base=0))
    timed("Tridiagonal N=%d" % N, lambda: fd.make_plan(fd.Tridiagonal(None, np.empty(N), None), None, colors, "forward"))
    timed("BandedMatrix (1,1) N=%d" % N, lambda: fd.make_plan(fd.BandedMatrix(None, N, 1, 1), None, colors, "forward"))
nx, ny = 4000, 2500
cp, rv = P.lap5_csc(nx, ny)
colors = P.lap5_colors(nx, ny)
J = fd.SparseMatrixCSC(nx * ny, nx * ny, cp, rv, None)
timed("5-point CSC 4000x2500", lambda: fd.make_plan(J, J, colors, "central"))
timed("5-point CSC 4000x2500 (again)", lambda: fd.make_plan(J, J, colors, "central"))
os.environ["FDJAC_PLAN_DEVICE"] = "0"
timed("5-point CSC 4000x2500, host builder", lambda: fd.make_plan(J, J, colors, "central"))
del os.environ["FDJAC_PLAN_DEVICE"]
dcp, drv, dcv = torch.as_tensor(cp, device="cuda"), torch.as_tensor(rv, device="cuda"), torch.as_tensor(colors, device="cuda")
torch.cuda.synchronize()
timed("5-point CSC 4000x2500, pattern on the device (Int64)", lambda: fd.make_plan_csc_device(nx * ny, nx * ny, dcp, drv, dcv, "central"))
os.environ["FDJAC_PLAN_TIMING"] = "1"
timed("5-point CSC 4000x2500, pattern on the device (again)", lambda: fd.make_plan_csc_device(nx * ny, nx * ny, dcp, drv, dcv, "central"))
del os.environ["FDJAC_PLAN_TIMING"]
del dcp, drv, dcv
lay = P.BlockBandedLayout(np.full(10 ** 4, 32), 1, 1)
Jb = fd.BlockBandedMatrix(None, lay)
timed("BlockBanded 1e4 x 32^2", lambda: fd.make_plan(Jb, Jb, lay.colors(), "complex"))
