#!/usr/bin/env python3
"""Which part of a plan differs between the host and the device builder?  (FDJAC_CHECKSUM_TRACE=1: running hash per array on stderr)
    python scripts/plan_checksum_trace.py band13 forward"""
import os
os.environ.setdefault("FDJAC_TEST_SWITCHES", "1")   # (the library honours its variant switches only on request)
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import finitediff_jl_amd as fd  # noqa: E402
from finitediff_jl_amd import patterns as P  # noqa: E402

case, fdtype = sys.argv[1], sys.argv[2]
N = 200_000
colptr, rowval = P.banded_csc(N, N, 6, 6)
colors = P.cyclic_colors(N, 13)
win = (N // 5 + 1, 4 * N // 5) if case.endswith("window") else None
J = fd.SparseMatrixCSC(N, N, colptr, rowval)
os.environ["FDJAC_CHECKSUM_TRACE"] = "1"
for dev in ("0", "1"):
    os.environ["FDJAC_PLAN_DEVICE"] = dev
    plan = fd.make_plan(J, J, colors, fdtype, col_window=win)
    print("builder", dev, file=sys.stderr, flush=True)
    plan.checksum()
