#!/usr/bin/env python3
"""Generates the committed fixtures of tests/golden/.

    python tests/golden/make_golden.py          # rewrites reference_known_answers.json + oracle_vectors.npz

The reference (FiniteDiff.jl) is Julia and cannot be executed in this image, so there are two
kinds of fixture:

1. ``reference_known_answers.json`` -- every closed-form answer the reference's OWN tests assert for
   the coloured-Jacobian path, transcribed with the file:line they come from (/root/reference is
   read only by the person transcribing; nothing is read from it at run time).  Both the CPU
   oracle (tests/test_golden_fixtures.py, CPU) and the HIP path (same file, -m gpu) must
   reproduce them to the tolerance the reference test states.

2. ``oracle_vectors.npz`` -- seeded inputs and the oracle's outputs for the storage types / fdtypes
   of the path at sizes the oracle finishes instantly.  They freeze the oracle: a later edit of
   oracle/fd_oracle.c that changes any result fails the CPU suite, and the GPU suite compares the
   device results with these committed numbers rather than with whatever the oracle computes today.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def known_answers():
    """Closed-form answers asserted by the reference's tests (matrices are row-major lists)."""
    n = 4
    x0 = [1.5, 2.5, 3.5, 4.5, 2.5, 3.5, 4.5, 5.5]
    x1, x2 = np.array(x0[:n]), np.array(x0[n:])
    nonsq = np.zeros((n, 2 * n))
    nonsq[np.arange(n), np.arange(n)] = 2 * (x1 - 3) + x2          # d/dx1 of (x1-3)^2 + x1*x2 + (x2+4)^2 - 3
    nonsq[np.arange(n), np.arange(n) + n] = x1 + 2 * (x2 + 4)      # d/dx2
    return {
        "_about": "closed-form answers asserted by FiniteDiff.jl's own tests for the coloured-Jacobian path",
        "tridiagonal_second_difference": {
            "cite": "test/coloring_tests.jl:5-13 (f!), :19-26 (second_derivative_stencil), :33-49, :51-70, :72-96",
            "N": 30, "colorvec": "repeat(1:3, 10)", "diag": -2.0, "offdiag": 1.0,
            "fcalls": {"forward": 4, "central": 6, "complex": 3},
            "tolerance": "isapprox default: rtol = sqrt(eps) on the Frobenius norm",
        },
        "nonsquare_4x8": {
            "cite": "test/coloring_tests.jl:124-159",
            "x0": x0, "colorvec": [1, 1, 1, 1, 2, 2, 2, 2], "J": nonsq.tolist(),
            "fcalls": {"forward": 3, "central": 4, "complex": 2}, "rtol": 1e-6,
        },
        "dense_sparsity": {
            "cite": "test/coloring_tests.jl:171-219 (dense-matrix `sparsity`, forward, default colorvec = 1:length(x))",
            "tolerance": "isapprox default: rtol = sqrt(eps) on the Frobenius norm",
            "cases": [
                {"cite": ":171-179", "f": "_f", "x": [5.0, 3.0], "sparsity": [[1, 1], [1, 1]], "J": [[10.0, 6.0], [1.0, 1.0]]},
                {"cite": ":181-189", "f": "_f2", "x": [-3.0, 2.0], "sparsity": [[1, 1], [1, 0]], "J": [[-6.0, 4.0], [1.0, 0.0]]},
                {"cite": ":192-200", "f": "_f3", "x": [-3.0, 2.0], "sparsity": [[1, 1]], "J": [[-7.0, 4.0]]},
                {"cite": ":202-210", "f": "_f4", "x": [-3.0, 2.0, 13.3],
                 "sparsity": [[1, 1, 0], [1, 1, 0], [1, 0, 1], [1, 0, 0]],
                 "J": [[-7.0, 4.0, 0.0], [2.0, -3.0, 0.0], [13.3, 0.0, -3.0], [1.0, 0.0, 0.0]]},
                {"cite": ":212-219", "f": "_f5", "x": [5.0, 3.0], "sparsity": [[1, 1]], "J": [[10.0, 6.0]]},
            ],
        },
        "cache_reuse": {
            "cite": "test/cache_reuse_tests.jl:8-14,64-83", "f": "foo_iip!: y = [2 x1, 3 x2, 4 x1]", "x": [1.0, 2.0],
            "poison": 1.0e10, "J": [[2.0, 0.0], [0.0, 3.0], [4.0, 0.0]], "atol": 1e-6, "x_restored_bitwise": True,
        },
        "dense_arm_tolerances": {
            "cite": "test/finitedifftests.jl:455-462", "forward": 1e-6, "central": 1e-8, "complex": 1e-14,
        },
        "config1_dense_sin": {
            "cite": "BASELINE.json configs[0]; src/jacobians.jl:240-259,277-331", "N": 1000, "seed": 1,
            "J": "Diagonal(cos.(x))", "max_abs_err": 1e-6,
        },
    }


def oracle_cases():
    """(name, kwargs for oracle.jacobian, fixture spec) -- small, seeded."""
    from finitediff_jl_amd import patterns as P
    from oracle import oracle as O
    cases = []
    rng = np.random.default_rng(20260926)

    def add(name, fdtype, fixture, x, colors, **kw):
        cases.append({"name": name, "fdtype": fdtype, "fixture": fixture, "x": x, "colors": colors, "kw": kw})

    N = 301
    cp, rv = P.tridiag_csc(N)
    x = rng.random(N)
    for fdt in ("forward", "central", "complex"):
        add("tridiag_nl_csc_" + fdt, fdt, ("tridiag_nl", N), x, P.cyclic_colors(N, 3), kind=O.PAT_CSC_COMMON,
            colptr=cp, rowval=rv)
    add("tridiag_nl_csc_dirneg", "forward", ("tridiag_nl", N), x, P.cyclic_colors(N, 3), kind=O.PAT_CSC_COMMON,
        colptr=cp, rowval=rv, dir=-1.0)
    cols = P.cyclic_colors(N, 3).copy()
    cols[[0, 100, 300]] = 0
    add("tridiag_nl_csc_uncoloured", "forward", ("tridiag_nl", N), x, cols, kind=O.PAT_CSC_COMMON, colptr=cp, rowval=rv)
    add("tridiag_nl_tridiagonal", "central", ("tridiag_nl", N), x, P.cyclic_colors(N, 3), kind=O.PAT_COO_TRIDIAG,
        rows_index=rv, cols_index=P.csc_cols(cp))
    add("tridiag_nl_banded", "forward", ("tridiag_nl", N), x, P.cyclic_colors(N, 3), kind=O.PAT_BANDED, l=1, u=1)
    nx, ny = 24, 17
    cp5, rv5 = P.lap5_csc(nx, ny)
    x5 = rng.random(nx * ny)
    for fdt in ("forward", "central", "complex"):
        add("lap5_csc_" + fdt, fdt, ("lap5", nx, ny), x5, P.lap5_colors(nx, ny), kind=O.PAT_CSC_COMMON, colptr=cp5, rowval=rv5)
    add("clamp5_csc_forward", "forward", ("clamp5", nx, ny), x5, P.lap5_colors(nx, ny), kind=O.PAT_CSC_COMMON,
        colptr=cp5, rowval=rv5)
    nb, bs = 7, 5
    lay = P.BlockBandedLayout(np.full(nb, bs), 1, 1)
    xb = rng.random(nb * bs)
    for fdt in ("forward", "complex"):
        add("blockcoupled_blockbanded_" + fdt, fdt, ("blockcoupled", nb, bs), xb, lay.colors(), kind=O.PAT_BLOCKBANDED,
            blk_sizes=lay.blk_sizes, bl=1, bu=1, block_starts=lay.block_starts, block_strides=lay.block_strides,
            out_len=lay.data_len)
    n = 6
    A = np.zeros((n, 2 * n))
    A[np.arange(n), np.arange(n)] = 1
    A[np.arange(n), np.arange(n) + n] = 1
    cpn, rvn = P.csc_from_dense(A)
    xn = rng.random(2 * n) * 4
    for fdt in ("forward", "central", "complex"):
        add("nonsquare_csc_" + fdt, fdt, ("nonsquare", n), xn, np.repeat([1, 2], n), M=n, kind=O.PAT_CSC_COMMON,
            colptr=cpn, rowval=rvn)
    Nd = 40
    xd = rng.random(Nd)
    for fdt in ("forward", "central", "complex"):
        add("tridiag_nl_dense_" + fdt, fdt, ("tridiag_nl", Nd), xd, np.arange(1, Nd + 1), kind=O.PAT_NONE)
    return cases


def run_oracle(case):
    from oracle import oracle as O
    kw = dict(case["kw"])
    M = kw.pop("M", None)
    r = O.jacobian(case["fdtype"], O.Fixture(*case["fixture"]), case["x"], case["colors"], M, **kw)
    out = r["out"]
    if isinstance(out, tuple):
        out = np.concatenate([np.asarray(o).ravel() for o in out])   # (dl, d, du)
    return np.asarray(out).ravel(order="F"), r["fcalls"]


def main():
    with open(os.path.join(HERE, "reference_known_answers.json"), "w") as fh:
        json.dump(known_answers(), fh, indent=1)
    arrays = {}
    names = []
    for case in oracle_cases():
        out, fcalls = run_oracle(case)
        nm = case["name"]
        names.append(nm)
        arrays[nm + "/x"] = case["x"]
        arrays[nm + "/colors"] = np.asarray(case["colors"], np.int64)
        arrays[nm + "/out"] = out
        arrays[nm + "/fcalls"] = np.array([fcalls], np.int64)
    arrays["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **arrays)
    print("wrote %d oracle cases, %.1f KB" % (len(names), os.path.getsize(os.path.join(HERE, "oracle_vectors.npz")) / 1e3))


if __name__ == "__main__":
    main()
