import os, sys
os.environ.setdefault("FDJAC_TEST_SWITCHES", "1")   # (the library honours its variant switches only on request)
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import finitediff_jl_amd as fd
from finitediff_jl_amd import patterns as P
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    N = int(rng.integers(20_000, 200_000))
    l, u = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    M = N + int(rng.integers(-20, 21)) if rng.random() < 0.4 else N
    w = l + u + 1
    C = w if rng.random() < 0.6 else int(rng.integers(1, 9))
    colors = ((np.arange(N) + int(rng.integers(0, C))) % C + 1).astype(np.int64)
    fdtype = ["forward", "central", "complex"][int(rng.integers(0, 3))]
    banded = rng.random() < 0.5
    win = None
    if rng.random() < 0.6:
        a = int(rng.integers(0, N // 3)); win = (a + 1, int(rng.integers(a + N // 3, N)))
    T = [None, "512", "1024", "2048"][int(rng.integers(0, 4))]
    dev = str(int(rng.integers(0, 2)))
    x = torch.as_tensor(rng.random(N), device="cuda")
    A = torch.as_tensor(rng.random((M, w)), device="cuda")
    def fn(fx, xx):
        idx = torch.arange(M, device="cuda"); acc = torch.zeros(M, dtype=xx.dtype, device="cuda")
        for k in range(w):
            acc = acc + A[:, k].to(xx.dtype) * xx[torch.clamp(idx - l + k, 0, N - 1)] ** 2
        fx.copy_(acc)
    outs = []; infos = []
    for comp in ("1", "0"):
        os.environ["FDJAC_BAND_DESC"] = comp; os.environ["FDJAC_PLAN_DEVICE"] = dev
        if T: os.environ["FDJAC_WIN_TILE"] = T
        else: os.environ.pop("FDJAC_WIN_TILE", None)
        if banded:
            plan = fd.make_plan(fd.BandedMatrix(None, M, l, u), None, colors, fdtype, col_window=win)
        else:
            cp, rv = P.banded_csc(M, N, l, u)
            J = fd.SparseMatrixCSC(M, N, cp, rv)
            plan = fd.make_plan(J, J, colors, fdtype, col_window=win)
        infos.append(plan.info(fd.lib.INFO_BAND_DESC))
        out = torch.full((plan.out_len(0),), float("nan"), dtype=torch.float64, device="cuda")
        plan.jacobian(fd.TorchF(fn, M, N), x, [out]); outs.append(out)
    nb = int((outs[0] != outs[1]).sum())
    if nb or torch.isnan(outs[1]).any():
        bad += 1
        print("MISMATCH", seed, N, M, l, u, C, fdtype, banded, win, T, dev, infos, nb)
print("done", sys.argv[1], sys.argv[2], "bad", bad)
