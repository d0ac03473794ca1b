"""numpy restatement of the DEFINED summation order of the masked sums of squares behind the step sizes (csrc/fdjac_kernels.hip,
k_eps_partial_reg: the rank-aligned two-level sum), element for element: thread-private accumulation over a block's tiles, the
64-lane shuffle tree, the block's four waves in order, a group's blocks in order, the 64 groups in order -- then the step rule of
src/epsilons.jl:26-29 / 50-53 with the sqrt of the 2-norm of src/jacobians.jl:561.  Every operation is an IEEE add / multiply /
sqrt in a fixed order, so the device's step sizes must equal these bit for bit, on any number of shards.  Test infrastructure."""
import numpy as np

GROUPS, BPG_MAX, TILE, BLOCK, U = 64, 16, 2048, 256, 4


def grid(n):
    tiles = (n + TILE - 1) // TILE
    tpg = (tiles + GROUPS - 1) // GROUPS
    tpb = (tpg + BPG_MAX - 1) // BPG_MAX
    bpg = (tpg + tpb - 1) // tpb
    return tpg, bpg, tpb


def masked_sumsq(x, colors0, ncolors, dtype=np.float64):
    """S[c] = sum over color0[j] == c of x[j]^2 in the device's order.  colors0: 0-based, < 0 = no colour."""
    n = x.size
    tpg, bpg, tpb = grid(n)
    nblocks = GROUPS * bpg
    xs = np.asarray(x, dtype=dtype).astype(np.float64)
    sq = xs * xs
    acc = np.zeros((ncolors, nblocks, BLOCK))
    g = np.arange(nblocks) // bpg
    k = np.arange(nblocks) % bpg
    t0 = g * tpg + k * tpb
    t1 = np.minimum(t0 + tpb, (g + 1) * tpg)
    thr = np.arange(BLOCK)
    for r in range(tpb):
        live = (t0 + r) < t1                                            # blocks that still have a tile in this round
        for u in range(U):
            for e in range(2):
                idx = (t0 + r)[:, None] * TILE + u * 2 * BLOCK + 2 * thr[None, :] + e
                ok = live[:, None] & (idx < n)
                idc = np.where(ok, idx, 0)
                v = np.where(ok, sq[idc], 0.0)
                col = np.where(ok, colors0[idc], -2)
                for c in range(ncolors):
                    acc[c] += np.where(col == c, v, 0.0)
    # wave tree: v += shfl_down(v, off), off = 32 .. 1 (lane 0's value), then the four waves in order
    w = acc.reshape(ncolors, nblocks, BLOCK // 64, 64)
    for off in (32, 16, 8, 4, 2, 1):
        w = w[..., :off] + w[..., off:2 * off]
    w = w[..., 0]
    blk = np.zeros((ncolors, nblocks))
    for i in range(BLOCK // 64):
        blk = blk + w[:, :, i]
    grp = np.zeros((ncolors, GROUPS))
    b = blk.reshape(ncolors, GROUPS, bpg)
    for i in range(bpg):
        grp = grp + b[:, :, i]
    tot = np.zeros(ncolors)
    for i in range(GROUPS):
        tot = tot + grp[:, i]
    return tot


def epsilons(x, colors0, ncolors, fdtype, relstep=None, absstep=None, dir=1.0, dtype=np.float64):
    T = np.dtype(dtype).type
    if relstep is None:
        e = np.finfo(dtype).eps
        relstep = float(np.sqrt(T(e))) if fdtype == "forward" else float(np.cbrt(T(e)))
    if absstep is None:
        absstep = relstep
    tot = masked_sumsq(x, colors0, ncolors, dtype)
    out = np.empty(ncolors, dtype=dtype)
    for c in range(ncolors):
        nrm = T(np.sqrt(tot[c]))
        xs = np.abs(np.sqrt(nrm))
        a = T(relstep) * xs
        e = a if a > T(absstep) else T(absstep)
        if fdtype == "forward":
            e = e * T(dir)
        out[c] = e
    return out
