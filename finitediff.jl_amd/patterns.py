"""Sparsity-pattern and colouring builders for the benchmark / test families (numpy only).

All index arrays follow the reference's (Julia) conventions: 1-based Int64, CSC with sorted
row indices per column, colours 1..C.  These are the *inputs* a FiniteDiff.jl user hands to
``finite_difference_jacobian!`` (``sparsity``, ``colorvec``); nothing here runs on the GPU.
"""
import numpy as np


def tridiag_csc(n):
    """SparseMatrixCSC pattern of an n x n tridiagonal matrix -> (colptr, rowval), 1-based Int64."""
    n = int(n)
    if n == 1:
        return np.array([1, 2], np.int64), np.array([1], np.int64)
    cnt = np.full(n, 3, np.int64)
    cnt[0] = cnt[-1] = 2
    colptr = np.empty(n + 1, np.int64)
    colptr[0] = 1
    np.cumsum(cnt, out=colptr[1:])
    colptr[1:] += 1
    nnz = 3 * n - 2
    # entry k of column j (0-based j): row = j-1+k for interior; first column starts at row j
    p = np.arange(nnz, dtype=np.int64)
    # position p belongs to column j = (p+1)//3 (valid for the tridiagonal layout above)
    j = (p + 1) // 3
    rowval = p - (colptr[j] - 1) + np.maximum(j - 1, 0) + 1
    return colptr, rowval


def banded_csc(m, n, l, u):
    """SparseMatrixCSC pattern of an m x n band matrix with bandwidths (l, u) -> (colptr, rowval), 1-based."""
    m, n, l, u = int(m), int(n), int(l), int(u)
    j = np.arange(n, dtype=np.int64)
    lo = np.maximum(j - u, 0)
    hi = np.minimum(j + l, m - 1)
    cnt = np.maximum(hi - lo + 1, 0)
    colptr = np.empty(n + 1, np.int64)
    colptr[0] = 1
    np.cumsum(cnt, out=colptr[1:])
    colptr[1:] += 1
    col = np.repeat(j, cnt)
    rowval = np.arange(col.size, dtype=np.int64) - (colptr[col] - 1) + lo[col] + 1
    return colptr, rowval


def cyclic_colors(n, c):
    """colorvec[i] = mod1(i, c) (1-based) -- the valid colouring of a band of width c."""
    return (np.arange(n, dtype=np.int64) % c) + 1


def lap5_csc(nx, ny):
    """5-point stencil pattern on an nx (fast index) x ny grid, N = nx*ny, sorted CSC, 1-based."""
    nx, ny = int(nx), int(ny)
    N = nx * ny
    k = np.arange(N, dtype=np.int64)
    i, j = k % nx, k // nx
    # neighbours in increasing row order: south (k-nx), west (k-1), self, east (k+1), north (k+nx)
    has = np.stack([j > 0, i > 0, np.ones(N, bool), i < nx - 1, j < ny - 1], axis=1)
    rows = np.stack([k - nx, k - 1, k, k + 1, k + nx], axis=1)
    cnt = has.sum(axis=1)
    colptr = np.empty(N + 1, np.int64)
    colptr[0] = 1
    np.cumsum(cnt, out=colptr[1:])
    colptr[1:] += 1
    rowval = rows[has] + 1
    return colptr, rowval.astype(np.int64)


def lap5_colors(nx, ny):
    """((i + 2j) mod 5) + 1 : a distance-2 colouring of the 5-point stencil with 5 colours."""
    k = np.arange(int(nx) * int(ny), dtype=np.int64)
    return ((k % nx) + 2 * (k // nx)) % 5 + 1


def csc_from_dense(A):
    """sparse(A) structure of a dense matrix -> (colptr, rowval) 1-based."""
    A = np.asarray(A)
    m, n = A.shape
    colptr = np.empty(n + 1, np.int64)
    colptr[0] = 1
    rv = []
    for j in range(n):
        r = np.nonzero(A[:, j])[0]
        rv.append(r + 1)
        colptr[j + 1] = colptr[j] + r.size
    return colptr, (np.concatenate(rv).astype(np.int64) if rv else np.zeros(0, np.int64))


def csc_to_dense(M, N, colptr, rowval, nzval):
    J = np.zeros((M, N), dtype=np.asarray(nzval).dtype)
    for j in range(N):
        sl = slice(colptr[j] - 1, colptr[j + 1] - 1)
        J[rowval[sl] - 1, j] = nzval[sl]
    return J


def csc_cols(colptr):
    """Column index (1-based) of every stored entry."""
    n = colptr.size - 1
    return np.repeat(np.arange(1, n + 1, dtype=np.int64), np.diff(colptr))


class BlockBandedLayout:
    """The storage layout a BlockBandedMatrix hands over (third-party BlockBandedMatrices.jl
    ``block_starts`` / ``block_strides``): block-column J's in-band blocks are stacked into one
    dense column-major panel of ``stride[J]`` rows; panels are laid out one after the other.

    blk_sizes: sizes of the (square) block structure; (bl, bu) block bandwidths.
    block_starts is the (bl+bu+1) x nblk band storage (column-major, like BandedMatrix data) of
    the 1-based start offset of block (K,J): block_starts[(bu + K - J) + (bl+bu+1)*J], 0-based K,J.
    """

    def __init__(self, blk_sizes, bl, bu):
        bs = np.asarray(blk_sizes, np.int64)
        nb = bs.size
        self.blk_sizes, self.bl, self.bu, self.nblk = bs, int(bl), int(bu), nb
        w = self.bl + self.bu + 1
        starts = np.zeros((w, nb), np.int64, order="F")
        strides = np.zeros(nb, np.int64)
        off = 1
        for J in range(nb):
            K0, K1 = max(0, J - bu), min(nb - 1, J + bl)
            strides[J] = bs[K0:K1 + 1].sum()
            o = off
            for K in range(K0, K1 + 1):
                starts[bu + K - J, J] = o
                o += bs[K]
            off += strides[J] * bs[J]
        self.block_starts = starts.reshape(-1, order="F").copy()
        self.block_strides = strides
        self.data_len = int(off - 1)
        self.N = int(bs.sum())

    def to_dense(self, data):
        bs, nb = self.blk_sizes, self.nblk
        rowoff = np.concatenate([[0], np.cumsum(bs)])
        A = np.zeros((self.N, self.N), dtype=np.asarray(data).dtype)
        for J in range(nb):
            K0, K1 = max(0, J - self.bu), min(nb - 1, J + self.bl)
            st = self.block_strides[J]
            s = self.block_starts[(self.bu + K0 - J) + (self.bl + self.bu + 1) * J] - 1
            panel = np.asarray(data[s:s + st * bs[J]]).reshape((st, bs[J]), order="F")
            A[rowoff[K0]:rowoff[K1 + 1], rowoff[J]:rowoff[J + 1]] = panel
        return A

    def index_of(self, rows0, cols0):
        """0-based offsets into data of entries (rows0, cols0) (0-based, must be in band)."""
        bs, nb = self.blk_sizes, self.nblk
        off = np.concatenate([[0], np.cumsum(bs)])
        K = np.searchsorted(off, rows0, side="right") - 1
        J = np.searchsorted(off, cols0, side="right") - 1
        assert np.all((K - J <= self.bl) & (J - K <= self.bu)), "entry outside the block band"
        st = self.block_starts[(self.bu + K - J) + (self.bl + self.bu + 1) * J] - 1
        assert np.all(st >= 0), "entry outside the block band"
        return st + (cols0 - off[J]) * self.block_strides[J] + (rows0 - off[K])

    def colors(self):
        """Valid colouring for dense in-band blocks: cols of block-columns J, J+bl+bu+1, ... share."""
        w = self.bl + self.bu + 1
        bsmax = int(self.blk_sizes.max())
        out = []
        for J, b in enumerate(self.blk_sizes):
            out.append(bsmax * (J % w) + np.arange(1, b + 1))
        return np.concatenate(out).astype(np.int64)


class BandedBlockBandedLayout:
    """Storage of a BandedBlockBandedMatrix as the reference's extension addresses it
    (ext/FiniteDiffBlockBandedMatricesExt.jl:29-36): every in-band block (K,J) owns a banded-data slab
    of (lam+mu+1) x n_J values with column stride ``st``; entry (k,j) of the block (1-based) lives at
    ``start(K,J) + (j-1)*st + mu + k - j`` (0-based offset from a 1-based start).  The slabs are laid out as
    BlockBandedMatrices.jl does [third-party layout, taken from its documentation, not verifiable here -- the C ABI
    receives the starts / strides explicitly, so a different layout only changes these arrays]: one
    ((bl+bu+1)*(lam+mu+1)) x N column-major matrix, block-band d = K-J+bu occupying rows d*(lam+mu+1) ...
    """

    def __init__(self, blk_sizes, bl, bu, lam, mu):
        bs = np.asarray(blk_sizes, np.int64)
        nb = bs.size
        self.blk_sizes, self.bl, self.bu, self.lam, self.mu, self.nblk = bs, int(bl), int(bu), int(lam), int(mu), nb
        w, sw = self.bl + self.bu + 1, self.lam + self.mu + 1
        self.N = int(bs.sum())
        R = w * sw
        off = np.concatenate([[0], np.cumsum(bs)])
        starts = np.zeros((w, nb), np.int64, order="F")
        for J in range(nb):
            for K in range(max(0, J - bu), min(nb - 1, J + bl) + 1):
                starts[bu + K - J, J] = 1 + off[J] * R + (bu + K - J) * sw
        self.block_starts = starts.reshape(-1, order="F").copy()
        self.block_strides = np.full(nb, R, np.int64)
        self.data_len = int(R * self.N)
        self._off = off

    def entries(self):
        """(rows, cols, dest): 1-based row / column and 0-based offset into data of every in-band entry,
        in column order -- what the shim enumerates once for fd_plan_create_entries."""
        bs, nb, off = self.blk_sizes, self.nblk, self._off
        w = self.bl + self.bu + 1
        rows, cols, dest = [], [], []
        for J in range(nb):
            n = int(bs[J])
            j = np.arange(n, dtype=np.int64)
            for K in range(max(0, J - self.bu), min(nb - 1, J + self.bl) + 1):
                m = int(bs[K])
                start = self.block_starts[(self.bu + K - J) + w * J] - 1
                st = self.block_strides[J]
                for t in range(-self.mu, self.lam + 1):        # k = j + t
                    k = j + t
                    ok = (k >= 0) & (k < m)
                    rows.append(off[K] + k[ok] + 1)
                    cols.append(off[J] + j[ok] + 1)
                    dest.append(start + j[ok] * st + self.mu + t)
        rows, cols, dest = np.concatenate(rows), np.concatenate(cols), np.concatenate(dest)
        order = np.lexsort((rows, cols))
        return rows[order], cols[order], dest[order]

    def colors(self):
        """(J mod (bl+bu+1))*(lam+mu+1) + (j mod (lam+mu+1)) + 1: a valid colouring of the pattern."""
        w, sw = self.bl + self.bu + 1, self.lam + self.mu + 1
        out = []
        for J, b in enumerate(self.blk_sizes):
            out.append(sw * (J % w) + (np.arange(b) % sw) + 1)
        return np.concatenate(out).astype(np.int64)


def banded_to_dense(data, M, N, l, u):
    """BandedMatrix data[(u + i - j), j] (0-based) -> dense."""
    A = np.zeros((M, N), dtype=np.asarray(data).dtype)
    for j in range(N):
        for i in range(max(0, j - u), min(M, j + l + 1)):
            A[i, j] = data[u + i - j, j]
    return A
