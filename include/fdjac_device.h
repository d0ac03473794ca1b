/*
 * fdjac_device.h -- the device-side piece of libfdjac's boundary (EXPERIMENTAL, opt-in: FDJAC_LAZY_STORE=1).
 *
 * With the differences handed over (FD_LAZY_CAP_DIFF) more than half of a Jacobian's HBM traffic is the hand-off between
 * f!'s launch and the decompression.  For a banded CSC Jacobian whose colours are cyclic the
 * storage position of the entry (row r, colour c) is arithmetic, so an f! kernel can store the finished difference quotient
 * itself:  nzval[fd_band_dest(&desc, r, c)] = (f(x + eps_c m_c)[r] - f(x)[r]) / eps_c  -- the operations of
 * src/jacobians.jl:565 / 607 and ext/FiniteDiffSparseArraysExt.jl:38-47 for that entry, on the values the plain path would
 * have stored (same bits); the library then launches nothing after f!.
 *
 * The library hands the descriptor to lazy launchers registered with FD_LAZY_CAP_STORE (fd_lazy_points.store) when -- and
 * only when -- the plan has verified that the pattern IS the band this arithmetic describes: every column j holds exactly
 * the rows max(0, j-u) .. min(M-1, j+l), colorvec[j] = (j + shift) mod C + 1 with C >= l + u + 1 (then at most one column of
 * a colour touches a row).  Plain C: usable from HIP kernels and from the host (the plan's verification runs it there).
 */
#ifndef FDJAC_DEVICE_H
#define FDJAC_DEVICE_H

#if defined(__HIPCC__) || defined(__CUDACC__)
#define FD_DEVICE_FN __host__ __device__ static inline
#else
#define FD_DEVICE_FN static inline
#endif

typedef struct fd_band_store {
    void *out;                     /* the stored values of the local column range (nzval + entry_begin), device memory */
    long long M, N;                /* matrix shape */
    long long entry_begin;         /* global 0-based index of the first stored entry of the local column range */
    long long col_begin, col_end;  /* local column range [col_begin, col_end), 0-based */
    int l, u;                      /* lower / upper bandwidth */
    int C, shift;                  /* 0-based colour of column j: (j + shift) mod C */
} fd_band_store;

/* 0-based global index of the first stored entry of column j (closed form of colptr for the exact band) */
FD_DEVICE_FN long long fd_band_colptr(const fd_band_store *d, long long j)
{
    const long long w = (long long)d->l + d->u + 1;
    /* rows cut off above the matrix: columns i < u lose u - i rows */
    const long long nt = j < d->u ? j : d->u;
    const long long top = nt * d->u - nt * (nt - 1) / 2;
    /* rows cut off below: columns i >= M - l lose i + l - (M - 1) rows */
    const long long b0 = d->M - d->l, f0 = b0 > 0 ? b0 : 0;
    long long bot = 0;
    if (j > f0) {
        const long long n = j - f0, a = f0 - b0 + 1;
        bot = n * a + n * (n - 1) / 2;
    }
    return w * j - top - bot;
}

/* index into fd_band_store.out of the stored entry in row r whose column has colour c (0-based), or -1 if there is none
   in the local column range */
FD_DEVICE_FN long long fd_band_dest(const fd_band_store *d, long long r, int c)
{
    const long long j0 = r - d->l;                                   /* first column that can touch row r */
    long long m = (j0 + d->shift) % d->C;
    if (m < 0) m += d->C;
    long long t = c - m;
    if (t < 0) t += d->C;
    const long long j = j0 + t;
    if (t > (long long)d->l + d->u || j < d->col_begin || j >= d->col_end || j < 0 || j >= d->N) return -1;
    const long long first = j - d->u > 0 ? j - d->u : 0;
    return fd_band_colptr(d, j) - d->entry_begin + (r - first);
}

#endif /* FDJAC_DEVICE_H */
