/*
 * fdjac_device.h -- the device-side piece of libfdjac's boundary: an f! kernel that stores the Jacobian itself.
 *
 * With the differences handed over (FD_LAZY_CAP_DIFF) more than half of a Jacobian's HBM traffic is the hand-off between
 * f!'s launch and the decompression.  For a banded Jacobian whose colours are cyclic the storage
 * position of the entry (row r, colour c) is arithmetic, so an f! kernel can store the finished difference quotient itself:
 *     fd_band_emit(&desc, r, c, (f(x + eps_c m_c)[r] - f(x)[r]) / eps_c)
 * -- the operations of src/jacobians.jl:565 / 607 and the assignment of ext/FiniteDiffSparseArraysExt.jl:38-47 (CSC nzval),
 * ext/FiniteDiffBandedMatricesExt.jl:13-27 (BandedMatrix data) or src/iteration_utils.jl:25-32 on a Tridiagonal (dl, d, du)
 * for that entry, on the values the plain path would have stored (same bits); the library then launches nothing after f!.
 *
 * The library hands the descriptor to lazy launchers registered with FD_LAZY_CAP_STORE (fd_lazy_points.store) when -- and
 * only when -- the plan has verified that the pattern IS the band this arithmetic describes: every column j holds exactly
 * the rows max(0, j-u) .. min(M-1, j+l), colorvec[j] = (j + shift) mod C + 1 with C >= l + u + 1 (then at most one column of
 * a colour touches a row).  The first part of this header is plain C, usable from HIP kernels and from the host (the plan's
 * verification runs it there); the second part (HIP C++) holds the store helpers:
 *     fd_band_emit<T>            one entry, any layout (an 8-byte store per entry: simple, not bandwidth-optimal)
 *     fd_colrange_emit<T>        one entry of a column-range storage (BlockBandedMatrix data), given (row, column)
 *     fd_band_emit_wave<T, W>    a wavefront's 128 columns at once, staged through a wave-private LDS window and written
 *                                back as dense, aligned, non-temporal 16-byte stores -- what the built-in tridiagonal
 *                                launcher uses (N = 10^7, Float64: 320 MB in 53-60 us, 0.67-0.76 of the 8 TB/s peak)
 * examples/user_f_store.hip is a user kernel built on them, compiled apart from the library.
 */
#ifndef FDJAC_DEVICE_H
#define FDJAC_DEVICE_H

#if defined(__HIPCC__)
#define FD_DEVICE_FN __host__ __device__ static inline
#else
#define FD_DEVICE_FN static inline
#endif

enum fd_band_layout {
    FD_BAND_CSC = 0,         /* SparseMatrixCSC nzval: the stored entries of column j are contiguous from fd_band_colptr(j)       */
    FD_BAND_BANDED = 1,      /* BandedMatrix data, (l+u+1) x N column-major: J[r,j] at (u + r - j) + (l+u+1) j; the slots of rows   */
                             /* outside the matrix exist and hold 0                                                                */
    FD_BAND_TRIDIAGONAL = 2  /* LinearAlgebra.Tridiagonal (l = u = 1, M = N): J[j+1,j] -> dl[j], J[j,j] -> d[j], J[j-1,j] -> du[j-1] */
};

typedef struct fd_band_store {
    void *out;                     /* CSC: nzval of the local column range; BANDED: its slice of data; TRIDIAGONAL: d -- device memory */
    long long M, N;                /* matrix shape */
    long long entry_begin;         /* CSC: global 0-based index of the first stored entry of the local column range */
    long long col_begin, col_end;  /* local column range [col_begin, col_end), 0-based: out holds only its entries */
    int l, u;                      /* lower / upper bandwidth */
    int C, shift;                  /* 0-based colour of column j: (j + shift) mod C */
    int layout;                    /* enum fd_band_layout */
    int elem_bytes;                /* 8 (Float64) or 4 (Float32) */
    void *out_dl, *out_du;         /* TRIDIAGONAL: dl and du of the local column range (dl[j - col_begin], du[j - 1 - max(col_begin - 1, 0)]) */
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

/* 0-based colour of column j */
FD_DEVICE_FN int fd_band_color(const fd_band_store *d, long long j)
{
    long long m = (j + d->shift) % d->C;
    if (m < 0) m += d->C;
    return (int)m;
}

/* the column of colour c (0-based) that touches row r, or -1 if there is none in the local column range */
FD_DEVICE_FN long long fd_band_column(const fd_band_store *d, long long r, int c)
{
    const long long j0 = r - d->l;                                   /* first column that can touch row r */
    long long m = (j0 + d->shift) % d->C;
    if (m < 0) m += d->C;
    long long t = c - m;
    if (t < 0) t += d->C;
    const long long j = j0 + t;
    if (t > (long long)d->l + d->u || j < d->col_begin || j >= d->col_end || j < 0 || j >= d->N) return -1;
    return j;
}

/* CSC / BANDED layouts: index into fd_band_store.out of the stored entry in row r whose column has colour c (0-based), or -1
   if there is none in the local column range */
FD_DEVICE_FN long long fd_band_dest(const fd_band_store *d, long long r, int c)
{
    const long long j = fd_band_column(d, r, c);
    if (j < 0 || r < 0 || r >= d->M) return -1;
    if (d->layout == FD_BAND_BANDED) return (d->u + r - j) + ((long long)d->l + d->u + 1) * (j - d->col_begin);
    const long long first = j - d->u > 0 ? j - d->u : 0;
    return fd_band_colptr(d, j) - d->entry_begin + (r - first);
}

/* ---- 5-point stencil on an nx x ny grid (natural ordering k = i + nx j, i fastest), SparseMatrixCSC nzval --------------------
 * Column k = (i, j) holds the rows k - nx (j > 0), k - 1 (i > 0), k, k + 1 (i < nx - 1), k + nx (j < ny - 1), in that order.
 * The plan hands this descriptor out (fd_lazy_points.store with store_kind = FD_STORE_STENCIL5) when it has verified that the
 * pattern is exactly that and that colorvec is a VALID colouring of it (the columns that share a row have different colours:
 * then the colour-c point seen from a row differs from x in at most one coordinate, and a column-centric evaluation -- x +- eps
 * e_k for the rows column k touches -- forms exactly the operands of the reference's colour-batched evaluation). */
typedef struct fd_stencil5_store {
    void *out;                     /* nzval of the local column range, device memory */
    long long nx, ny;              /* grid shape; M = N = nx * ny */
    long long entry_begin;         /* global 0-based index of the first stored entry of the local column range */
    long long col_begin, col_end;  /* local column range [col_begin, col_end) */
    const void *color;             /* device: 0-based colour of every column (all N), color_bytes each (1 or 4) */
    int color_bytes, C;
    int elem_bytes, reserved0;
} fd_stencil5_store;

/* 0-based global index of the first stored entry of column k = j * nx + i (grid point (i, j)); a kernel that knows (i, j) calls
 * this form -- no 64-bit division (about 150 instructions per call on the device) */
FD_DEVICE_FN long long fd_stencil5_colptr_ij(const fd_stencil5_store *d, long long i, long long j)
{
    const long long k = j * d->nx + i;
    const long long north = k - (d->ny - 1) * d->nx;           /* columns of the last grid row before k (no k + nx entry) */
    return 5 * k - (k < d->nx ? k : d->nx)                     /* ... of the first grid row (no k - nx entry) */
           - (j + (i > 0 ? 1 : 0)) - j                         /* first / last columns of the grid rows before k */
           - (north > 0 ? north : 0);
}
/* ... of column k */
FD_DEVICE_FN long long fd_stencil5_colptr(const fd_stencil5_store *d, long long k)
{
    const long long j = k / d->nx;
    return fd_stencil5_colptr_ij(d, k - j * d->nx, j);
}

/* ---- storage in which the stored rows of every column are CONSECUTIVE (BlockBandedMatrix data: the in-band blocks of a
 * block-column are stacked, ext/FiniteDiffBlockBandedMatricesExt.jl:44-68): row r of local column j lives at
 * out[dest[j - col_begin] + (r - row_first[j - col_begin])], row_first <= r < row_first + row_count.  Handed out
 * (store_kind = FD_STORE_COLRANGE) when the plan has verified that colorvec is a VALID colouring (columns that share a row
 * differ in colour), so that a column-centric evaluation forms the operands of the colour-batched one. */
typedef struct fd_colrange_store {
    void *out;                       /* the stored values of the local column range, device memory */
    long long M, N;
    long long col_begin, col_end;
    const int *row_first;            /* device, per local column */
    const int *row_count;            /* device */
    const long long *dest;           /* device */
    const void *color;               /* device: 0-based colour of every column (all N), color_bytes each */
    int color_bytes, C;
    int elem_bytes;
    int pairs;                       /* 1: every row_first, row_count and dest is even (16-byte stores of row pairs are aligned) */
    long long nblk, block_size;      /* block structure if the blocks are uniform (block_size = 0: ragged), block bandwidths */
    int bl, bu;
} fd_colrange_store;

/* ---- ANY SparseMatrixCSC pattern, ANY colouring: the column-centric store -------------------------------------------------------
 * The three descriptors above need a closed form for "where does the entry of row r and colour c live".  A general pattern has
 * none -- but the reference's decompression (ext/FiniteDiffSparseArraysExt.jl:38-47) never asks that question: it walks the
 * stored entries of every column j of the current colour and assigns  nzval[q] = (f(x + eps_c m_c)[r] - f(x)[r]) / eps_c  for
 * q = (r, j), c = colorvec[j].  A kernel that can evaluate ONE row of the residual at a colour's point does exactly that, column by
 * column: thread j walks column j's entries, evaluates each entry's row at the point of ITS colour and stores -- the values land in
 * storage order (a thread's entries are contiguous in nzval, neighbouring threads' runs follow each other: the L2 merges them into
 * whole lines), no hand-off arrays, and no requirement on the colouring at all (the point is formed for the whole colour, so an
 * invalid colouring gives what the reference gives).  The plan keeps a compact copy of the local pattern for it
 * (fd_plan_opts.flags & FD_PLAN_STORE_CSC: 4 bytes per stored entry + 4 per column) and hands this descriptor
 * (store_kind = FD_STORE_CSC) to launchers registered with FD_LAZY_CAP_STORE_CSC.  Forward differences: f(x) is evaluated ONCE by
 * the plain launcher (or is the caller's f_in) and arrives as `fx_base`; the launch evaluates one row per stored entry
 * (central: two) -- M + nnz row evaluations instead of the (1 + C) M of the colour-by-colour loop.  A launcher that also declares
 * FD_LAZY_CAP_STORE_CSC_BASE evaluates the unperturbed row itself when `fx_base` is NULL (no plain evaluation, no f(x) array: worth
 * it when a row is cheap next to a gather -- the 7-point family; fd_csc_store_cols does it for any functor).
 * (Measured alternative, round 4: a ROW-centric kernel with a per-(row, colour) destination table needs a valid colouring and
 * scatters 8-byte stores over nzval -- every one a 32-byte read-modify-write at the memory side: 1.8 GB written for 446 MB of
 * values on the 200^3 7-point pattern, profiles/r04_g_rowcentric_store_pmc_*.md.) */
typedef struct fd_csc_store {
    void *out;                     /* nzval of the local column range, device memory */
    long long M, N;
    long long col_begin, col_end;  /* local column range */
    const int *colptr;             /* device, col_end - col_begin + 1 offsets into out / rowval: column j holds [colptr[j - col_begin], colptr[j - col_begin + 1]) */
    const int *rowval;             /* device, per local entry: 0-based row */
    const void *color;             /* device: 0-based colour of every column (all N), color_bytes each; "none" = 0xFF / -1 */
    const void *fx_base;           /* device: f(x) (M elements) for forward differences; NULL for central differences */
    int color_bytes, C;
    int elem_bytes;
    int valid_coloring;            /* 1: the plan has verified that the columns sharing a row differ in colour -- then, seen from the row of a */
                                   /* stored entry (r, j), the colour's point differs from x in coordinate j ONLY, and a kernel may form it as */
                                   /* x[i] + eps * (i == j) without reading colours; 0: not verified (form the whole colour's point)          */
    unsigned long long *note;      /* device, 4 words owned by the PLAN and zero when it is created: a launcher's memory about THIS pattern   */
                                   /* (which never changes while the plan lives) -- e.g. "verified on an earlier call: it is exactly my stencil, */
                                   /* the row indices need not be read again" (the 7-point family: 290 -> 210 us).  May be NULL.              */
    long long reach;               /* max |row - column| over the local stored entries (<= 0: not computed / diagonal).  A HINT for staging: the rows a    */
                                   /* workgroup's 256 columns touch lie within `reach` of them, and -- for a residual whose rows read the     */
                                   /* columns of the Jacobian's own pattern -- the coordinates those rows read within 2 * reach               */
                                   /* (fd_csc_store_cols_win keeps that window of x in LDS; anything outside it is read from memory)         */
    unsigned long long plan_serial;/* unique per plan (never reused): what a launcher may key its own HOST-side memory about the plan on      */
    /* the same pattern BY ROWS (plans created with FD_PLAN_STORE_CSC_ROWS that hold every column; else NULL): what fd_csc_store_rows walks   */
    const int *row_ptr;            /* device, M + 1 offsets: row r's stored entries are [row_ptr[r], row_ptr[r + 1]) of the two lists below   */
    const int *row_col;            /* device, per entry in row-major order: its 0-based column (ascending within a row)                       */
    const int *row_slot;           /* device, per entry in row-major order: its index in out / rowval (the entry's slot in J's CSC storage)   */
    const int *row_pack;           /* device, 2 ints per row, tile of 256 rows by tile, the tile's rows by DESCENDING length (the order        */
                                   /* fd_csc_store_rows hands them to its threads in: a wavefront then holds rows of similar length):          */
                                   /* {row, (its first entry's place in the tile's run of the lists, <= 65535) | (its length, <= 32767) << 16} */
    const int *row_tile;           /* device, ceil(M / 256) + 1 offsets: tile t's run of the lists is [row_tile[t], row_tile[t + 1])           */
    /* the ENTRIES tile by tile in the order of row_pack (or NULL: a tile with more than 2048 entries, a row with more than 255): what     */
    /* fd_csc_store_ents walks -- a thread per entry; tile t's entries are [row_tile[t], row_tile[t + 1]) of these lists too                  */
    const int *ent_col;            /* device, per entry: its 0-based column                                                                    */
    const int *ent_slot;           /* device, per entry: its index in out / rowval                                                             */
    const int *ent_info;           /* device, per entry: (its row's place in the tile's order) | (its place in the row) << 8 | (the row's length) << 16 */
    int ent_tile_max;              /* the largest number of entries a tile holds (<= 2048): sizes fd_csc_store_ents' LDS                       */
} fd_csc_store;

/* ---- BandedBlockBandedMatrix storage with UNIFORM blocks (ext/FiniteDiffBlockBandedMatricesExt.jl:16-42, round 5) --------------------
 * Block (K, J) of the block band (-bu <= K - J <= bl) owns a slab of banded data: entry (k, j) of the block (0-based, -mu <= k - j <= lam)
 * lives at  start[(bu + K - J) + (bl + bu + 1) J] + j stride[J] + mu + k - j  (the reference's raw offsets, its lines 29-36).  A launcher
 * that knows its residual's structure writes every slot of every in-band slab of its columns itself -- the quotient where the row
 * depends on the column, 0 elsewhere (for a colouring that is valid for the BBB pattern, which the plan verified, the reference's
 * difference of an independent row is exactly 0) -- and nothing is launched after f!.  Whole column range, all colours in one batch. */
typedef struct fd_bbb_store {
    void *out;                     /* J.data, device */
    long long N, nblk, block_size; /* N = nblk * block_size */
    int bl, bu, lam, mu;
    const long long *start;        /* device, (bl + bu + 1) * nblk slab starts (0-based), -1: no slab */
    const long long *stride;       /* device, nblk column strides */
    const void *color;             /* device: 0-based colour of every column, color_bytes each; "none" = 0xFF / -1 */
    int color_bytes, C;
    int elem_bytes;
} fd_bbb_store;

enum fd_store_kind { FD_STORE_NONE = 0, FD_STORE_BAND = 1, FD_STORE_STENCIL5 = 2, FD_STORE_COLRANGE = 3, FD_STORE_CSC = 4, FD_STORE_BBB = 5 };   /* what fd_lazy_points.store points to */

#if defined(__HIPCC__) && defined(__cplusplus)
/* ---------------------------------------------------------------------------------------------------------------------------
 * Store helpers for HIP kernels.  T = double or float (= the plan's element type, fd_band_store.elem_bytes).
 * ------------------------------------------------------------------------------------------------------------------------- */

/* One entry: the quotient of (row r, colour c).  Entries outside the local column range are ignored. */
template <typename T> __device__ inline void fd_band_emit(const fd_band_store *d, long long r, int c, T value)
{
    if (d->layout == FD_BAND_TRIDIAGONAL) {
        const long long j = fd_band_column(d, r, c);
        if (j < 0 || r < 0 || r >= d->M) return;
        const long long du0 = d->col_begin > 0 ? d->col_begin - 1 : 0;
        if (r == j) ((T *)d->out)[j - d->col_begin] = value;
        else if (r == j + 1) ((T *)d->out_dl)[j - d->col_begin] = value;
        else ((T *)d->out_du)[j - 1 - du0] = value;
        return;
    }
    const long long at = fd_band_dest(d, r, c);
    if (at < 0) return;
    ((T *)d->out)[at] = value;
    if (d->layout == FD_BAND_BANDED) {
        /* the slots of rows outside the matrix hold 0 (fd_plan_create_banded's contract): written with the column's first /
           last row inside the matrix (the plan hands the descriptor out only if every column has one) */
        const long long j = fd_band_column(d, r, c);
        if (r == 0) for (long long k = 1; k <= d->u - j; ++k) ((T *)d->out)[at - k] = (T)0;
        if (r == d->M - 1) for (long long k = 1; k <= j + d->l - (d->M - 1); ++k) ((T *)d->out)[at + k] = (T)0;
    }
}

/* A whole column: q[k] = quotient of (row j - u + k, column j), k = 0 .. l+u.  BANDED: the slots of rows outside the matrix
   are written as 0 (fd_plan_create_banded's contract). */
template <typename T> __device__ inline void fd_band_emit_column(const fd_band_store *d, long long j, const T *q)
{
    if (j < d->col_begin || j >= d->col_end || j < 0 || j >= d->N) return;
    const int w = d->l + d->u + 1;
    if (d->layout == FD_BAND_TRIDIAGONAL) {
        const long long du0 = d->col_begin > 0 ? d->col_begin - 1 : 0;
        if (j > 0) ((T *)d->out_du)[j - 1 - du0] = q[0];
        ((T *)d->out)[j - d->col_begin] = q[1];
        if (j + 1 < d->M) ((T *)d->out_dl)[j - d->col_begin] = q[2];
        return;
    }
    if (d->layout == FD_BAND_BANDED) {
        T *o = (T *)d->out + (long long)w * (j - d->col_begin);
        for (int k = 0; k < w; ++k) { const long long r = j - d->u + k; o[k] = (r >= 0 && r < d->M) ? q[k] : (T)0; }
        return;
    }
    const long long first = j - d->u > 0 ? j - d->u : 0;
    T *o = (T *)d->out + (fd_band_colptr(d, j) - d->entry_begin) - (first - (j - d->u));
    for (int k = 0; k < w; ++k) { const long long r = j - d->u + k; if (r >= 0 && r < d->M) o[k] = q[k]; }
}

/* Elements of the wave-private LDS window fd_band_emit_wave needs (16-byte aligned):
       __shared__ __attribute__((aligned(16))) T win[WAVES_PER_BLOCK][FD_BAND_WAVE_LDS(W)];                                  */
#define FD_BAND_WAVE_LDS(W) (128 * (W) + 8)

template <typename T> struct fd_band_pair_of;
template <> struct fd_band_pair_of<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct fd_band_pair_of<float> { typedef float type __attribute__((ext_vector_type(2))); };

/*
 * A wavefront's 128 columns at once.  Every lane t of a FULL wavefront (all 64 lanes must call) holds the quotients of the
 * two columns j = jw + 2t and j + 1:  q[k] = (row j - u + k, column j), q[W + k] = (row j + 1 - u + k, column j + 1),
 * k = 0 .. W-1, W = l + u + 1 (compile time; must equal the descriptor's).  jw is even and the same in all lanes.
 * Interior wavefronts (every column complete and inside the local range) place the 128 W values in `win` in storage order
 * and write them as aligned 16-byte (Float32: 8-byte) non-temporal (NT = false: plain) stores, each 128-byte line exactly once; wavefronts that
 * touch a corner of the matrix or an end of the local column range take fd_band_emit_column.  Values of columns outside the
 * local range are ignored, quotients of rows outside the matrix are never read.
 */
template <typename T, int W, bool NT = true>
__device__ inline void fd_band_emit_wave(const fd_band_store *d, T *win, long long jw, const T *q)
{
    typedef typename fd_band_pair_of<T>::type pair_t;
    const int lane = (int)(threadIdx.x & 63);
    const long long j = jw + 2 * lane;
    const long long jlast = jw + 127;
    const bool inside = jw >= d->col_begin && jlast < d->col_end && jlast < d->N;
    bool fast = inside && d->layout != FD_BAND_TRIDIAGONAL && ((((unsigned long long)d->out) & (2 * sizeof(T) - 1)) == 0);
    if (d->layout == FD_BAND_CSC) fast = fast && jw >= d->u && jlast + d->l <= d->M - 1;          /* no column cut by the matrix edge */
    if (d->layout == FD_BAND_BANDED) fast = fast && jw >= d->u && jlast + d->l <= d->M - 1;       /* (corner slots hold zeros) */
    if (W == 3 && d->layout == FD_BAND_TRIDIAGONAL && inside && jw >= 1 && jlast + 1 <= d->M - 1) {
        /* three dense diagonals: d and dl pairs are aligned with the column pair, du is one element behind */
        T *pd = (T *)d->out + (j - d->col_begin), *pl = (T *)d->out_dl + (j - d->col_begin);
        const long long du0 = d->col_begin > 0 ? d->col_begin - 1 : 0;
        T *pu = (T *)d->out_du + (j - du0);                                   /* du[j]: the entry of column j + 1 */
        const T nq0 = __shfl_down(q[0], 1, 64);                               /* du[j + 1]: the next lane's first quotient */
        const T q1 = q[1], q2 = q[2], q3 = q[W], q4 = q[W + 1], q5 = q[W + 2];   /* (W == 3 here) */
        if ((((unsigned long long)pd) & (2 * sizeof(T) - 1)) == 0) __builtin_nontemporal_store(pair_t{q1, q4}, (pair_t *)pd);
        else { pd[0] = q1; pd[1] = q4; }
        if ((((unsigned long long)pl) & (2 * sizeof(T) - 1)) == 0) __builtin_nontemporal_store(pair_t{q2, q5}, (pair_t *)pl);
        else { pl[0] = q2; pl[1] = q5; }
        if (lane == 0) pu[-1] = q[0];
        if (lane < 63 && (((unsigned long long)pu) & (2 * sizeof(T) - 1)) == 0) __builtin_nontemporal_store(pair_t{q3, nq0}, (pair_t *)pu);
        else { pu[0] = q3; if (lane < 63) pu[1] = nq0; }
        return;
    }
    if (!fast) {
        fd_band_emit_column<T>(d, j, q);
        fd_band_emit_column<T>(d, j + 1, q + W);
        return;
    }
    /* local index of the wave's first value, and its parity: slot `off` of the window holds it, so that slot 0 is aligned */
    const long long P0 = d->layout == FD_BAND_BANDED ? (long long)W * (jw - d->col_begin)
                                                     : (long long)W * jw - (long long)d->u * (d->u + 1) / 2 - d->entry_begin;
    const int off = (int)(P0 & 1);
#pragma unroll
    for (int m = 0; m < 2 * W; ++m) win[off + 2 * W * lane + m] = q[m];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    T *base = (T *)d->out + (P0 - off);                                        /* slot 0 <-> an even local index */
#pragma unroll
    for (int a = 0; a < W; ++a) {
        const int sl = 2 * (64 * a + lane);
        const pair_t v = *(const pair_t *)(win + sl);
        if (off && sl == 0) base[1] = v.y;                                     /* slot 0 belongs to the wavefront before */
        else if (NT) __builtin_nontemporal_store(v, (pair_t *)(base + sl));
        else *(pair_t *)(base + sl) = v;
    }
    if (off && lane == 63) base[128 * W] = q[2 * W - 1];                        /* the odd last value */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                                           /* the window may be reused */
}
/*
 * The same for 4-byte elements with FOUR columns per lane: a wavefront's 256 columns, written as 16-byte stores (a Float32 pair
 * store moves 512 B per instruction, half of what the memory pipeline takes).  Lane t of a FULL wavefront holds the quotients of the
 * columns j = jw + 4t .. j + 3:  q[c W + k] = (row j + c - u + k, column j + c), c = 0 .. 3, k = 0 .. W-1.  jw is a multiple of 4
 * and the same in all lanes.  Window: FD_BAND_WAVE4_LDS(W) elements, 16-byte aligned.  Interior wavefronts of the CSC / BANDED
 * layouts with a 16-byte aligned `out` stage and store quads (the first and the last quad of a wavefront whose first value is not
 * on a quad boundary are shared with its neighbours: single elements there); a Tridiagonal's three diagonals are written as quads
 * directly; everything else takes fd_band_emit_column.
 */
#define FD_BAND_WAVE4_LDS(W) (256 * (W) + 8)
template <typename T> struct fd_band_quad_of;
template <> struct fd_band_quad_of<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct fd_band_quad_of<double> { typedef double type __attribute__((ext_vector_type(4))); };
template <typename T, int W, bool NT = true>
__device__ inline void fd_band_emit_wave4(const fd_band_store *d, T *win, long long jw, const T *q)
{
    typedef typename fd_band_quad_of<T>::type quad_t;
    const int lane = (int)(threadIdx.x & 63);
    const long long j = jw + 4 * lane;
    const long long jlast = jw + 255;
    const bool inside = jw >= d->col_begin && jlast < d->col_end && jlast < d->N;
    const unsigned long long qmask = 4 * sizeof(T) - 1;
    bool fast = inside && d->layout != FD_BAND_TRIDIAGONAL && ((((unsigned long long)d->out) & qmask) == 0) &&
                jw >= d->u && jlast + d->l <= d->M - 1;                         /* no column cut by the matrix edge */
    if (W == 3 && d->layout == FD_BAND_TRIDIAGONAL && inside && jw >= 1 && jlast + 1 <= d->M - 1) {
        /* three dense diagonals: d and dl quads are aligned with the column quad, du is one element behind */
        T *pd = (T *)d->out + (j - d->col_begin), *pl = (T *)d->out_dl + (j - d->col_begin);
        const long long du0 = d->col_begin > 0 ? d->col_begin - 1 : 0;
        T *pu = (T *)d->out_du + (j - du0);                                   /* du[j]: the entry of column j + 1 */
        const T nq0 = __shfl_down(q[0], 1, 64);                               /* du[j + 3]: the next lane's first quotient */
        const quad_t vd = {q[1], q[W + 1], q[2 * W + 1], q[3 * W + 1]}, vl = {q[2], q[W + 2], q[2 * W + 2], q[3 * W + 2]};
        const quad_t vu = {q[W], q[2 * W], q[3 * W], nq0};
        if ((((unsigned long long)pd) & qmask) == 0) { if (NT) __builtin_nontemporal_store(vd, (quad_t *)pd); else *(quad_t *)pd = vd; }
        else { pd[0] = vd.x; pd[1] = vd.y; pd[2] = vd.z; pd[3] = vd.w; }
        if ((((unsigned long long)pl) & qmask) == 0) { if (NT) __builtin_nontemporal_store(vl, (quad_t *)pl); else *(quad_t *)pl = vl; }
        else { pl[0] = vl.x; pl[1] = vl.y; pl[2] = vl.z; pl[3] = vl.w; }
        if (lane == 0) pu[-1] = q[0];
        if (lane < 63 && (((unsigned long long)pu) & qmask) == 0) { if (NT) __builtin_nontemporal_store(vu, (quad_t *)pu); else *(quad_t *)pu = vu; }
        else { pu[0] = vu.x; pu[1] = vu.y; pu[2] = vu.z; if (lane < 63) pu[3] = vu.w; }
        return;
    }
    if (!fast) {
#pragma unroll
        for (int c = 0; c < 4; ++c) fd_band_emit_column<T>(d, j + c, q + c * W);
        return;
    }
    /* local index of the wave's first value: slot `off` of the window holds it, so that slot 0 is on a quad boundary */
    const long long P0 = d->layout == FD_BAND_BANDED ? (long long)W * (jw - d->col_begin)
                                                     : (long long)W * jw - (long long)d->u * (d->u + 1) / 2 - d->entry_begin;
    const int off = (int)(P0 & 3);
#pragma unroll
    for (int m = 0; m < 4 * W; ++m) win[off + 4 * W * lane + m] = q[m];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    T *base = (T *)d->out + (P0 - off);                                        /* slot 0 <-> a local index that is a multiple of 4 */
#pragma unroll
    for (int a = 0; a < W; ++a) {
        const int sl = 4 * (64 * a + lane);
        const quad_t v = *(const quad_t *)(win + sl);
        if (off && sl == 0) {                                                  /* the slots before `off` belong to the wavefront before */
            if (off <= 1) base[1] = v.y;
            if (off <= 2) base[2] = v.z;
            base[3] = v.w;
        } else if (NT) __builtin_nontemporal_store(v, (quad_t *)(base + sl));
        else *(quad_t *)(base + sl) = v;
    }
    if (lane < off) base[256 * W + lane] = win[256 * W + lane];                /* the values past the last whole quad */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                                           /* the window may be reused */
}

/* the slots [lo, hi) of a wave-private LDS window, slot 0 <-> the 16-byte aligned element `base`: aligned pair stores, single
   elements at the two ends (all 64 lanes call) */
template <typename T, bool NT> __device__ inline void fd_wave_store_window(T *base, const T *win, int lo, int hi)
{
    typedef typename fd_band_pair_of<T>::type pair_t;
    const int lane = (int)(threadIdx.x & 63);
    for (int sl = 2 * lane; sl < hi; sl += 128) {
        const bool l0 = sl >= lo, l1 = sl + 1 < hi;
        if (l0 && l1) {
            const pair_t v = *(const pair_t *)(win + sl);
            if (NT) __builtin_nontemporal_store(v, (pair_t *)(base + sl));
            else *(pair_t *)(base + sl) = v;
        } else {
            if (l0) base[sl] = win[sl];
            if (l1 && sl + 1 >= lo) base[sl + 1] = win[sl + 1];
        }
    }
}

/* One column of the 5-point stencil: q[0..4] = the quotients of the rows k - nx, k - 1, k, k + 1, k + nx (those that exist). */
template <typename T> __device__ inline void fd_stencil5_emit_column_ij(const fd_stencil5_store *d, long long i, long long j, const T *q)
{
    const long long k = j * d->nx + i;
    if (k < d->col_begin || k >= d->col_end) return;
    T *o = (T *)d->out + (fd_stencil5_colptr_ij(d, i, j) - d->entry_begin);
    if (j > 0) *o++ = q[0];
    if (i > 0) *o++ = q[1];
    *o++ = q[2];
    if (i < d->nx - 1) *o++ = q[3];
    if (j < d->ny - 1) *o++ = q[4];
}
template <typename T> __device__ inline void fd_stencil5_emit_column(const fd_stencil5_store *d, long long k, const T *q)
{
    const long long j = k / d->nx;
    fd_stencil5_emit_column_ij<T>(d, k - j * d->nx, j, q);
}

#define FD_STENCIL5_WAVE_LDS 648   /* elements of the wave-private window of fd_stencil5_emit_wave (16-byte aligned) */

/*
 * 64 * CPL consecutive columns of one grid row at once (CPL = columns per lane, 1 or 2): lane t of a FULL wavefront holds
 * q[0..4] of column (i0 + CPL t, j) and, CPL = 2, q[5..9] of the column after it; i0 is a multiple of CPL and the same in all lanes
 * (lanes whose column lies beyond the grid row are ignored).  Interior grid rows inside the local column range go through the
 * window (the first column of a grid row has no west entry, the last no east entry: their slots close up) and out as dense
 * aligned stores; everything else through fd_stencil5_emit_column.
 */
template <typename T, bool NT = true, int CPL = 2>
__device__ inline void fd_stencil5_emit_wave(const fd_stencil5_store *d, T *win, long long j, long long i0, const T *q)
{
    const int lane = (int)(threadIdx.x & 63);
    const long long nx = d->nx, i = i0 + CPL * lane, k0 = j * nx + i0;
    const int nc = (int)(nx - i0 < 64 * CPL ? nx - i0 : 64 * CPL);
    const bool fast = j >= 1 && j <= d->ny - 2 && k0 >= d->col_begin && k0 + nc <= d->col_end &&
                      ((((unsigned long long)d->out) & (2 * sizeof(T) - 1)) == 0);
    if (!fast) {
#pragma unroll
        for (int o = 0; o < CPL; ++o)
            if (i + o < nx) fd_stencil5_emit_column_ij<T>(d, i + o, j, q + 5 * o);
        return;
    }
    const long long P0 = fd_stencil5_colptr_ij(d, i0, j) - d->entry_begin;
    const int off = (int)(P0 & 1);
    const int cnt = 5 * nc - (i0 == 0 ? 1 : 0) - (i0 + nc == nx ? 1 : 0);
#pragma unroll
    for (int o = 0; o < CPL; ++o) {
        const long long ii = i + o;
        if (ii >= nx) continue;
        const int base = off + 5 * (int)(ii - i0) - ((i0 == 0 && ii > 0) ? 1 : 0);
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            if ((m == 1 && ii == 0) || (m == 3 && ii == nx - 1)) continue;
            int sl = base + m;
            if (ii == 0 && m > 1) sl -= 1;
            if (ii == nx - 1 && m == 4) sl -= 1;
            win[sl] = q[5 * o + m];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    fd_wave_store_window<T, NT>((T *)d->out + (P0 - off), win, off, off + cnt);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
/*
 * The same with FOUR columns per lane (round 6; Float32, where a pair store moves only 512 B per instruction): lane t of a FULL
 * wavefront holds q[5 o + m], o = 0 .. 3, of the columns (i0 + 4 t + o, j); i0 is a multiple of 4 and the same in all lanes.  Interior
 * grid rows inside the local column range with a 16-byte aligned `out` leave as aligned 16-byte stores (single elements at the two
 * ends, which share their quad with the neighbouring wavefronts); everything else through fd_stencil5_emit_column.
 * Window: FD_STENCIL5_WAVE4_LDS elements, 16-byte aligned.
 */
#define FD_STENCIL5_WAVE4_LDS 1288
template <typename T, bool NT = true>
__device__ inline void fd_stencil5_emit_wave4(const fd_stencil5_store *d, T *win, long long j, long long i0, const T *q)
{
    typedef typename fd_band_quad_of<T>::type quad_t;
    const int lane = (int)(threadIdx.x & 63);
    const long long nx = d->nx, i = i0 + 4 * lane, k0 = j * nx + i0;
    const int nc = (int)(nx - i0 < 256 ? nx - i0 : 256);
    const bool fast = j >= 1 && j <= d->ny - 2 && k0 >= d->col_begin && k0 + nc <= d->col_end &&
                      ((((unsigned long long)d->out) & (4 * sizeof(T) - 1)) == 0);
    if (!fast) {
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (i + o < nx) fd_stencil5_emit_column_ij<T>(d, i + o, j, q + 5 * o);
        return;
    }
    const long long P0 = fd_stencil5_colptr_ij(d, i0, j) - d->entry_begin;
    const int off = (int)(P0 & 3);
    const int lo = off, hi = off + 5 * nc - (i0 == 0 ? 1 : 0) - (i0 + nc == nx ? 1 : 0);
    if (i0 > 0 && i0 + nc < nx) {
        /* no column of the tile is the first or the last of its grid row: 20 consecutive slots per lane */
        if (i < nx) {
#pragma unroll
            for (int m = 0; m < 20; ++m) win[off + 20 * lane + m] = q[m];
        }
    } else {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const long long ii = i + o;
            if (ii >= nx) continue;
            const int base = off + 5 * (int)(ii - i0) - ((i0 == 0 && ii > 0) ? 1 : 0);
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                if ((m == 1 && ii == 0) || (m == 3 && ii == nx - 1)) continue;
                int sl = base + m;
                if (ii == 0 && m > 1) sl -= 1;
                if (ii == nx - 1 && m == 4) sl -= 1;
                win[sl] = q[5 * o + m];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    T *base = (T *)d->out + (P0 - off);                  /* slot 0 <-> a local index that is a multiple of 4 */
    for (int sl = 4 * lane; sl < hi; sl += 256) {
        if (sl >= lo && sl + 3 < hi) {
            const quad_t v = *(const quad_t *)(win + sl);
            if (NT) __builtin_nontemporal_store(v, (quad_t *)(base + sl));
            else *(quad_t *)(base + sl) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (sl + e >= lo && sl + e < hi) base[sl + e] = win[sl + e];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
/* One entry of a column-range storage (fd_colrange_store): the value of (row r, column j).  Entries outside the local column
   range or outside the column's stored rows are ignored. */
template <typename T> __device__ inline void fd_colrange_emit(const fd_colrange_store *d, long long j, long long r, T value)
{
    if (j < d->col_begin || j >= d->col_end) return;
    const long long jj = j - d->col_begin, k = r - d->row_first[jj];
    if (k < 0 || k >= d->row_count[jj]) return;
    ((T *)d->out)[d->dest[jj] + k] = value;
}
/* ---- fd_csc_store: the ten-line storing launch for ANY residual that can evaluate one row ---------------------------------------
 * `F` is a callable  T f(long long r, const P &X)  returning row r of the residual at the point X, where X(j) yields coordinate j
 * (call X(j) for every coordinate the row reads, in the order the residual's arithmetic needs).  The kernel hands it the colour's
 * own point, formed exactly as the reference forms it: x[j] + eps_c * (color[j] == c), i.e. x[j] + 0.0 elsewhere (minus point of a
 * central difference: x[j] - eps_c * ..., x[j] - 0.0 == x[j] elsewhere).  One thread per local column.
 *     hipLaunchKernelGGL((fd_csc_store_cols<double, unsigned char, 0, F>), fd_xcd_grid((st.col_end - st.col_begin + 255) / 256), 256, 0, stream,
 *                        f, x, eps, c_lo, c_hi, st);
 * MODE 0: forward (st.fx_base = f(x)), 1: central.  Columns whose colour lies outside [c_lo, c_hi) are skipped (colour chunks /
 * ownership); columns without a colour are written as 0 when c_lo == 0 (fill_matrix!, src/jacobians.jl:530-532). */
template <typename T, typename CT> struct fd_colour_point {
    typedef T value_type;      /* what X(j) yields: a functor generic in it (`typename P::value_type`, `auto`) also serves the complex step */
    const T *x;
    const CT *color;
    int c;          /* 0-based colour of the point */
    T e;            /* step */
    int minus;      /* 0: the plus point x + e m (x + 0.0 elsewhere); 1: the minus point x - e m (x - 0.0 == x elsewhere); 2: x itself */
    __device__ T operator()(long long j) const
    {
        const T v = x[j];
        if (minus == 2) return v;
        const bool hit = (int)color[j] == c;
        return minus ? (hit ? v - e : v) : v + (hit ? e : (T)0);
    }
};
/* the same point when the colouring is known to be valid (fd_csc_store.valid_coloring): only coordinate j is perturbed */
template <typename T> struct fd_column_point {
    typedef T value_type;
    const T *x;
    long long j;
    T e;
    int minus;
    __device__ T operator()(long long i) const
    {
        const T v = x[i];
        if (minus == 2) return v;                        /* (x itself) */
        const bool hit = i == j;
        return minus ? (hit ? v - e : v) : v + (hit ? e : (T)0);
    }
};
/* MI355X dispatches workgroup b to XCD b % 8, and every XCD has its own L2: consecutive workgroups that share inputs (a stencil's
 * neighbouring columns) would pull them into eight L2s.  Launch 8 * ceil(n / 8) workgroups and let workgroup b work on block
 * fd_xcd_block(b, n) -- XCD x then owns the contiguous block range [x * ceil(n / 8), (x + 1) * ceil(n / 8)); blocks >= n do
 * nothing.  A performance assumption only: results never depend on placement. */
__device__ inline long long fd_xcd_block(long long b, long long n) { return (b & 7) * ((n + 7) / 8) + (b >> 3); }
__host__ inline unsigned fd_xcd_grid(long long n) { return (unsigned)(8 * ((n + 7) / 8)); }

/* Where a wavefront's values go.  The entries of 64 consecutive columns are ONE contiguous run of nzval; lane t holds the pieces
 * of column j0 + t, i.e. 8-byte stores at a lane stride of the column length -- partial lines the L2 does not merge well (measured:
 * 1.0 TB/s on the 7-point pattern).  When every lane of the wavefront writes its whole column (no colour chunk skips one) and the
 * run fits, the values are staged in a wave-private LDS window in storage order and leave as dense, aligned 16-byte stores
 * (fd_wave_store_window); otherwise each value is stored directly.  All 64 lanes call begin and flush. */
#ifndef FD_CSC_WAVE_CAP
#define FD_CSC_WAVE_CAP 1024      /* elements of the window per wavefront */
#endif
template <typename T> struct fd_csc_wave_run {
    T *out;
    T *win;
    int q0a;        /* even position that slot 0 of the window stands for */
    int lo, hi;     /* slots that hold values */
    bool staged;
    /* a, b: the lane's entry range (a == b for a lane without a column); writes: the lane will write ALL of [a, b) */
    /* cap: elements of `win` (FD_CSC_WAVE_CAP, or less for a kernel that knows its columns are short) */
    __device__ void begin(T *out_, T *win_, int a, int b, bool writes, int cap = FD_CSC_WAVE_CAP)
    {
        out = out_; win = win_;
        int qmin = a, qmax = b;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int m = __shfl_xor(qmin, o, 64), M = __shfl_xor(qmax, o, 64);
            qmin = m < qmin ? m : qmin;
            qmax = M > qmax ? M : qmax;
        }
        q0a = qmin & ~1;
        lo = qmin - q0a;
        hi = qmax - q0a;
        staged = __all(writes) && hi <= cap && ((((unsigned long long)out_) & (2 * sizeof(T) - 1)) == 0);
    }
    __device__ void put(int q, T v) const
    {
        if (staged) win[q - q0a] = v;
        else out[q] = v;
    }
    template <bool NT> __device__ void flush() const
    {
        if (!staged) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        fd_wave_store_window<T, NT>(out + q0a, win, lo, hi);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
};
template <typename T, int MODE, class F, class P>
__device__ inline void fd_csc_store_column(const F &f, P &X, const fd_csc_store &st, const fd_csc_wave_run<T> &run, int a, int b, T h)
{
    const T *base = (const T *)st.fx_base;
    /* four entries at a time: their row indices first (one round trip), then the rows -- a functor whose loads are unconditional
       lets the four evaluations overlap (measured with the 7-point functor: 432 -> 397 us; the functor's own form matters more:
       loads inside per-lane conditionals are waited for one by one, 1050 -> 432 us, profiles/r04_zz_pattern_store.md) */
    constexpr int U = 4;
    for (int q0 = a; q0 < b; q0 += U) {
        long long r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = st.rowval[q0 + u < b ? q0 + u : b - 1];
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (q0 + u >= b) { v[u] = 0; continue; }
            X.minus = 0;
            const T vp = f(r[u], X);
            T vm, div = h;
            if (MODE == 1) { X.minus = 1; vm = f(r[u], X); div = 2 * h; }
            else if (base) vm = base[r[u]];
            else { X.minus = 2; vm = f(r[u], X); }                            /* (the unperturbed point: FD_LAZY_CAP_STORE_CSC_BASE) */
            v[u] = (vp - vm) / div;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q0 + u < b) run.put(q0 + u, v[u]);
    }
}
template <typename T, typename CT, int MODE, class F>
__global__ void __launch_bounds__(256) fd_csc_store_cols(F f, const T *__restrict__ x, const T *__restrict__ eps, int c_lo, int c_hi, fd_csc_store st)
{
    __shared__ __attribute__((aligned(16))) T s_win[256 / 64][FD_CSC_WAVE_CAP];
    const long long nblk = (st.col_end - st.col_begin + 255) / 256, blk = fd_xcd_block(blockIdx.x, nblk);      /* launch fd_xcd_grid(nblk) workgroups */
    if (blk >= nblk) return;
    const long long j = st.col_begin + blk * 256 + threadIdx.x;
    const bool in = j < st.col_end;
    const int a = in ? st.colptr[j - st.col_begin] : st.colptr[st.col_end - st.col_begin];
    const int b = in ? st.colptr[j - st.col_begin + 1] : a;
    const CT *color = (const CT *)st.color;
    const int c = in ? (int)color[j] : 0;
    const bool none = in && c == (int)(CT)(-1);                           /* "none" is all-ones in CT */
    const bool mine = in && !none && c >= c_lo && c < c_hi;
    fd_csc_wave_run<T> run;
    run.begin((T *)st.out, s_win[threadIdx.x >> 6], a, b, !in || mine || (none && c_lo == 0));
    if (none && c_lo == 0)
        for (int q = a; q < b; ++q) run.put(q, (T)0);
    if (mine) {
        const T h = eps[c];
        if (st.valid_coloring) {
            fd_column_point<T> X = {x, j, h, 0};
            fd_csc_store_column<T, MODE>(f, X, st, run, a, b, h);
        } else {
            fd_colour_point<T, CT> X = {x, color, c, h, 0};
            fd_csc_store_column<T, MODE>(f, X, st, run, a, b, h);
        }
    }
    run.template flush<true>();
}

/* ---- the same launch with the workgroup's window of x in LDS (round 5) -------------------------------------------------------------
 * fd_csc_store_cols spends its time in the address path: every stored entry evaluates one row, every row reads a handful of
 * coordinates -- ~20 scattered 8-byte loads per entry, each to an address some neighbouring lane also reads (random band, 6 entries
 * per column: 283 us for 246 MB, 0.11 of the peak).  When the pattern is locally banded (st.reach known and small) the coordinates a
 * workgroup's rows read lie in ONE short window of x: it is loaded once, coalesced, into LDS and X(i) is served from there; f(x) of
 * the touched rows (forward differences) likewise.  Coordinates outside the window are read from memory as before -- the window is a
 * cache, never a requirement, so ANY functor and pattern give the bits of fd_csc_store_cols.
 * A functor may keep its own per-row data in LDS too: if it has a member
 *     Staged stage(FD_LDS_PTR(unsigned char) lds, long long r_lo, long long r_hi, long long w0, long long w1, int cap) const   (device; all 256 threads;
 *                  [w0, w1) = the staged window of x: X.at(i - w0) serves coordinate i from it)
 * the kernel calls it with `stage_bytes` bytes of LDS (16-byte aligned) for the rows [r_lo, r_hi) its columns can touch and uses the
 * returned object (same call operator) instead of the functor; the launcher sizes stage_bytes / cap.
 * Dynamic LDS: fd_csc_win_lds_bytes<T>(reach, forward_with_base) + stage_bytes.  Launch as fd_csc_store_cols; needs
 * st.valid_coloring (one perturbed coordinate per column) -- the launcher falls back to fd_csc_store_cols otherwise. */
#ifndef FD_CSC_WIN_WAVE_CAP
#define FD_CSC_WIN_WAVE_CAP 512
#endif
/* pointers into LDS carry their address space: a generic pointer makes the compiler emit flat_load -- the memory pipeline with an
   aperture check -- where ds_read is meant (measured: the first form of this kernel issued 140 flat loads and 3 LDS reads) */
#define FD_LDS_PTR(T) __attribute__((address_space(3))) T *
template <typename T> struct fd_window_column_point {
    typedef T value_type;
    const T *x;         /* global x */
    const FD_LDS_PTR(T) wx;        /* LDS copy of x[w0, w1) */
    long long w0, w1;
    long long j;        /* the perturbed coordinate */
    T e;
    int minus;
    unsigned joff;      /* j - w0 if j lies in the window, else 0xFFFFFFFF */
    __device__ T operator()(long long i) const
    {
        const T v = (i >= w0 && i < w1) ? wx[i - w0] : x[i];
        if (minus == 2) return v;
        const bool hit = i == j;
        return minus ? (hit ? v - e : v) : v + (hit ? e : (T)0);
    }
    /* coordinate w0 + off, known to lie in the window (a functor that staged its indices as window offsets: 32-bit arithmetic only) */
    __device__ T at(unsigned off) const
    {
        const T v = wx[off];
        if (minus == 2) return v;
        const bool hit = off == joff;
        return minus ? (hit ? v - e : v) : v + (hit ? e : (T)0);
    }
};
template <class F, class = void> struct fd_has_stage { static constexpr bool value = false; };
template <class F> struct fd_has_stage<F, decltype((void)&F::stage, void())> { static constexpr bool value = true; };
__host__ __device__ inline long long fd_csc_win_xlen(long long reach) { return 256 + 4 * reach + 2; }       /* elements of the x window (even start) */
__host__ __device__ inline long long fd_csc_win_rlen(long long reach) { return 256 + 2 * reach; }           /* rows a workgroup's columns can touch */
template <typename T> __host__ __device__ inline size_t fd_csc_win_lds_bytes(long long reach, bool with_base)
{
    return sizeof(T) * (size_t)(4 * FD_CSC_WIN_WAVE_CAP + fd_csc_win_xlen(reach) + (with_base ? fd_csc_win_rlen(reach) + 1 : 0)) + 16;
}
/* a / b for a divisor shared by several quotients, y = 1 / b computed once: one multiplication and two FMA correction steps give the
   CORRECTLY ROUNDED quotient (Markstein: q1 is a faithful rounding of a / b, so q2 = RN(a / b) when nothing over- or underflows;
   zero, tiny, huge and non-finite operands take the true division) -- the bits of IEEE a / b at about a third of its instructions
   (scripts/ubench/exact_div_probe.hip: 5e10 random and next-to-tie operand pairs, 0 mismatches).  Float64; Float32 divides. */
/* the same with the divisor's range tested ONCE by the caller (b_ok = fd_div_shared_ok(b)): |b| in [2^-100, 2^100] and |a| in
   [2^-800, 2^800] put the first quotient inside [2^-900, 2^900] by themselves -- two comparisons per quotient instead of four */
template <typename T> __device__ inline bool fd_div_shared_ok(T b)
{
    const T mb = b < 0 ? -b : b;
    return mb >= (T)0x1p-100 && mb <= (T)0x1p100;
}
template <typename T> __device__ inline T fd_div_shared(T a, T b, T y, bool b_ok)
{
    if constexpr (sizeof(T) == 8) {
        const double ma = __builtin_fabs(a);
        if (!(b_ok && ma >= 0x1p-800 && ma <= 0x1p800)) return a / b;
        const double q0 = a * y;
        const double r0 = __builtin_fma(-b, q0, a);
        const double q1 = __builtin_fma(r0, y, q0);
        const double r1 = __builtin_fma(-b, q1, a);
        return __builtin_fma(r1, y, q1);
    } else {
        return a / b;
    }
}
template <typename T> __device__ inline T fd_div_shared(T a, T b, T y)
{
    if constexpr (sizeof(T) == 8) {
        const double q0 = a * y;
        const double m = __builtin_fabs(q0), ma = __builtin_fabs(a);
        if (!(m >= 0x1p-900 && m <= 0x1p900 && ma >= 0x1p-900 && ma <= 0x1p900)) return a / b;
        const double r0 = __builtin_fma(-b, q0, a);
        const double q1 = __builtin_fma(r0, y, q0);
        const double r1 = __builtin_fma(-b, q1, a);
        return __builtin_fma(r1, y, q1);
    } else {
        return a / b;
    }
}
template <typename T, int MODE, class F, class P>
__device__ inline void fd_csc_store_column_win(const F &f, P &X, const fd_csc_store &st, const fd_csc_wave_run<T> &run, int a, int b, T h,
                                               const FD_LDS_PTR(T) wb, long long r_lo, long long r_hi)
{
    const T *base = (const T *)st.fx_base;
    const T dv = MODE == 1 ? 2 * h : h, yd = (T)1 / dv;          /* every quotient of the column has this divisor */
    /* the row indices of up to eight entries in one round trip, then the entries two at a time (the rows come out of LDS: short
       chains; a deeper unroll only grows the code) */
    constexpr int U = 8;
    for (int q0 = a; q0 < b; q0 += U) {
        int r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = st.rowval[q0 + u < b ? q0 + u : b - 1];
#pragma unroll 1
        for (int u0 = 0; u0 < U && q0 + u0 < b; u0 += 2) {
            T v[2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int u = u0 + d;
                const long long rr = u == 0 ? r[0] : u == 1 ? r[1] : u == 2 ? r[2] : u == 3 ? r[3] : u == 4 ? r[4] : u == 5 ? r[5] : u == 6 ? r[6] : r[7];
                if (q0 + u >= b) { v[d] = 0; continue; }
                X.minus = 0;
                const T vp = f(rr, X);
                T vm;
                if (MODE == 1) { X.minus = 1; vm = f(rr, X); }
                else if (base) vm = (rr >= r_lo && rr < r_hi) ? wb[rr - r_lo] : base[rr];
                else { X.minus = 2; vm = f(rr, X); }
                v[d] = fd_div_shared<T>(vp - vm, dv, yd);
            }
            run.put(q0 + u0, v[0]);
            if (q0 + u0 + 1 < b) run.put(q0 + u0 + 1, v[1]);
        }
    }
}
template <typename T, typename CT, int MODE, class F>
__global__ void __launch_bounds__(256) fd_csc_store_cols_win(F f, const T *__restrict__ x, const T *__restrict__ eps, int c_lo, int c_hi, fd_csc_store st,
                                                             int reach, int stage_bytes, int stage_cap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fd_csc_lds[];
    T *s_win = (T *)fd_csc_lds;                                        /* 4 wave windows of the output */
    FD_LDS_PTR(T) s_x = (FD_LDS_PTR(T))(s_win + 4 * FD_CSC_WIN_WAVE_CAP);     /* x[w0, w1) */
    const long long nblk = (st.col_end - st.col_begin + 255) / 256, blk = fd_xcd_block(blockIdx.x, nblk);      /* launch fd_xcd_grid(nblk) workgroups */
    if (blk >= nblk) return;
    const long long j0 = st.col_begin + blk * 256, jn = j0 + 256 < st.col_end ? j0 + 256 : st.col_end;
    const long long r_lo = j0 - reach > 0 ? j0 - reach : 0, r_hi = jn + reach < st.M ? jn + reach : st.M;
    long long w0 = r_lo - reach > 0 ? r_lo - reach : 0, w1 = r_hi + reach < st.N ? r_hi + reach : st.N;
    w0 &= ~1ll;
    const bool with_base = MODE == 0 && st.fx_base != nullptr;
    FD_LDS_PTR(T) s_b = s_x + fd_csc_win_xlen(reach);                  /* f(x)[r_lo, r_hi) */
    const unsigned f_off = (unsigned)(sizeof(T) * (size_t)(4 * FD_CSC_WIN_WAVE_CAP + fd_csc_win_xlen(reach) + (with_base ? fd_csc_win_rlen(reach) + 1 : 0)) + 15) & ~15u;
    FD_LDS_PTR(unsigned char) s_f = (FD_LDS_PTR(unsigned char))fd_csc_lds + f_off;      /* the functor's own staging area, 16-byte aligned */
    /* coalesced fills, every load of a batch issued before the first is used (a loop of load-then-store costs one memory round trip
       per iteration and workgroup -- the first form of this kernel spent 30 of them): x as 16-byte pairs (w0 is even, x 16-byte
       aligned), f(x) of the rows element by element */
    {
        typedef T fd_pair_t __attribute__((ext_vector_type(2)));
        const long long nx = w1 - w0, npair = nx / 2, nb = with_base ? r_hi - r_lo : 0;
        const T *base = (const T *)st.fx_base;
        for (long long i0 = 0; i0 < npair || i0 < nb; i0 += 4 * 256) {
            fd_pair_t vx[4];
            T vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long i = i0 + u * 256 + threadIdx.x;
                if (i < npair) vx[u] = *reinterpret_cast<const fd_pair_t *>(x + w0 + 2 * i);
                if (i < nb) vb[u] = base[r_lo + i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long i = i0 + u * 256 + threadIdx.x;
                if (i < npair) { s_x[2 * i] = vx[u].x; s_x[2 * i + 1] = vx[u].y; }
                if (i < nb) s_b[i] = vb[u];
            }
        }
        if ((nx & 1) && threadIdx.x == 0) s_x[nx - 1] = x[w0 + nx - 1];
    }
    const long long j = j0 + threadIdx.x;
    const bool in = j < st.col_end;
    const int a = in ? st.colptr[j - st.col_begin] : st.colptr[st.col_end - st.col_begin];
    const int b = in ? st.colptr[j - st.col_begin + 1] : a;
    const CT *color = (const CT *)st.color;
    const int c = in ? (int)color[j] : 0;
    const bool none = in && c == (int)(CT)(-1);                           /* "none" is all-ones in CT */
    const bool mine = in && !none && c >= c_lo && c < c_hi;
    fd_csc_wave_run<T> run;
    run.begin((T *)st.out, s_win + (threadIdx.x >> 6) * FD_CSC_WIN_WAVE_CAP, a, b, !in || mine || (none && c_lo == 0), FD_CSC_WIN_WAVE_CAP);
    if (none && c_lo == 0)
        for (int q = a; q < b; ++q) run.put(q, (T)0);
    const T h = mine ? eps[c] : (T)1;
    fd_window_column_point<T> X = {x, s_x, w0, w1, j, h, 0, (j >= w0 && j < w1) ? (unsigned)(j - w0) : 0xFFFFFFFFu};
    if constexpr (fd_has_stage<F>::value) {
        const auto fs = f.stage(s_f, r_lo, r_hi, w0, w1, stage_cap);
        __syncthreads();
        if (mine) fd_csc_store_column_win<T, MODE>(fs, X, st, run, a, b, h, s_b, r_lo, r_hi);
    } else {
        __syncthreads();
        if (mine) fd_csc_store_column_win<T, MODE>(f, X, st, run, a, b, h, s_b, r_lo, r_hi);
    }
    run.template flush<true>();
}

/* ---- SEPARABLE residuals: the Jacobian stored ROW BY ROW (round 6) ---------------------------------------------------------------------
 * The column kernels above give every stored entry (r, j) its own evaluation of row r: nnz x (row length) terms.  Many residuals on a
 * general pattern are sums of ONE-coordinate terms over the pattern's own entries (a graph Laplacian with nonlinear edge terms, a reaction
 * network, a finite-volume flux sum):
 *     f(x)_r = t(r, j_0, x[j_0]) + t(r, j_1, x[j_1]) + ...      over the stored entries (r, j_k) of J's row r, j_0 < j_1 < ..., added left to right.
 * Such a residual is SEPARABLE on the pattern, and its functor says so:
 *     struct MyTerms { static constexpr bool fd_separable = true;  <parameters>
 *                      template <class T> __device__ T term(long long r, long long j, T v) const { ... } };
 * Then the L entries of a row share everything but one term: the thread that owns row r forms the row's L plain terms once, and for
 * entry k the sum (t_0 + .. + t_{k-1}) + t'_k + t_{k+1} + .. + t_{L-1} with the perturbed term t'_k = t(r, j_k, x[j_k] +- eps) -- the
 * additions of the full evaluation at the colour's point (ext/FiniteDiffSparseArraysExt.jl:38-47 on the f! values of
 * src/jacobians.jl:562-568 / 602-609) in the same order, so THE SAME BITS as fd_csc_store_cols, with 2 L term evaluations per row
 * instead of L^2.  Needs a colouring the plan has verified (st.valid_coloring) and the plan's row lists (st.row_ptr: FD_PLAN_STORE_CSC_ROWS).
 * fd_sep_rows<TF> makes the row functor the other kernels take (plain evaluation, column store, complex step) out of the same terms and
 * the same lists, so one `term` serves every route.
 * Launch: fd_xcd_grid((M + 255) / 256) workgroups of 256 threads, fd_csc_rows_lds_bytes<T>(reach, c_hi - c_lo, cap) bytes of LDS,
 * reach = st.reach (1 .. 700), M == N >= 2, every column local, cap = entries of a 256-row tile kept in LDS (a tile with more reads the
 * rest from memory).  The window of x and of the colours, the step sizes and the tile's run of the row lists are requested up front --
 * every load of the prologue in flight at once -- and parked in LDS; the loops after the barrier touch memory only to store. */
template <class TF> struct fd_sep_rows {
    static constexpr bool fd_separable = true;
    TF t;
    const int *row_ptr, *row_col;
    template <class T> __device__ T term(long long r, long long j, T v) const { return t.term(r, j, v); }
    template <class P> __device__ typename P::value_type operator()(long long r, const P &X) const
    {
        typedef typename P::value_type V;
        const int a = row_ptr[r], b = row_ptr[r + 1];
        V s = V();
        /* four entries at a time: their columns in one round trip (from positions clamped into the row), their coordinates in a second */
        for (int k0 = a; k0 < b; k0 += 4) {
            long long j[4];
            V v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) j[u] = row_col[k0 + u < b ? k0 + u : b - 1];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = X(j[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const V w = t.term(r, j[u], v[u]);
                if (k0 + u < b) s = k0 + u == a ? w : s + w;
            }
        }
        return s;
    }
};
template <class F, class = void> struct fd_is_separable { static constexpr bool value = false; };
template <class F> struct fd_is_separable<F, decltype((void)F::fd_separable, void())> { static constexpr bool value = F::fd_separable; };
/* where fd_sep_rows keeps its two list pointers (a host that assembles the functor object byte for byte needs the offset) */
template <class F> struct fd_sep_rows_layout { static constexpr unsigned lists_offset = 0, terms_bytes = 0; };
template <class TF> struct fd_sep_rows_layout<fd_sep_rows<TF>> {
    static constexpr unsigned lists_offset = (unsigned)__builtin_offsetof(fd_sep_rows<TF>, row_ptr), terms_bytes = (unsigned)sizeof(TF);
};
#define FD_CSC_ROWS_REGS 14        /* entries of a row fd_csc_store_rows keeps in registers (longer rows loop over LDS) */
/* LDS: [x window][step, reciprocal per colour of the batch][column, slot of the tile's entries][colours of the window] */
template <typename T> __host__ __device__ inline size_t fd_csc_rows_lds_bytes(long long reach, int ncolors, int cap)
{
    const size_t xlen = (size_t)(256 + 2 * reach + 2);
    return sizeof(T) * (xlen + 2 * (size_t)ncolors) + 4 * (2 * (size_t)cap + xlen) + 64;
}
/* one row of 1 .. RL entries in REGISTERS, every loop unrolled and predicated */
template <typename T, int MODE, int RL, class F>
__device__ __attribute__((always_inline)) inline void fd_csc_rows_regs(const F &f, long long r, int b0, int L, const FD_LDS_PTR(int) s_j, const FD_LDS_PTR(int) s_q, const FD_LDS_PTR(T) s_x,
                                        const FD_LDS_PTR(int) s_c, const FD_LDS_PTR(T) s_h, const FD_LDS_PTR(T) s_y, int w0, int c_lo, int c_hi, T *out,
                                        bool given, T fx_given)
{
    int jj[RL], qq[RL];
    T tt[RL];
#pragma unroll
    for (int u = 0; u < RL; ++u) { const int i = b0 + (u < L ? u : L - 1); jj[u] = s_j[i]; qq[u] = s_q[i]; }
    T fx = 0;
#pragma unroll
    for (int u = 0; u < RL; ++u) {
        tt[u] = f.term(r, (long long)jj[u], (T)s_x[(unsigned)(jj[u] - w0)]);
        fx = u == 0 ? tt[0] : (u < L ? fx + tt[u] : fx);
    }
    T pre = 0;
#pragma unroll
    for (int k = 0; k < RL; ++k) {
        if (k < L) {
            const unsigned off = (unsigned)(jj[k] - w0);
            const int c = s_c[off];
            const T v = s_x[off];
            if (c < 0) {
                if (c_lo == 0) out[qq[k]] = (T)0;
            } else if (c >= c_lo && c < c_hi) {
                const T h = s_h[c - c_lo], y = s_y[c - c_lo];
                T sp = f.term(r, (long long)jj[k], v + h), sm = MODE == 1 ? f.term(r, (long long)jj[k], v - h) : (T)0;
                if (k > 0) { sp = pre + sp; if (MODE == 1) sm = pre + sm; }
#pragma unroll
                for (int u = k + 1; u < RL; ++u)
                    if (u < L) { sp = sp + tt[u]; if (MODE == 1) sm = sm + tt[u]; }
                out[qq[k]] = fd_div_shared<T>(sp - (MODE == 1 ? sm : given ? fx_given : fx), MODE == 1 ? 2 * h : h, y);
            }
            pre = k == 0 ? tt[0] : pre + tt[k];
        }
    }
}
template <typename T, typename CT, int MODE, class F>
__global__ void __launch_bounds__(256, 4) fd_csc_store_rows(F f, const T *__restrict__ x, const T *__restrict__ eps, int c_lo, int c_hi, fd_csc_store st, int reach,
                                                         int cap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fd_rows_lds[];
    typedef T fd_pair_t __attribute__((ext_vector_type(2)));
    const long long ntile = (st.M + 255) / 256, tile = fd_xcd_block(blockIdx.x, ntile);
    if (tile >= ntile) return;
    const long long R0 = tile * 256, R1 = R0 + 256 < st.M ? R0 + 256 : st.M;
    /* thread -> row through the plan's order (the tile's rows by descending length: the 64 rows of a wavefront are alike, and the
       unrolled row code below is taken in the size the wavefront's longest row needs); which quarter of the order a wavefront takes
       rotates with the tile, so that the long rows do not always land on the same SIMD */
    const long long pos = R0 + (((int)threadIdx.x + 64 * (int)(tile & 3)) & 255);
    const bool in = pos < R1;
    typedef int fd_int2_t __attribute__((ext_vector_type(2)));
    const fd_int2_t pk = *reinterpret_cast<const fd_int2_t *>(st.row_pack + 2 * (in ? pos : R0));      /* (one load, no chain through row_ptr) */
    const long long r = pk.x;
    const int A0 = st.row_tile[tile], A1 = st.row_tile[tile + 1];
    long long w0 = R0 - reach > 0 ? R0 - reach : 0;
    const long long w1 = R1 + reach < st.N ? R1 + reach : st.N;
    w0 &= ~1ll;
    const int xlen = 256 + 2 * reach + 2, nchunk = c_hi - c_lo;
    FD_LDS_PTR(T) s_x = (FD_LDS_PTR(T))fd_rows_lds;
    FD_LDS_PTR(T) s_h = s_x + xlen;                                    /* step of colour c_lo + i */
    FD_LDS_PTR(T) s_y = s_h + nchunk;                                  /* 1 / (step or 2 step) */
    FD_LDS_PTR(int) s_j = (FD_LDS_PTR(int))(s_y + nchunk);             /* column of the tile's i-th entry */
    FD_LDS_PTR(int) s_q = s_j + cap;                                   /* its slot in out */
    FD_LDS_PTR(int) s_c = s_q + cap;                                   /* colour of column w0 + i (-1: none) */
    const CT *color = (const CT *)st.color;
    const int nx = (int)(w1 - w0), npair = nx >> 1;
    const int nst = A1 - A0 < cap ? A1 - A0 : cap;
    if (nst <= 0) return;                                              /* (a tile of empty rows) */
    {
        /* STRAIGHT-LINE: every load unconditional, from an index clamped into range (a batch the tile does not need re-reads its last
           element), all in flight together and waited for once -- a load inside a branch, even a wave-uniform one, is waited for
           with everything outstanding at the first use of its value (see fd_csc_store_ents) */
        constexpr int XP = 4, XC = 8, XL = 8;      /* 2048 coordinates, 2048 colours, 2048 entries */
        const int tid = (int)threadIdx.x;
        fd_pair_t vx[XP];
        CT vc[XC];
        int vj[XL], vq[XL];
#pragma unroll
        for (int u = 0; u < XP; ++u) { const int i = u * 256 + tid; vx[u] = *reinterpret_cast<const fd_pair_t *>(x + w0 + 2 * (i < npair ? i : npair - 1)); }
#pragma unroll
        for (int u = 0; u < XC; ++u) { const int i = u * 256 + tid; vc[u] = color[w0 + (i < nx ? i : nx - 1)]; }
#pragma unroll
        for (int u = 0; u < XL; ++u) { const int i = u * 256 + tid, ic = A0 + (i < nst ? i : nst - 1); vj[u] = st.row_col[ic]; vq[u] = st.row_slot[ic]; }
        const T hmine = eps[c_lo + (tid < nchunk ? tid : 0)];
#pragma unroll
        for (int u = 0; u < XP; ++u) { const int i = u * 256 + tid; if (i < npair) { s_x[2 * i] = vx[u].x; s_x[2 * i + 1] = vx[u].y; } }
        if ((nx & 1) && tid == 0) s_x[nx - 1] = x[w0 + nx - 1];
#pragma unroll
        for (int u = 0; u < XC; ++u) { const int i = u * 256 + tid; if (i < nx) s_c[i] = vc[u] == (CT)(-1) ? -1 : (int)vc[u]; }
#pragma unroll
        for (int u = 0; u < XL; ++u) { const int i = u * 256 + tid; if (i < nst) { s_j[i] = vj[u]; s_q[i] = vq[u]; } }
        if (tid < nchunk) { s_h[tid] = hmine; s_y[tid] = (T)1 / (MODE == 1 ? 2 * hmine : hmine); }
        for (int i = 256 + tid; i < nchunk; i += 256) { const T h = eps[c_lo + i]; s_h[i] = h; s_y[i] = (T)1 / (MODE == 1 ? 2 * h : h); }      /* (more than 256 colours in a batch) */
        for (int i0 = XL * 256; i0 < nst; i0 += 256) {                  /* (a tile with more than 2048 staged entries: the rest, trip by trip) */
            const int i = i0 + tid;
            if (i < nst) { s_j[i] = st.row_col[A0 + i]; s_q[i] = st.row_slot[A0 + i]; }
        }
    }
    __syncthreads();
    int b0 = pk.y & 0xFFFF, L = in ? (int)((unsigned)pk.y >> 16) : 0;   /* the row's entries are the tile's [b0, b0 + L) */
    T *out = (T *)st.out;
    /* forward differences: the subtrahend is what the plan hands over (the caller's f_in, or f(x) from a plain evaluation) -- or,
       without one (FD_LAZY_CAP_STORE_CSC_BASE), the row's own plain sum */
    const bool given = MODE == 0 && st.fx_base != nullptr;
    const T fx_given = (given && in) ? ((const T *)st.fx_base)[r] : (T)0;
    if (A1 - A0 <= cap) {
        /* the whole tile is staged, and every column of a row lies in the window (|r - j| <= reach): no load from memory below.  That
           matters more than it looks: this target counts loads and stores in ONE counter, and a loop body that MAY load makes the
           compiler wait for everything outstanding -- the store of the iteration before -- on every trip */
        int Lmax = L;                                                  /* the wavefront's longest row (wave-uniform) */
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(Lmax, o, 64); Lmax = t > Lmax ? t : Lmax; }
        Lmax = __builtin_amdgcn_readfirstlane(Lmax);
        if (Lmax <= FD_CSC_ROWS_REGS) {
#define FD_ROWS_REGS(RL) do { if (L > 0) fd_csc_rows_regs<T, MODE, RL>(f, r, b0, L, s_j, s_q, s_x, s_c, s_h, s_y, (int)w0, c_lo, c_hi, out, given, fx_given); } while (0)
            if (Lmax <= 4) FD_ROWS_REGS(4);
            else if (Lmax <= 6) FD_ROWS_REGS(6);
            else if (Lmax <= 8) FD_ROWS_REGS(8);
            else if (Lmax <= 10) FD_ROWS_REGS(10);
            else FD_ROWS_REGS(FD_CSC_ROWS_REGS);
#undef FD_ROWS_REGS
            return;
        }
        if (L == 0) return;
        T fx = 0;
        for (int u = 0; u < L; ++u) {
            const int j = s_j[b0 + u];
            const T t = f.term(r, (long long)j, (T)s_x[(unsigned)(j - (int)w0)]);
            fx = u == 0 ? t : fx + t;
        }
        T pre = 0;
        for (int k = 0; k < L; ++k) {
            const int j = s_j[b0 + k], q = s_q[b0 + k];
            const unsigned off = (unsigned)(j - (int)w0);
            const int c = s_c[off];
            const T v = s_x[off];
            const T tk = f.term(r, (long long)j, v);
            if (c < 0) {
                if (c_lo == 0) out[q] = (T)0;
            } else if (c >= c_lo && c < c_hi) {
                const T h = s_h[c - c_lo], y = s_y[c - c_lo];
                T sp = f.term(r, (long long)j, v + h), sm = MODE == 1 ? f.term(r, (long long)j, v - h) : (T)0;
                if (k > 0) { sp = pre + sp; if (MODE == 1) sm = pre + sm; }
                for (int u = k + 1; u < L; ++u) {
                    const int ju = s_j[b0 + u];
                    const T t = f.term(r, (long long)ju, (T)s_x[(unsigned)(ju - (int)w0)]);
                    sp = sp + t;
                    if (MODE == 1) sm = sm + t;
                }
                out[q] = fd_div_shared<T>(sp - (MODE == 1 ? sm : given ? fx_given : fx), MODE == 1 ? 2 * h : h, y);
            }
            pre = k == 0 ? tk : pre + tk;
        }
        return;
    }
    /* a tile with more entries than fit: the lists beyond the staged part are read from memory (same indices, same order, same bits);
       the packed place / length may be clamped here: the row's own offsets say */
    if (!in) return;
    b0 = st.row_ptr[r] - A0;
    L = st.row_ptr[r + 1] - st.row_ptr[r];
    if (L == 0) return;
    auto col_of = [&](int i) -> int { return i < nst ? (int)s_j[i] : st.row_col[A0 + i]; };
    auto slot_of = [&](int i) -> int { return i < nst ? (int)s_q[i] : st.row_slot[A0 + i]; };
    T fx = 0;
    for (int u = 0; u < L; ++u) {
        const int j = col_of(b0 + u);
        const T t = f.term(r, (long long)j, (T)s_x[(unsigned)(j - (int)w0)]);
        fx = u == 0 ? t : fx + t;
    }
    T pre = 0;
    for (int k = 0; k < L; ++k) {
        const int j = col_of(b0 + k), q = slot_of(b0 + k);
        const unsigned off = (unsigned)(j - (int)w0);
        const int c = s_c[off];
        const T v = s_x[off];
        const T tk = f.term(r, (long long)j, v);
        if (c < 0) {
            if (c_lo == 0) out[q] = (T)0;
        } else if (c >= c_lo && c < c_hi) {
            const T h = s_h[c - c_lo], y = s_y[c - c_lo];
            T sp = f.term(r, (long long)j, v + h), sm = MODE == 1 ? f.term(r, (long long)j, v - h) : (T)0;
            if (k > 0) { sp = pre + sp; if (MODE == 1) sm = pre + sm; }
            for (int u = k + 1; u < L; ++u) {
                const int ju = col_of(b0 + u);
                const T t = f.term(r, (long long)ju, (T)s_x[(unsigned)(ju - (int)w0)]);
                sp = sp + t;
                if (MODE == 1) sm = sm + t;
            }
            out[q] = fd_div_shared<T>(sp - (MODE == 1 ? sm : given ? fx_given : fx), MODE == 1 ? 2 * h : h, y);
        }
        pre = k == 0 ? tk : pre + tk;
    }
}

/* ---- the same store, a thread per ENTRY (round 6; measured EQUAL to the row form on the random band -- neither is bound by its arithmetic,
 *      profiles/NOTES.md -- so the plans build its lists only on request: test switch FDJAC_ROWS_ENTS=1) ------------------------------------------------------------------------------------------
 * fd_csc_store_rows gives a thread a ROW: the lanes of a wavefront idle while its longest row is worked through, and the unrolled row
 * code wants 120 registers.  Here a thread takes ENTRIES -- tile t's entries in the plan's order of the rows (descending length:
 * fd_csc_store.ent_*), entry i = u 256 + thread -- so that the 64 lanes of a wavefront hold entries of rows of one length:
 *   1. every entry's plain term t(r, j, x_j) once, into LDS;
 *   2. every entry (r, j_k) then adds its row's terms left to right with its own replaced by t(r, j_k, x_j +- eps) -- the L - 1
 *      additions of the full evaluation at the colour's point, the same in every lane of the wavefront (and, for forward differences
 *      without a given f(x), the row's plain sum beside them) -- divides and stores.
 * Same additions in the same order as fd_csc_store_rows / fd_csc_store_cols: same bits; no register arrays (twice the resident
 * wavefronts).  Needs st.ent_* (every tile <= 2048 entries, rows <= 255 entries) on top of what fd_csc_store_rows needs; same launch
 * shape; LDS fd_csc_ents_lds_bytes<T>(reach, c_hi - c_lo, st.ent_tile_max).  (A budget of six wavefronts per SIMD -- 76 registers, no spill -- measured
 * SLOWER: 103-110 us against 98-101.) */
#define FD_CSC_ENTS_PER_THREAD 8
template <typename T> __host__ __device__ inline size_t fd_csc_ents_lds_bytes(long long reach, int ncolors, int tile_max)
{
    const size_t xlen = (size_t)(256 + 2 * reach + 2), nt = ((size_t)tile_max + 63) & ~(size_t)63;      /* (as the kernel lays it out) */
    return sizeof(T) * (xlen + 2 * (size_t)ncolors + nt) + 4 * (xlen + 256) + 64;
}
template <typename T, typename CT, int MODE, class F>
__global__ void __launch_bounds__(256) fd_csc_store_ents(F f, const T *__restrict__ x, const T *__restrict__ eps, int c_lo, int c_hi, fd_csc_store st, int reach)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fd_ents_lds[];
    typedef T fd_pair_t __attribute__((ext_vector_type(2)));
    constexpr int NE = FD_CSC_ENTS_PER_THREAD;
    const long long ntile = (st.M + 255) / 256, tile = fd_xcd_block(blockIdx.x, ntile);
    if (tile >= ntile) return;
    const long long R0 = tile * 256, R1 = R0 + 256 < st.M ? R0 + 256 : st.M;
    const int A0 = st.row_tile[tile], nst = st.row_tile[tile + 1] - A0;
    if (nst <= 0) return;                                              /* (a tile of empty rows) */
    long long w0l = R0 - reach > 0 ? R0 - reach : 0;
    const long long w1 = R1 + reach < st.N ? R1 + reach : st.N;
    w0l &= ~1ll;
    const int w0 = (int)w0l, xlen = 256 + 2 * reach + 2, nchunk = c_hi - c_lo;
    FD_LDS_PTR(T) s_x = (FD_LDS_PTR(T))fd_ents_lds;
    FD_LDS_PTR(T) s_h = s_x + xlen;                                    /* step of colour c_lo + i */
    FD_LDS_PTR(T) s_y = s_h + nchunk;                                  /* 1 / (step or 2 step) */
    FD_LDS_PTR(T) s_t = s_y + nchunk;                                  /* plain term of the tile's i-th entry */
    FD_LDS_PTR(int) s_c = (FD_LDS_PTR(int))(s_t + ((st.ent_tile_max + 63) & ~63));      /* colour of column w0 + i (-1: none) */
    FD_LDS_PTR(int) s_r = s_c + xlen;                                  /* the row at place p of the tile's order */
    const CT *color = (const CT *)st.color;
    const int nx = (int)(w1 - w0l), npair = nx >> 1;
    const int tid = (int)threadIdx.x;
    /* The prologue is STRAIGHT-LINE code: every load unconditional, from an index clamped into range (a batch the tile does not need
       re-reads its last element), all in flight together and waited for once.  A load inside a branch -- even a wave-uniform one --
       leaves the compiler unsure on some path whether it has been waited for: it then waits for EVERYTHING outstanding at the first use
       of the value, and in the loop below that includes the store of the pass before (one counter for loads and stores on this target):
       a store round trip per pass made the first form of this kernel 105 us instead of 60. */
    constexpr int XP = 4, XC = 8;
    fd_pair_t vx[XP];
    CT vc[XC];
    int ej[NE], eq[NE], ei[NE];
    const long long pos = R0 + tid;
    const int myrow = st.row_pack[2 * (pos < R1 ? pos : R1 - 1)];
#pragma unroll
    for (int u = 0; u < XP; ++u) { const int i = u * 256 + tid; vx[u] = *reinterpret_cast<const fd_pair_t *>(x + w0l + 2 * (i < npair ? i : npair - 1)); }
#pragma unroll
    for (int u = 0; u < XC; ++u) { const int i = u * 256 + tid; vc[u] = color[w0l + (i < nx ? i : nx - 1)]; }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int i = u * 256 + tid, ic = A0 + (i < nst ? i : nst - 1);
        ej[u] = st.ent_col[ic]; eq[u] = st.ent_slot[ic]; ei[u] = st.ent_info[ic];
    }
    const T hmine = eps[c_lo + (tid < nchunk ? tid : 0)];
#pragma unroll
    for (int u = 0; u < XP; ++u) { const int i = u * 256 + tid; if (i < npair) { s_x[2 * i] = vx[u].x; s_x[2 * i + 1] = vx[u].y; } }
    if ((nx & 1) && tid == 0) s_x[nx - 1] = x[w0l + nx - 1];
#pragma unroll
    for (int u = 0; u < XC; ++u) { const int i = u * 256 + tid; if (i < nx) s_c[i] = vc[u] == (CT)(-1) ? -1 : (int)vc[u]; }
    s_r[tid] = myrow;
    if (tid < nchunk) { s_h[tid] = hmine; s_y[tid] = (T)1 / (MODE == 1 ? 2 * hmine : hmine); }
    for (int i = 256 + tid; i < nchunk; i += 256) { const T h = eps[c_lo + i]; s_h[i] = h; s_y[i] = (T)1 / (MODE == 1 ? 2 * h : h); }      /* (more than 256 colours in a batch) */
    __syncthreads();
    const bool given = MODE == 0 && st.fx_base != nullptr;
    /* 1: plain terms (and, where f(x) is handed over, the row's value: requested here, unconditionally, before any store of this thread) */
    T fxg[NE];
    int rr[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) rr[u] = s_r[ei[u] & 255];
    if (given) {
#pragma unroll
        for (int u = 0; u < NE; ++u) fxg[u] = ((const T *)st.fx_base)[rr[u]];
    } else {
#pragma unroll
        for (int u = 0; u < NE; ++u) fxg[u] = 0;
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        if (u * 256 < nst) {
            const int i = u * 256 + tid;
            const T t = f.term((long long)rr[u], (long long)ej[u], (T)s_x[(unsigned)(ej[u] - w0)]);
            if (i < nst) s_t[i] = t;
        }
    }
    __syncthreads();
    /* 2: every entry's quotient */
    T *out = (T *)st.out;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        if (u * 256 >= nst) break;
        const int i = u * 256 + tid;
        if (i >= nst) continue;
        const unsigned off = (unsigned)(ej[u] - w0);
        const int c = s_c[off];
        if (c < 0) { if (c_lo == 0) out[eq[u]] = (T)0; continue; }
        if (c < c_lo || c >= c_hi) continue;
        const int k = (ei[u] >> 8) & 255, L = (ei[u] >> 16) & 255, b = i - k;
        const long long r = rr[u];
        const T h = s_h[c - c_lo], y = s_y[c - c_lo], v = s_x[off];
        const T tp = f.term(r, (long long)ej[u], v + h), tm = MODE == 1 ? f.term(r, (long long)ej[u], v - h) : (T)0;
        const T t0 = s_t[b];
        T sp = k == 0 ? tp : t0, sm = k == 0 ? tm : t0, sa = t0;
        for (int uu = 1; uu < L; ++uu) {
            const T tv = s_t[b + uu];
            const bool me = uu == k;
            sp = sp + (me ? tp : tv);
            if (MODE == 1) sm = sm + (me ? tm : tv);
            if (MODE == 0 && !given) sa = sa + tv;
        }
        out[eq[u]] = fd_div_shared<T>(sp - (MODE == 1 ? sm : given ? fxg[u] : sa), MODE == 1 ? 2 * h : h, y);
    }
}

/* ---- a ROW FUNCTOR storing an exact band itself (round 5) ----------------------------------------------------------------------------
 * F: `template <class P> __device__ T operator()(long long i, const P &X) const` -- row i of the residual at the point X (X(j) = j-th
 * coordinate) -- as for fd_csc_store_cols.  For a Jacobian whose pattern is the exact band (L, U) with cyclic colours (what a plan
 * hands a FD_LAZY_CAP_STORE launcher as `fd_band_store`: CSC nzval of the band, BandedMatrix data or a Tridiagonal's three
 * diagonals) no index is read at all: lane t of a wavefront owns the columns jw + 2t, jw + 2t + 1, evaluates the L + U + 1 rows each
 * touches at x + eps e_j (and at x - eps e_j, or at x: the launch forms f(x) of its rows itself), divides, and the wavefront's
 * 128 (L + U + 1) quotients leave through fd_band_emit_wave as dense 16-byte stores.  The window of x the workgroup's 512 columns can
 * see is staged in LDS (a functor that reads further takes those coordinates from memory).  Same points, same operations as
 * fd_csc_store_cols: same bits.  Launch: (col_end - jstart + 511) / 512 workgroups of 256 threads, jstart = col_begin rounded down
 * to even; all colours in one launch.                                                                                               */
template <typename T, int MODE, class F, int L, int U>
__global__ void __launch_bounds__(256) fd_band_store_cols(F f, const T *__restrict__ x, const T *__restrict__ eps, fd_band_store st, long long jstart)
{
    constexpr int W = L + U + 1, HALO = L + U;
    __shared__ __attribute__((aligned(16))) T s_emit[4][FD_BAND_WAVE_LDS(W)];
    __shared__ __attribute__((aligned(16))) T s_x[512 + 2 * HALO + 4];
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const long long j0 = jstart + (long long)blockIdx.x * 512;
    if (j0 >= st.col_end) return;
    long long w0 = j0 - HALO > 0 ? j0 - HALO : 0, w1 = j0 + 512 + HALO < st.N ? j0 + 512 + HALO : st.N;
    w0 &= ~1ll;
    {
        typedef T fd_pair_t __attribute__((ext_vector_type(2)));
        const long long nx = w1 - w0, npair = nx / 2;
        fd_pair_t vx[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { const long long i = u * 256 + threadIdx.x; if (i < npair) vx[u] = *reinterpret_cast<const fd_pair_t *>(x + w0 + 2 * i); }
#pragma unroll
        for (int u = 0; u < 2; ++u) { const long long i = u * 256 + threadIdx.x; if (i < npair) { s_x[2 * i] = vx[u].x; s_x[2 * i + 1] = vx[u].y; } }
        if ((nx & 1) && threadIdx.x == 0) s_x[nx - 1] = x[w0 + nx - 1];
    }
    __syncthreads();
    const long long jw = j0 + (long long)wave * 128;
    if (jw >= st.col_end) return;                                          /* (wave-uniform: fd_band_emit_wave wants whole wavefronts) */
    const long long j = jw + 2 * lane;
    T q[2 * W];
    /* forward differences: f(x) of the W + 1 rows the lane's two columns touch, once (row j - U + m <-> base[m]) */
    T base[W + 1];
    if (MODE == 0) {
        const bool any = j < st.col_end && j + 1 >= st.col_begin && j < st.N;
        fd_window_column_point<T> X0 = {x, (const FD_LDS_PTR(T))s_x, w0, w1, j, (T)0, 2, 0xFFFFFFFFu};
#pragma unroll
        for (int m = 0; m <= W; ++m) {
            const long long r = j - U + m;
            base[m] = (any && r >= 0 && r < st.M) ? f(r, X0) : (T)0;
        }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const long long jj = j + o;
        const bool valid = jj >= st.col_begin && jj < st.col_end && jj < st.N;
        const T h = valid ? eps[fd_band_color(&st, jj)] : (T)1;
        const T dv = MODE == 1 ? 2 * h : h, yd = (T)1 / dv;
            fd_window_column_point<T> X = {x, (const FD_LDS_PTR(T))s_x, w0, w1, jj, h, 0, (jj >= w0 && jj < w1) ? (unsigned)(jj - w0) : 0xFFFFFFFFu};
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const long long r = jj - U + k;
            T v = 0;
            if (valid && r >= 0 && r < st.M) {
                X.minus = 0;
                const T vp = f(r, X);
                T vm;
                if (MODE == 1) { X.minus = 1; vm = f(r, X); }
                else vm = base[o + k];
                v = fd_div_shared<T>(vp - vm, dv, yd);
            }
            q[o * W + k] = v;
        }
    }
    fd_band_emit_wave<T, W>(&st, s_emit[wave], jw, q);
}

/* ---- the complex step for row functors (src/jacobians.jl:623-648) ---------------------------------------------------------------------
 * J[:, j] = imag(f(x + i eps e_j)) / eps with eps = eps(T), no subtraction.  A functor serves it if its call operator is generic in the
 * VALUE TYPE of the point -- `template <class P> __device__ typename P::value_type operator()(long long r, const P &X) const`, its
 * arithmetic written on `typename P::value_type` instead of the element type: X(j) then yields fd_cplx<T> and the row is evaluated in
 * complex arithmetic (the operators below: the products and sums of Julia's Complex; sin / cos / exp by their real formulas).
 * fd_csc_store_cols_cplx is fd_csc_store_cols for that step: every stored entry's row at x + i eps e_j (any colouring: at the colour's
 * point), imag / eps stored.  Launch as fd_csc_store_cols.  A functor that names the element type explicitly still compiles for
 * forward / central differences; the complex instantiation is compiled separately (fd_f_compile_rows: on first use). */
template <typename T> struct fd_cplx { T re, im; };
template <typename T> __device__ inline fd_cplx<T> operator+(fd_cplx<T> a, fd_cplx<T> b) { return {a.re + b.re, a.im + b.im}; }
template <typename T> __device__ inline fd_cplx<T> operator-(fd_cplx<T> a, fd_cplx<T> b) { return {a.re - b.re, a.im - b.im}; }
template <typename T> __device__ inline fd_cplx<T> operator-(fd_cplx<T> a) { return {-a.re, -a.im}; }
template <typename T> __device__ inline fd_cplx<T> operator*(fd_cplx<T> a, fd_cplx<T> b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
template <typename T> __device__ inline fd_cplx<T> operator/(fd_cplx<T> a, fd_cplx<T> b)
{
    const T d = b.re * b.re + b.im * b.im;
    return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
template <typename T> __device__ inline fd_cplx<T> operator+(fd_cplx<T> a, T s) { return {a.re + s, a.im}; }
template <typename T> __device__ inline fd_cplx<T> operator+(T s, fd_cplx<T> a) { return {s + a.re, a.im}; }
template <typename T> __device__ inline fd_cplx<T> operator-(fd_cplx<T> a, T s) { return {a.re - s, a.im}; }
template <typename T> __device__ inline fd_cplx<T> operator-(T s, fd_cplx<T> a) { return {s - a.re, -a.im}; }
template <typename T> __device__ inline fd_cplx<T> operator*(T s, fd_cplx<T> a) { return {s * a.re, s * a.im}; }
template <typename T> __device__ inline fd_cplx<T> operator*(fd_cplx<T> a, T s) { return {a.re * s, a.im * s}; }
template <typename T> __device__ inline fd_cplx<T> operator/(fd_cplx<T> a, T s) { return {a.re / s, a.im / s}; }
template <typename T> __device__ inline fd_cplx<T> sin(fd_cplx<T> a) { return {sin(a.re) * cosh(a.im), cos(a.re) * sinh(a.im)}; }
template <typename T> __device__ inline fd_cplx<T> cos(fd_cplx<T> a) { return {cos(a.re) * cosh(a.im), -(sin(a.re) * sinh(a.im))}; }
template <typename T> __device__ inline fd_cplx<T> exp(fd_cplx<T> a) { const T m = exp(a.re); return {m * cos(a.im), m * sin(a.im)}; }
/* the colour's complex point x + i e m_c; with a valid colouring only coordinate j carries the step */
template <typename T, typename CT> struct fd_cplx_colour_point {
    typedef fd_cplx<T> value_type;
    const T *x;
    const CT *color;
    int c;
    T e;
    __device__ fd_cplx<T> operator()(long long j) const { return fd_cplx<T>{x[j], ((int)color[j] == c) ? e : (T)0}; }
};
template <typename T> struct fd_cplx_column_point {
    typedef fd_cplx<T> value_type;
    const T *x;
    long long j;
    T e;
    __device__ fd_cplx<T> operator()(long long i) const { return fd_cplx<T>{x[i], i == j ? e : (T)0}; }
};
/* a materialised complex point: (re, im) pairs */
template <typename T> struct fd_cplx_plain_point {
    typedef fd_cplx<T> value_type;
    const T *x;
    __device__ fd_cplx<T> operator()(long long j) const { return fd_cplx<T>{x[2 * j], x[2 * j + 1]}; }
};
template <typename T, typename CT, class F>
__global__ void __launch_bounds__(256) fd_csc_store_cols_cplx(F f, const T *__restrict__ x, const T *__restrict__ eps, int c_lo, int c_hi, fd_csc_store st)
{
    __shared__ __attribute__((aligned(16))) T s_win[256 / 64][FD_CSC_WAVE_CAP];
    const long long nblk = (st.col_end - st.col_begin + 255) / 256, blk = fd_xcd_block(blockIdx.x, nblk);
    if (blk >= nblk) return;
    const long long j = st.col_begin + blk * 256 + threadIdx.x;
    const bool in = j < st.col_end;
    const int a = in ? st.colptr[j - st.col_begin] : st.colptr[st.col_end - st.col_begin];
    const int b = in ? st.colptr[j - st.col_begin + 1] : a;
    const CT *color = (const CT *)st.color;
    const int c = in ? (int)color[j] : 0;
    const bool none = in && c == (int)(CT)(-1);
    const bool mine = in && !none && c >= c_lo && c < c_hi;
    fd_csc_wave_run<T> run;
    run.begin((T *)st.out, s_win[threadIdx.x >> 6], a, b, !in || mine || (none && c_lo == 0));
    if (none && c_lo == 0)
        for (int q = a; q < b; ++q) run.put(q, (T)0);
    if (mine) {
        const T h = eps[c];
        constexpr int U = 4;
        for (int q0 = a; q0 < b; q0 += U) {
            long long r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = st.rowval[q0 + u < b ? q0 + u : b - 1];
            T v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (q0 + u >= b) { v[u] = 0; continue; }
                fd_cplx<T> w;
                if (st.valid_coloring) { const fd_cplx_column_point<T> X = {x, j, h}; w = f(r[u], X); }
                else { const fd_cplx_colour_point<T, CT> X = {x, color, c, h}; w = f(r[u], X); }
                v[u] = w.im / h;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (q0 + u < b) run.put(q0 + u, v[u]);
        }
    }
    run.template flush<true>();
}
/* ---- a ROW FUNCTOR storing COLUMN-RANGE storage itself (BlockBandedMatrix data; round 6) ------------------------------------------------
 * fd_colrange_store: column j holds the contiguous rows [row_first, row_first + row_count) at out + dest (ext/FiniteDiffBlockBandedMatricesExt.jl:
 * 44-68 assigns exactly those).  A workgroup takes EIGHT columns and deals their rows out to its 256 threads as one flat list (a column of
 * 96 rows alone would leave a quarter of the lanes idle): consecutive threads take consecutive rows of a column -- the values leave as
 * dense runs, and rows of a block read mostly the same coordinates (broadcast loads).  F as for fd_csc_store_cols; the plan offers this
 * descriptor only with a colouring it has verified, so the point is x +- eps e_j (MODE 0 forward: the row at x is evaluated beside it;
 * 1 central; 2 the complex step: imag(f(x + i eps e_j)) / eps, src/jacobians.jl:633-635).  Same points, same operations as the
 * hand-over path: same bits.  `base`: f(x) of all M rows for MODE 0 (a dense block column shares its rows with bs - 1 others: evaluating
 * f(x) once halves the row evaluations), or NULL.  Launch fd_xcd_grid((col_end - col_begin + 7) / 8) workgroups of 256 threads. */
template <typename T, typename CT, int MODE, class F>
__global__ void __launch_bounds__(256) fd_colrange_store_cols(F f, const T *__restrict__ x, const T *__restrict__ eps, int c_lo, int c_hi, fd_colrange_store st,
                                                              const T *__restrict__ base)
{
    constexpr int G = 8;                         /* columns per workgroup */
    __shared__ int s_pre[G + 1], s_rlo[G];
    __shared__ long long s_dst[G];
    __shared__ T s_e[G];
    const long long ncol = st.col_end - st.col_begin, nw = (ncol + G - 1) / G, wg = fd_xcd_block(blockIdx.x, nw);
    if (wg >= nw) return;
    if (threadIdx.x < G) {
        const long long jl = wg * G + threadIdx.x;
        int cnt = 0;
        if (jl < ncol) {
            const int c = (int)((const CT *)st.color)[st.col_begin + jl];
            if (c != (int)(CT)(-1) && c >= c_lo && c < c_hi) {      /* (a column of another colour chunk, or without a colour: none of its rows) */
                cnt = st.row_count[jl];
                s_rlo[threadIdx.x] = st.row_first[jl];
                s_dst[threadIdx.x] = st.dest[jl];
                s_e[threadIdx.x] = eps[c];
            }
        }
        s_pre[threadIdx.x + 1] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s_pre[0] = 0;
        for (int g = 0; g < G; ++g) s_pre[g + 1] += s_pre[g];
    }
    __syncthreads();
    const int total = s_pre[G];
    for (int idx = (int)threadIdx.x; idx < total; idx += 256) {
        int g = 0;
#pragma unroll
        for (int u = 1; u < G; ++u) g += idx >= s_pre[u] ? 1 : 0;
        const int t = idx - s_pre[g];
        const long long j = st.col_begin + wg * G + g, r = (long long)s_rlo[g] + t;
        const T e = s_e[g];
        T *out = (T *)st.out + s_dst[g];
        if constexpr (MODE == 2) {
            const fd_cplx_column_point<T> X = {x, j, e};
            const fd_cplx<T> w = f(r, X);
            out[t] = w.im / e;
        } else {
            fd_column_point<T> X = {x, j, e, 0};
            const T vp = f(r, X);
            T vm;
            if (MODE == 0 && base) vm = base[r];                           /* f(x), evaluated once for all columns */
            else { X.minus = MODE == 1 ? 1 : 2; vm = f(r, X); }
            out[t] = (vp - vm) / (MODE == 1 ? 2 * e : e);
        }
    }
}
/* ---- a ROW FUNCTOR storing BandedBlockBandedMatrix data itself (uniform blocks; round 6) -------------------------------------------------
 * fd_bbb_store: every slot of every in-band slab of every column (ext/FiniteDiffBlockBandedMatricesExt.jl:16-42 assigns the rows inside
 * their block's sub-band and leaves the others 0).  One thread per (column, slot), consecutive threads on consecutive slots of a
 * column: slot (d, t) of column j = block J's local column jj is row k = jj + t - mu of block K = J + d - bu; a row inside its block gets
 * (f(x + eps e_j)[r] - f(x)[r]) / eps (central: the two-sided quotient) -- exactly +-0 when the row does not depend on the column, as
 * the hand-over path has it -- every other slot 0.  The plan offers the descriptor only with a colouring it has verified for the
 * BBB pattern, all colours in one batch.  F as for fd_csc_store_cols; MODE 0 forward (base: f(x) of all rows, or NULL = evaluated
 * here), 1 central.  Launch (N * (bl + bu + 1) * (lam + mu + 1) + 255) / 256 workgroups of 256 threads. */
template <typename T, typename CT, int MODE, class F>
__global__ void __launch_bounds__(256) fd_bbb_store_cols(F f, const T *__restrict__ x, const T *__restrict__ eps, fd_bbb_store st, const T *__restrict__ base)
{
    const int w = st.bl + st.bu + 1, sw = st.lam + st.mu + 1, R = w * sw;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= st.N * R) return;
    const long long j = e / R, bs = st.block_size;
    const int s = (int)(e - j * R), d = s / sw, t = s - d * sw;
    const long long J = j / bs, K = J + d - st.bu, jj = j - J * bs, k = jj + t - st.mu;
    const long long st0 = st.start[d + (long long)w * J];
    if (st0 < 0) return;                                             /* no such block and no slab reserved for it */
    T *o = (T *)st.out + st0 + jj * st.stride[J] + t;
    const int c = (int)((const CT *)st.color)[j];
    if (!(K >= 0 && K < st.nblk && k >= 0 && k < bs) || c == (int)(CT)(-1)) { *o = (T)0; return; }
    const long long r = K * bs + k;
    const T h = eps[c];
    fd_column_point<T> X = {x, j, h, 0};
    const T vp = f(r, X);
    T vm;
    if (MODE == 0 && base) vm = base[r];
    else { X.minus = MODE == 1 ? 1 : 2; vm = f(r, X); }
    *o = (vp - vm) / (MODE == 1 ? 2 * h : h);
}
#endif /* __HIPCC__ && __cplusplus */

#endif /* FDJAC_DEVICE_H */
