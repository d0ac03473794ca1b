/*
 * fdjac.h -- C ABI of libfdjac: the MI355X (gfx950) implementation of FiniteDiff.jl's
 * coloured sparse-Jacobian hot path.
 *
 * One entry point, fd_jacobian(), replaces the body of the cached in-place
 *     FiniteDiff.finite_difference_jacobian!(J, f, x, cache::JacobianCache, f_in; relstep,
 *                                            absstep, colorvec, sparsity, dir)
 * (reference: src/jacobians.jl:504-653) for a cache whose `sparsity`/`colorvec` have been
 * compiled into a device-resident *plan*.  Plans play the role of the reference's
 * per-matrix-type `_colorediteration!` overloads (the package-extension plugin surface):
 *
 *   fd_plan_create_csc          ext/FiniteDiffSparseArraysExt.jl:38-47,51-52 (common-pattern fast path)
 *   fd_plan_create_csc_dense    ext/FiniteDiffSparseArraysExt.jl:20-28       (sparse pattern, dense J)
 *   fd_plan_create_coo_dense    src/iteration_utils.jl:25-32 + src/jacobians.jl:473-488,524-528
 *   fd_plan_create_entries      any J storage enumerated by the caller (incl. ext/FiniteDiffBlockBandedMatricesExt.jl:16-42)
 *   fd_plan_create_tridiagonal  src/iteration_utils.jl:25-32 on a LinearAlgebra.Tridiagonal J
 *   fd_plan_create_dense        sparsity === nothing: src/jacobians.jl:548-557,590-598,626-631
 *   fd_plan_create_banded       ext/FiniteDiffBandedMatricesExt.jl:13-27
 *   fd_plan_create_blockbanded  ext/FiniteDiffBlockBandedMatricesExt.jl:44-68
 *   fd_plan_create_bandedblockbanded  ext/FiniteDiffBlockBandedMatricesExt.jl:16-42
 *
 * Step sizes follow src/epsilons.jl:26-29,50-53,104-107 with the masked-norm rule of
 * src/jacobians.jl:559-561 / 600-602 / 624.  Arithmetic is Float64 (fd_*) or Float32 (fd32_*, end of this file).
 *
 * Conventions
 *   - plain C: pointers and sizes only, no exceptions, int status returns (0 = FD_OK);
 *     fd_last_error() returns the calling thread's last message.
 *   - pattern / colour arrays are HOST pointers read once at plan creation; idx_bytes is 4 or
 *     8, idx_base 0 or 1 (Julia passes its own Int64 1-based arrays untouched).
 *   - colorvec holds colours 1..C (the reference's convention); columns whose colour is < 1
 *     are never perturbed and their stored entries are written as 0 (fill_matrix!).
 *   - x, f_in and the outputs are HOST or DEVICE pointers as stated by the FD_HOST/FD_DEVICE
 *     argument next to them.  Device pointers should be 16-byte aligned (others are staged).
 *   - x is never written (stronger than the reference, whose central arm perturbs and
 *     restores the caller's x: src/jacobians.jl:604,620).
 *   - one in-flight fd_jacobian per plan; different plans may be used from different threads.
 */
#ifndef FDJAC_H
#define FDJAC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDJAC_VERSION 500

typedef struct fd_ctx fd_ctx;
typedef struct fd_plan fd_plan;

enum fd_status {
    FD_OK = 0,
    FD_ERR_ARG = 1,         /* null / out-of-range argument */
    FD_ERR_SHAPE = 2,       /* DimensionMismatch (src/jacobians.jl:516) or inconsistent pattern */
    FD_ERR_UNSUPPORTED = 3, /* fdtype_error (src/epsilons.jl:159-167) / not built */
    FD_ERR_HIP = 4,         /* HIP runtime failure (message has the hipError string) */
    FD_ERR_CALLBACK = 5,    /* the f! launcher returned non-zero */
    FD_ERR_NOMEM = 6,
    FD_ERR_NODEVICE = 7,    /* no usable gfx950 device */
    FD_ERR_COMM = 8,        /* RCCL failure, or RCCL not available (fd_comm_*) */
    FD_ERR_STALE = 9        /* a deferred content check (fd_plan_matches_async) found that the plan is no longer the plan of the caller's */
                            /* arrays: results enqueued since that check are those of the OLD pattern / colours -- recompile, recompute  */
};

enum fd_fdtype { FD_FORWARD = 0, FD_CENTRAL = 1, FD_COMPLEX = 2 }; /* Val(:forward|:central|:complex) */
enum fd_memkind { FD_HOST = 0, FD_DEVICE = 1 };

/*
 * The user's f!(fx, x) as a *launcher*: enqueue on `stream` (a hipStream_t) the evaluation of
 * nbatch independent points  fx[b*fx_stride + r] = f(x[b*x_stride + :])[r],  b = 0..nbatch-1.
 * Strides are in elements; elements are double, or (re,im) double pairs when is_complex != 0.
 * Only rows row_begin <= r < row_end are consumed by the library (the whole range unless the
 * plan has a column window); writing the other rows is allowed.  Must not synchronise; return
 * 0 on success.  Replaces the calls at src/jacobians.jl:541,563,605-606,634.
 */
typedef int (*fd_f_launch)(void *fctx, void *fx, const void *x, int64_t nbatch, int64_t x_stride,
                           int64_t fx_stride, int64_t row_begin, int64_t row_end, int is_complex,
                           void *stream);

/*
 * Optional fast path: f! at LAZILY perturbed points.  A coloured finite difference only ever
 * evaluates f! at x + eps_c * [color == c]; instead of materialising those points
 * (src/jacobians.jl:562 does it in place) the launcher receives the base x plus the perturbation
 * rule and applies it while loading.  Point b (0 <= b < ncolors*pts) of the batch is
 *     forward :  x[j] + eps[c_lo+b] * (color[j] == c_lo+b)
 *     central :  x[j] +/- eps[c_lo + b%ncolors] * (color[j] == c_lo + b%ncolors),  minus for b >= ncolors
 *     complex :  (x[j],  eps[c_lo+b] * (color[j] == c_lo+b))           [outputs are (re,im) pairs]
 * written to fx[b*fx_stride + r].  If base_out != NULL the unperturbed f!(x) is also written to
 * base_out (forward differences without f_in; saves one launch).  Results are bit-identical to the
 * materialised path.  In Julia the same thing is a lazy AbstractVector wrapper handed to f!.
 */
typedef struct fd_lazy_points {
    const void *x;        /* base point, double[N]                                           */
    const void *color;    /* per-column colour, 0-based; none = 0xFF (1-byte) / -1 (4-byte)  */
    const void *eps;      /* step size per colour (device, element type of x), indexed by 0-based colour */
    void *base_out;       /* NULL or double[M] for f!(x)                                      */
    int32_t color_bytes;  /* 1 or 4                                                           */
    int32_t c_lo;         /* first colour of the batch                                        */
    int32_t ncolors;      /* colours in the batch                                             */
    int32_t pts;          /* 1 forward/complex, 2 central                                     */
    int32_t is_complex;
    int32_t imag_only;    /* complex step only, set only for launchers registered with FD_LAZY_CAP_IMAG_ONLY: write  */
                          /* fx[b*fx_stride + r] = imag(f(point b))[r] as a REAL array (fx_stride in doubles) -- the */
                          /* real parts of a complex-step evaluation are never used (src/jacobians.jl:635)           */
    int32_t part;         /* reserved: always 0 / 1 (round 2 evaluated f! in row strips through these; measured slower,      */
    int32_t nparts;       /* removed in round 3)                                                                             */
    int32_t diff;         /* forward / central only, set only for launchers registered with FD_LAZY_CAP_DIFF: write  */
                          /* the DIFFERENCES the reference forms next (src/jacobians.jl:565,607) instead of the      */
                          /* values -- forward: fx[b*fx_stride + r] = f(point b)[r] - f(x)[r]; central: ncolors      */
                          /* arrays, fx[b*fx_stride + r] = f(plus point b)[r] - f(minus point b)[r] -- each an IEEE  */
                          /* subtraction of the two values the plain path would have stored (same bits, half the     */
                          /* f! output traffic for central differences, no f(x) pass for forward ones).  base_out is */
                          /* NULL then.  2 = as 1, and f(x) counts as evaluated by this call (bookkeeping only)      */
    int32_t store_kind;   /* what `store` points to (include/fdjac_device.h: FD_STORE_BAND a fd_band_store, FD_STORE_STENCIL5 a    */
                          /* fd_stencil5_store, FD_STORE_CSC a fd_csc_store -- any pattern, column by column;                     */
                          /* fd_stencil5_store, FD_STORE_COLRANGE a fd_colrange_store -- BlockBandedMatrix, complex step:    */
                          /* store imag(f(point of the column's colour)[r]) / eps, is_complex = imag_only = 1); 0 when store  */
                          /* is NULL                                                                                         */
    const void *store;    /* launchers registered with FD_LAZY_CAP_STORE only: non-NULL = a `fd_band_store`                  */
                          /* (include/fdjac_device.h, host memory, valid during the call): store the finished quotients into */
                          /* the Jacobian yourself -- fd_band_emit(store, r, c, (f(point of colour c)[r] - f(x)[r]) / eps[c]) */
                          /* (central: (f(+) - f(-)) / (2 eps[c])) for every row r in [row_begin, row_end) and every colour  */
                          /* of the batch, CSC nzval / BandedMatrix data / Tridiagonal diagonals alike; fx / base_out are    */
                          /* not written and no decompression follows.  The plan hands it out only for a verified exact band */
                          /* with cyclic colours (the default then; FDJAC_LAZY_STORE=0: never); the launcher may still       */
                          /* return FD_LAZY_DECLINED                                                                         */
    const void *eps_job;  /* launchers registered with FD_LAZY_CAP_FUSED_EPS only (the library's built-in families): non-NULL = */
                          /* the step sizes have NOT been computed -- this launch runs the step-size reduction too (its first   */
                          /* workgroups) and the storing wavefronts wait for the published step sizes: the whole Jacobian in    */
                          /* ONE launch (N <= 2^21 on one GPU; every sharded call).  `eps` is written by that launch.  A        */
                          /* launcher that returns FD_LAZY_DECLINED is called again after the library's own reduction.         */
} fd_lazy_points;
typedef int (*fd_f_launch_lazy)(void *fctx, void *fx, const fd_lazy_points *points, int64_t fx_stride,
                                int64_t row_begin, int64_t row_end, void *stream);
/* A lazy launcher may return FD_LAZY_DECLINED without having enqueued anything (e.g. a batch too large for its
   on-chip staging); the library then materialises that batch's points and calls the plain fd_f_launch. */
#define FD_LAZY_DECLINED 100

typedef struct fd_plan_opts {
    int32_t fdtype;        /* enum fd_fdtype */
    int32_t flags;         /* FD_PLAN_* bits, 0 = defaults */
    int64_t col_begin;     /* column window [col_begin, col_end), 0-based; 0,0 = all columns.   */
    int64_t col_end;       /*   outputs then hold only the window's stored entries (multi-GPU)  */
    int64_t x_begin;       /* entries of x the windowed f! reads, [x_begin, x_end); 0,0 = all   */
    int64_t x_end;
    int64_t scratch_bytes; /* cap for the batched perturbed-point scratch; 0 = default (64 GiB) */
    int64_t color_begin;   /* colour ownership [color_begin, color_end), 0-based colours; 0,0 = all colours.    */
    int64_t color_end;     /*   Only the owned colours are perturbed / evaluated and only stored values whose   */
                           /*   column has an owned colour are written (values of columns without colour: by    */
                           /*   the owner of colour 0); everything else in outs is left untouched.  This is the */
                           /*   "each GPU owns a disjoint subset of colours" split: it needs no cooperation     */
                           /*   from f! (an opaque f! is evaluated on full vectors), caps at C ranks, and the   */
                           /*   owners' outputs add up to the full result when outs start from zero.            */
} fd_plan_opts;
/* fd_plan_opts.flags */
#define FD_PLAN_EPS_CONTIGUOUS 1   /* accepted and ignored since round 5: the step-size reduction always sums contiguous groups of x    */
                                   /* (see "Sharded step-size reduction" below)                                                     */

#define FD_PLAN_COMPLEX_X 2         /* x, the f! values and J are COMPLEX (returntype <: Complex with Val(:forward) / Val(:central):  */
                                   /* src/jacobians.jl:94-128, 537-622; test/finitedifftests.jl:480-513).  M, N, the pattern and     */
                                   /* colorvec are given in complex elements, as the reference has them; x / f_in / outs are        */
                                   /* (re, im) pairs (Complex{T} arrays as they lie in memory); the launcher is called with         */
                                   /* is_complex = 1.  The masked norm is over complex elements, the step REAL, the real parts are   */
                                   /* perturbed, the quotient is complex / real -- the reference's generic loop.  Plan kinds: CSC    */
                                   /* (common pattern), dense J with a CSC / index-list pattern, entry lists, the dense arm; no     */
                                   /* lazy launchers; fd_plan_info reports lengths in REAL numbers (2 x the complex counts).       */
                                   /* Val(:complex) with this flag is the reference's fdtype_error (FD_ERR_UNSUPPORTED).            */

#define FD_PLAN_STORE_CSC 8          /* SparseMatrixCSC common-pattern plans: keep a compact copy of the local pattern on the device      */
                                   /* (fd_csc_store, include/fdjac_device.h: 4 bytes per stored entry + 4 per column) so that a launcher  */
                                   /* registered with FD_LAZY_CAP_STORE_CSC can store the Jacobian of ANY pattern itself, column by       */
                                   /* column.  No requirement on the colouring.  Needs fewer than 2^31 local entries; plans of an exact   */
                                   /* band / 5-point stencil skip it (their closed-form descriptors are cheaper).  FD_INFO_STORE_CSC      */
                                   /* reports whether it was built.                                                                      */
#define FD_PLAN_STORE_CSC_ALWAYS 16   /* (with FD_PLAN_STORE_CSC) build that copy for exact bands / 5-point stencils too: for launchers that store     */
                                   /* through fd_csc_store ONLY -- runtime-compiled functors (fd_f_compile_rows) on a banded pattern             */
#define FD_PLAN_STORE_CSC_ROWS 32     /* (with FD_PLAN_STORE_CSC, plans that hold every column) also keep the pattern BY ROWS on the device --  */
                                   /* per row its entries' columns and their slots in nzval (fd_csc_store.row_ptr / row_col / row_slot, 8 bytes */
                                   /* per stored entry + 4 per row) -- for the row-wise store of SEPARABLE residuals (fd_csc_store_rows,       */
                                   /* include/fdjac_device.h; fd_f_compile_terms); fd_plan_row_lists hands the lists out                     */
#define FD_PLAN_FINGERPRINT 4        /* record 64-bit content fingerprints of the pattern / colour arrays the plan is compiled from, so  */
                                   /* that fd_plan_matches can later tell whether the caller's arrays still hold that content        */

/* ---- context ------------------------------------------------------------------------- */
/* stream: an existing hipStream_t to enqueue on (e.g. the caller's), NULL to create a private non-blocking stream,
   or FD_STREAM_DEFAULT for the device's legacy default (null) stream -- what a host framework whose "current stream"
   is the default stream needs so that its own work stays ordered with the library's. */
#define FD_STREAM_DEFAULT ((void *)(uintptr_t)1)
int fd_ctx_create(int device, void *stream, fd_ctx **out);
int fd_ctx_destroy(fd_ctx *ctx);
void *fd_ctx_stream(fd_ctx *ctx);
int fd_ctx_synchronize(fd_ctx *ctx);
const char *fd_last_error(void);
int fd_version(void);

/* ---- plans ------------------------------------------------------------------------------ */
/* SparseMatrixCSC J sharing colptr/rowval with `sparsity`: outs[0] = nzval (nnz of the window). */
int fd_plan_create_csc(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr, const void *rowval,
                       int idx_bytes, int idx_base, const void *colorvec, int color_bytes,
                       const fd_plan_opts *opts, fd_plan **out);
/* The same plan from a pattern that ALREADY LIVES ON THE DEVICE (e.g. the colPtr / rowVal of a device sparse matrix and a
   device colour vector, the caller's index width and base): the plan is compiled by kernels, nothing crosses PCIe
   (fdjac_planbuild.hip).  Patterns the device builder does not handle are copied to the host once and built there. */
int fd_plan_create_csc_device(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr_dev, const void *rowval_dev,
                              int idx_bytes, int idx_base, const void *colorvec_dev, int color_bytes,
                              const fd_plan_opts *opts, fd_plan **out);
/* FNV-1a checksum of the plan's compiled pattern (diagnostic: equal checksums <=> the kernels are driven identically;
   the device plan builder is tested against the host builder with it). */
int fd_plan_checksum(fd_plan *plan, uint64_t *checksum_out);
/* SparseMatrixCSC sparsity, dense column-major J (M x N): outs[0] = J. */
int fd_plan_create_csc_dense(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr,
                             const void *rowval, int idx_bytes, int idx_base, const void *colorvec,
                             int color_bytes, const fd_plan_opts *opts, fd_plan **out);
/* rows_index / cols_index lists (findstructralnz), dense column-major J: outs[0] = J. */
int fd_plan_create_coo_dense(fd_ctx *ctx, int64_t M, int64_t N, const void *rows_index,
                             const void *cols_index, int64_t nnz, int idx_bytes, int idx_base,
                             const void *colorvec, int color_bytes, const fd_plan_opts *opts,
                             fd_plan **out);
/* The general form: stored entry k lives at row rows_index[k], column cols_index[k] and is
   written to outs[0][dest[k]] (dest 0-based, < out_len).  outs[0] is zero-filled first.  Any
   matrix type whose storage the caller can enumerate goes through here -- e.g. a
   BandedBlockBandedMatrix (ext/FiniteDiffBlockBandedMatricesExt.jl:16-42) or a SparseMatrixCSC
   J whose pattern differs from `sparsity` (ext/FiniteDiffSparseArraysExt.jl:20-28). */
int fd_plan_create_entries(fd_ctx *ctx, int64_t M, int64_t N, const void *rows_index,
                           const void *cols_index, const int64_t *dest, int64_t nnz,
                           int64_t out_len, int idx_bytes, int idx_base, const void *colorvec,
                           int color_bytes, const fd_plan_opts *opts, fd_plan **out);
/* Dense uncoloured arm, `sparsity === nothing` (src/jacobians.jl:548-557, 590-598, 626-631): column i
   (1 <= i <= ncols = maximum(colorvec)) perturbs x[i] alone with the per-element step
   compute_epsilon(fdtype, x[i], relstep, absstep, dir); outs[0] = J[:, 1:ncols], column-major M x ncols.
   (BASELINE config 1's family.) */
int fd_plan_create_dense(fd_ctx *ctx, int64_t M, int64_t N, int64_t ncols, const fd_plan_opts *opts,
                         fd_plan **out);
/* LinearAlgebra.Tridiagonal J (N x N): outs = {dl (N-1), d (N), du (N-1)}. */
int fd_plan_create_tridiagonal(fd_ctx *ctx, int64_t N, const void *colorvec, int color_bytes,
                               const fd_plan_opts *opts, fd_plan **out);
/* BandedMatrix J (M x N, bandwidths l,u): outs[0] = data, (l+u+1) x N column-major,
   data[(u + i - j) + (l+u+1)*j] = J[i,j] (0-based), out-of-matrix slots written as 0. */
int fd_plan_create_banded(fd_ctx *ctx, int64_t M, int64_t N, int64_t l, int64_t u,
                          const void *colorvec, int color_bytes, const fd_plan_opts *opts,
                          fd_plan **out);
/* BlockBandedMatrix J: square block structure blk_sizes[nblk], block bandwidths (bl,bu),
   block_starts = (bl+bu+1) x nblk band storage of the idx_base-based start of block (K,J) in
   data: block_starts[(bu + K - J) + (bl+bu+1)*J] (0-based K,J); block_strides[J] = column
   stride of block-column J.  outs[0] = data. */
int fd_plan_create_blockbanded(fd_ctx *ctx, int64_t nblk, const void *blk_sizes, int64_t bl,
                               int64_t bu, const void *block_starts, const void *block_strides,
                               int idx_bytes, int idx_base, const void *colorvec, int color_bytes,
                               const fd_plan_opts *opts, fd_plan **out);
/* BandedBlockBandedMatrix J (round 4): square block structure blk_sizes[nblk], block bandwidths (bl, bu), sub-block bandwidths
   (lam, mu); block_starts[(bu + K - J) + (bl+bu+1)*J] = idx_base-based start of block (K, J)'s banded-data slab in data
   (pointer(bandeddata(view(J, K, J)))), block_strides[J] = its column stride: entry (k, j) of the block (0-based) lives at
   start + j*stride + mu + k - j (ext/FiniteDiffBlockBandedMatricesExt.jl:29-36).  outs[0] = data; every slot of every in-band
   slab is written: the quotient, or 0 for rows outside the block / columns without colour; data_len = length of data (slots no
   slab reaches are zero-filled, as the reference's fill!(J, 0) leaves them).  Whole column range only. */
int fd_plan_create_bandedblockbanded(fd_ctx *ctx, int64_t nblk, const void *blk_sizes, int64_t bl, int64_t bu, int64_t lam, int64_t mu,
                                     const void *block_starts, const void *block_strides, int64_t data_len, int idx_bytes, int idx_base,
                                     const void *colorvec, int color_bytes, const fd_plan_opts *opts, fd_plan **out);
int fd_plan_destroy(fd_plan *plan);

/* Is `plan` still the plan of THESE arrays?  The reference holds colorvec / sparsity by reference and re-reads them on every
   call (src/jacobians.jl:512-513; the O(nnz) pattern comparison of ext/FiniteDiffSparseArraysExt.jl:51-52): an in-place edit
   takes effect on the next call.  A plan is a snapshot.  A host shim finds its plan in O(1) by the identity of the arrays and
   offers an explicit invalidate; a caller that wants the reference's re-read semantics creates its plans with
   FD_PLAN_FINGERPRINT and asks here before a call: the CONTENT of the current arrays is compared with what the plan was
   compiled from -- by kernels when the arrays live on the device (nothing is copied or allocated; ~0.04 ms for an N = 10^7
   tridiagonal Int32 pattern), by host threads otherwise.  Blocking.  *matches_out = 1 / 0.
     idx_a / idx_b : colptr / rowval (CSC plans; the plan's column window of them is compared), rows_index / cols_index
                     (index-list plans), ignored by structural plans (Tridiagonal, BandedMatrix, BlockBandedMatrix, dense arm)
     NULL members are not compared; lengths are in elements; a length that differs from the plan's is a mismatch.
   FD_ERR_UNSUPPORTED if the plan was created without FD_PLAN_FINGERPRINT. */
typedef struct fd_pattern_arrays {
    const void *idx_a;
    int64_t len_a;
    const void *idx_b;
    int64_t len_b;
    const void *colorvec;
    int64_t len_color;
    int32_t idx_bytes;     /* 4 or 8 */
    int32_t idx_base;      /* 0 or 1 */
    int32_t color_bytes;   /* 4 or 8 */
    int32_t memkind;       /* FD_HOST or FD_DEVICE: where ALL the arrays live */
} fd_pattern_arrays;
int fd_plan_matches(fd_plan *plan, const fd_pattern_arrays *now, int *matches_out);
/* The same comparison WITHOUT stopping the stream (device arrays only): ONE fused kernel fingerprints the three arrays and compares
   with the plan's words on the device; a mismatch raises a sticky status in pinned host memory -- no copy back, no synchronisation,
   the call returns at once and the Jacobian calls behind it keep the stream full (fd_plan_matches costs 3 launches, a copy and a
   hipStreamSynchronize per call).  The verdict is DEFERRED: it is reported as FD_ERR_STALE by the next fd_jacobian* call on this plan
   and by fd_ctx_synchronize -- whichever the host reaches first after the kernel ran -- and by fd_plan_stale at any time.  Results
   enqueued between the check and the report came from the stale plan: on FD_ERR_STALE drop the plan, compile a new one, recompute.
   (A mismatch the HOST can see -- another length -- is reported immediately: FD_ERR_STALE from this call.)  */
int fd_plan_matches_async(fd_plan *plan, const fd_pattern_arrays *now);
int fd_plan_stale(fd_plan *plan, int *stale_out);          /* non-blocking; 1 once a completed check of this plan found a mismatch (sticky) */

/* Introspection: what[] selectors for fd_plan_info. */
enum fd_plan_info_key {
    FD_INFO_M = 0, FD_INFO_N = 1, FD_INFO_NCOLORS = 2, FD_INFO_NOUTS = 3,
    FD_INFO_OUT0_LEN = 4, FD_INFO_OUT1_LEN = 5, FD_INFO_OUT2_LEN = 6,
    FD_INFO_ROW_BEGIN = 7, FD_INFO_ROW_END = 8, FD_INFO_NCHUNKS = 9, FD_INFO_SCRATCH_BYTES = 10,
    FD_INFO_NNZ_LOCAL = 11, FD_INFO_FCALLS_LAST = 12, FD_INFO_ENTRY_BEGIN = 13,
    FD_INFO_SORTED_GATHER = 14,       /* 1 if the LDS-transposed (colour-sorted) decompression kernel is used */
    FD_INFO_LINES_DIRECT_X100 = 15,   /* plan-time estimate, x100: 128-B lines per wave gather, storage order */
    FD_INFO_LINES_SORTED_X100 = 16,   /*   ... colour-sorted order */
    FD_INFO_WINDOW = 17,              /* 1 if the row-window (dense loads -> LDS) decompression kernel is used */
    FD_INFO_WIN_OVERREAD_X100 = 18,   /*   x100: f! values loaded per stored entry by that kernel (100 = none wasted) */
    FD_INFO_WINDOW2D = 19,            /*   1 if its tiles are 2-D (column runs one stencil stride apart) */
    FD_INFO_WIN_PERIOD = 20,          /*   period (in stored entries) of the regular tiles' entry codes, 0 = none */
    FD_INFO_COLRANGE_WG = 21,         /* block-banded plans: 1 = one workgroup per 32 columns, 0 = one wave per column */
    FD_INFO_SMALL_FUSED = 22,         /* 1 if the plan uses the fused single-workgroup launches of small problems */
    /* 23, 26, 28, 30: keys of round-2 kernel variants that were measured slower and removed in round 3 (LDS-DMA staging, row
       strips, rolling row windows, the computed-index band kernel; DESIGN section 5 keeps their numbers) */
    FD_INFO_EPS_CYCLIC = 24,          /* C if colorvec is cyclic (the step-size reduction computes the colours), else 0 */
    FD_INFO_EPS_NT = 25,              /* 1 if the step-size reduction reads x with non-temporal loads */
    FD_INFO_BAND_DESC = 31,           /* number of row-window tiles whose descriptors the kernel computes instead of loading (uniform band) */
    FD_INFO_LAZY_STORE = 32,          /* 1 if a FD_LAZY_CAP_STORE launcher stores the Jacobian of this plan itself (exact band, cyclic colours; FDJAC_LAZY_STORE=0: never) */
    FD_INFO_LAZY_DIFF = 29,           /* 1 if the plan asks a FD_LAZY_CAP_DIFF launcher for differences (FDJAC_LAZY_DIFF=0: never) */
    FD_INFO_BUILT_ON_DEVICE = 27,     /* 1 if the pattern was compiled by the device plan builder */
    FD_INFO_STORE_CSC = 33            /* number of local entries of the compact pattern copy of FD_PLAN_STORE_CSC, 0 = none */
};
/* Kernel variants are chosen when the plan is created (the FDJAC_* environment switches of DESIGN.md section 5 are
   read there, not per process and not per launch -- except FDJAC_COLRANGE_VEC, which only re-vectorises the same work);
   FD_INFO_WINDOW also reports the row-window kernel of Tridiagonal plans. */
int fd_plan_info(const fd_plan *plan, int key, int64_t *value);
/* The plan's pattern BY ROWS on the device (FD_PLAN_STORE_CSC | FD_PLAN_STORE_CSC_ROWS; ext/FiniteDiffSparseArraysExt.jl:38-47 walks the
   same entries column by column): row_ptr (M + 1 int32 offsets), row_col (per entry its 0-based column, ascending within a row), row_slot
   (per entry its index in nzval), the number of entries and the plan's serial (fd_csc_store.plan_serial).  Owned by the plan.
   FD_ERR_UNSUPPORTED when the plan has no such lists (flag not given, a column window, a pattern the storing copy is skipped for). */
int fd_plan_row_lists(const fd_plan *plan, const void **row_ptr_dev, const void **row_col_dev, const void **row_slot_dev, int64_t *entries,
                      uint64_t *plan_serial);

/* ---- the hot path ------------------------------------------------------------------------ */
/*
 * Fill the Jacobian's stored values.  f_in may be NULL (forward: f!(fx,x) is then evaluated
 * once, src/jacobians.jl:540-545); it is ignored for central / complex.  relstep <= 0 selects
 * default_relstep (src/epsilons.jl:133-144); absstep < 0 selects absstep = relstep.  dir is
 * the reference's `dir` (true/+1 or -1; forward only).  Blocks until outs are complete.
 */
int fd_jacobian(fd_plan *plan, fd_f_launch f, void *fctx, const void *x, int x_kind,
                const void *f_in, int f_in_kind, double relstep, double absstep, double dir,
                void *const *outs, int out_kind);
/* Same, but only enqueues on the context's stream (device x / f_in / outs only). */
int fd_jacobian_async(fd_plan *plan, fd_f_launch f, void *fctx, const void *x, const void *f_in,
                      double relstep, double absstep, double dir, void *const *outs);

/* Install (or clear, with NULL) the lazy-point launcher used for the perturbed batches of this
   plan; it shares the fctx passed to fd_jacobian.  The plain launcher is still required (it is
   used whenever the lazy one cannot be: f_in given to a central plan never happens; chunking is fine). */
int fd_plan_set_lazy_f(fd_plan *plan, fd_f_launch_lazy lazy);
/* Optional capabilities of the installed lazy launcher (bit mask; cleared by fd_plan_set_lazy_f). */
#define FD_LAZY_CAP_IMAG_ONLY 1   /* honours fd_lazy_points.imag_only: halves the f! output traffic of the complex step */
#define FD_LAZY_CAP_ROW_WINDOW 2  /* writes ONLY rows [row_begin & ~1, row_end + 1) of fx / base_out (informational since round 3: */
                                  /* the row-strip schedule that needed it was measured slower and removed)                  */
#define FD_LAZY_CAP_DIFF 4        /* honours fd_lazy_points.diff: writes f(point) - f(x) / f(plus) - f(minus) itself         */
#define FD_LAZY_CAP_STORE 8       /* honours fd_lazy_points.store: f!'s launch stores the Jacobian of a verified exact band itself           */
#define FD_LAZY_CAP_STORE_CSC 16  /* honours fd_lazy_points.store with store_kind = FD_STORE_CSC (fd_csc_store): the launch stores the Jacobian  */
                                  /* of ANY pattern column by column through the plan's compact copy of the pattern (FD_PLAN_STORE_CSC)       */
#define FD_LAZY_CAP_STORE_CSC_BASE 32  /* ... and evaluates f(x) of the rows it needs itself: forward differences without a caller's f_in run NO */
                                  /* plain evaluation before the storing launch (fd_csc_store.fx_base is NULL then; with f_in it is f_in)        */
#define FD_LAZY_CAP_STORE_CSC_COMPLEX 64  /* ... and serves the COMPLEX step through the column store too (fd_lazy_points.is_complex = 1 with   */
                                  /* store_kind = FD_STORE_CSC): every stored entry's row at x + i eps e_j, imag / eps stored                   */
                                  /* (src/jacobians.jl:623-648 + ext/FiniteDiffSparseArraysExt.jl:38-47 in one launch); without the bit a       */
                                  /* complex-step plan hands the values over as before                                                          */
#define FD_LAZY_CAP_STORE_COLRANGE 256  /* (with FD_LAZY_CAP_STORE) serves store_kind = FD_STORE_COLRANGE (BlockBandedMatrix data) for FORWARD and  */
                                       /* CENTRAL differences too, forming f(x) of its rows itself -- runtime-compiled functors                  */
                                       /* (fd_colrange_store_cols); without it a column-range plan offers the store to the complex step only     */
#define FD_LAZY_CAP_FUSED_EPS 128 /* honours fd_lazy_points.eps_job (library-internal protocol of the built-in storing launchers)               */
int fd_plan_set_lazy_caps(fd_plan *plan, int caps);

/* The step sizes of the last call, eps[c] for colours 1..C (host array of C doubles). */
int fd_plan_get_epsilons(fd_plan *plan, double *eps_out);
/* Diagnostic: wall_clock64 marks (100 MHz ticks) of the plan's last FUSED launch -- 0 first reduction workgroup starts, 1 last block sum
   published, 2 finisher starts, 3 finisher has every block sum, 4 step sizes published, 5 first storing workgroup starts, 6 / 7 first / last
   storing workgroup has the step sizes, 8 last storing wavefront done, 9 last storing workgroup starts.  Only in a process that set
   FDJAC_TEST_SWITCHES=1 and FDJAC_FUSED_TRACE=1 before the plan's first fused call (FD_ERR_UNSUPPORTED otherwise). */
int fd_plan_fused_trace(fd_plan *plan, long long *marks16);

/* Per-stage GPU time of the calls made since timing was enabled (HIP events on the plan's
   stream).  stage: 0 eps-reduce, 1 perturb, 2 f!, 3 diff+decompress (a FD_LAZY_CAP_STORE launch that evaluates f!, divides
   and stores in one kernel is recorded HERE: it is the difference + decompression), 4 whole call.
   ms_sum / launches accumulate; fd_plan_get_timings / fd_plan_enable_timing synchronise the stream, the calls
   themselves never wait for the device (finished spans are harvested with hipEventQuery).
   on: 0 = off, 1 = the diff+decompress kernel only (2 events per call), 2 = every stage and the whole call, 3 = the whole
   call only (2 events per call: what a per-call median should be taken from -- level 2's extra markers cost ~15 us per call),
   4 = every stage without the whole-call markers. */
enum fd_stage { FD_STAGE_EPS = 0, FD_STAGE_PERTURB = 1, FD_STAGE_F = 2, FD_STAGE_DECOMPRESS = 3,
                FD_STAGE_TOTAL = 4, FD_STAGE_EXCHANGE = 5 /* sharded calls: halo + group sums + step sizes (inside FD_STAGE_EPS's span) */,
                FD_NSTAGES = 6 };
int fd_plan_enable_timing(fd_plan *plan, int on);
int fd_plan_get_timings(fd_plan *plan, double *ms_sum /*[FD_NSTAGES]*/, int64_t *launches /*[FD_NSTAGES]*/);
/* Level 1 on a SAMPLE of the calls: only every stride-th call (the first after this call, then every stride-th) carries the two events.
   A pair of events costs ~8 us of stream time per call on MI355X (two marker packets that keep consecutive launches from overlapping
   their start-up: scripts/ubench/ext_launch_probe.hip) -- 10 % of a 75-us Jacobian, 40 % of a 20-us one; sampled, a timed loop runs at
   its untimed speed and the graded kernel is still measured inside it. */
int fd_plan_set_timing_stride(fd_plan *plan, int stride);
/* The individual span durations behind ms_sum (most recent first dropped: the first `cap` spans since timing was enabled,
   in call order; at most 65536 are kept per stage): what a MEDIAN over individually timed runs needs. */
int fd_plan_get_timing_samples(fd_plan *plan, int stage, double *ms_out, int64_t cap, int64_t *n_out);

/* ---- built-in device f! families (the reference's test fixtures + benchmark configs) ---- */
enum fd_builtin_family {
    FD_F_TRIDIAG = 0,      /* params {n}: dx[i] = x[i-1] - 2x[i] + x[i+1]   (test/coloring_tests.jl:5-13) */
    FD_F_TRIDIAG_NL = 1,   /* params {n}: ... + x[i]^2 * x[i+1]             (J depends on x)              */
    FD_F_LAP5 = 2,         /* params {nx,ny}: zero-Dirichlet 5-point Laplacian                              */
    FD_F_CLAMP5 = 3,       /* params {nx,ny}: clamped-edge sum stencil       (test/coloring_tests.jl:99-108) */
    FD_F_BLOCKCOUPLED = 4, /* params {nblk,bs}: x_b[k]*(sig_{b-1}+sig_b+sig_{b+1}) + sin(x_b[k])            */
    FD_F_NONSQUARE = 5,    /* params {n}: (x1-3)^2 + x1*x2 + (x2+4)^2 - 3    (test/coloring_tests.jl:124-133) */
    FD_F_LAP5_NL = 6,      /* params {nx,ny}: FD_F_LAP5 + x[k]^2 * x[k+1]  (same pattern; J depends on x)        */
    FD_F_LAP7 = 7,         /* params {nx,ny,nz}: zero-Dirichlet 7-point Laplacian on an nx x ny x nz grid + x[k]^2 * x[k+1]:      */
                           /*   ((((((x[k-nx*ny] + x[k-nx]) + x[k-1]) + x[k+1]) + x[k+nx]) + x[k+nx*ny]) - 6 x[k]) + (x[k] x[k]) x[k+1]  */
    FD_F_SPARSE = 8        /* fd_builtin_f_create_sparse: f_r = sum over the pattern's entries (r, j), ascending j, of w(r, j) phi(x_j) */
};
int fd_builtin_f_create(fd_ctx *ctx, int family, const int64_t *params, int nparams,
                        fd_f_launch *fn_out, void **fctx_out);
/* A residual with ANY sparsity pattern (FD_F_SPARSE): given the CSC pattern of its Jacobian (M x N, host arrays, the caller's
   index width and base), f_r(x) = sum over the stored entries (r, j) of row r, by ascending j, of w(r, j) * phi(x_j), summed left
   to right, with phi(t) = t + 0.25 t^2 and w(r, j) = 1 + 0.125 ((r + 3 j) mod 8) (0-based r, j).  The launcher keeps the
   transposed pattern on the device.  Its lazy launcher stores the Jacobian column by column (FD_LAZY_CAP_STORE_CSC). */
int fd_builtin_f_create_sparse(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr, const void *rowval, int idx_bytes, int idx_base,
                               fd_f_launch *fn_out, void **fctx_out);
int fd_builtin_f_destroy(void *fctx);
/* number of launcher invocations / points evaluated since creation (call-count parity tests) */
int fd_builtin_f_counts(void *fctx, int64_t *launches, int64_t *points);
/* introspection of a built-in family.  FD_F_INFO_ROW_STORES: storing launches of the sparse family that went row by row
   (k_f_sparse_store_rows: the plan's pattern known to be the one the residual was created from) instead of column by column. */
enum fd_builtin_f_info_key { FD_F_INFO_ROW_STORES = 1 };
int fd_builtin_f_info(void *fctx, int key, int64_t *value);
/* The lazy-point launcher of a built-in family (FD_ERR_UNSUPPORTED if the family has none) and its capabilities. */
int fd_builtin_f_lazy(void *fctx, fd_f_launch_lazy *fn_out);
int fd_builtin_f_lazy_caps(void *fctx, int *caps_out);

/* ---- Jacobian-vector products: finite_difference_jvp!, src/jvp.jl:238-274 ---- */
typedef struct fd_jvp_plan fd_jvp_plan;
/* fdtype: FD_FORWARD or FD_CENTRAL (complex mode is rejected as the reference does, src/jvp.jl:248-250). */
int fd_jvp_plan_create(fd_ctx *ctx, int64_t M, int64_t N, int fdtype, fd_jvp_plan **out);
int fd_jvp_plan_destroy(fd_jvp_plan *plan);
/* jvp_out[M] = J(x) v by finite differences.  x, v share xv_kind; f_in (forward only) may be NULL. */
int fd_jvp(fd_jvp_plan *plan, fd_f_launch f, void *fctx, const void *x, const void *v, int xv_kind,
           const void *f_in, int f_in_kind, double relstep, double absstep, double dir, void *jvp_out,
           int out_kind);
/* Same, device pointers only, enqueues on the context's stream and returns (the inner loop of a Newton-Krylov solver). */
int fd_jvp_async(fd_jvp_plan *plan, fd_f_launch f, void *fctx, const void *x, const void *v, const void *f_in,
                 double relstep, double absstep, double dir, void *jvp_out);
int fd_jvp_get_epsilon(fd_jvp_plan *plan, double *eps_out);

/* Optional lazy-point launcher of a JVP (the JVP counterpart of fd_f_launch_lazy): instead of points materialised by
   the library (x + eps*v, src/jvp.jl:260; x -/+ eps*v, :265-267) the launcher gets x, v and the DEVICE address of
   the step and perturbs while loading -- in Julia: the unchanged f! applied to a lazy `x .+ eps .* v` wrapper.
   It writes f at   forward: member 0 = x + eps*v  (and f(x) into base_out when that is non-NULL);
                    central: member 0 = x - eps*v, member 1 = x + eps*v          (member k at fx_out + k*fx_stride),
   forming every point as  x[j] + (eps*v[j])  /  x[j] - (eps*v[j])  in the element type (=> the same bits as the
   materialised points).  May return FD_LAZY_DECLINED; the library then materialises the points. */
typedef struct fd_lazy_jvp_points {
    const void *x;        /* base point, device, N elements */
    const void *v;        /* direction, device, N elements */
    const void *eps;      /* device address of the ONE step size (element type of the plan) */
    void *base_out;       /* forward arm without f_in: f(x) goes here (M elements); NULL otherwise */
    int central;          /* 0 forward (1 point), 1 central (2 points) */
    int reserved0;
    void *quotient_out;   /* launchers registered with FD_LAZY_JVP_CAP_QUOTIENT only: non-NULL = write the finished       */
                          /* difference quotient here (M elements) -- (f(x+eps v) - f(x)) / eps, or                       */
                          /* (f(x+eps v) - f(x-eps v)) / (2 eps): the subtraction and the division of                     */
                          /* src/jvp.jl:258,263,267,270 inside f!'s launch, each an IEEE operation on the values the plain */
                          /* path would have stored (same bits); fx_out / base_out are not written then                   */
} fd_lazy_jvp_points;
typedef int (*fd_f_launch_lazy_jvp)(void *fctx, void *fx_out, const fd_lazy_jvp_points *pts, int64_t fx_stride,
                                    void *stream);
int fd_jvp_plan_set_lazy_f(fd_jvp_plan *plan, fd_f_launch_lazy_jvp lazy);   /* NULL clears it (and the capabilities) */
#define FD_LAZY_JVP_CAP_QUOTIENT 1   /* honours fd_lazy_jvp_points.quotient_out: the whole JVP is the step-size reduction + ONE f! launch */
int fd_jvp_plan_set_lazy_caps(fd_jvp_plan *plan, int caps);
int fd_builtin_f_lazy_jvp_caps(void *fctx, int *caps_out);
/* The lazy JVP launcher of a built-in family (FD_ERR_UNSUPPORTED if it has none: block-coupled, non-square). */
int fd_builtin_f_lazy_jvp(void *fctx, fd_f_launch_lazy_jvp *fn_out);

/* ---- plan-time colouring: the step BEFORE the path ------------------------- */
/* Greedy distance-1 column colouring of the column intersection graph of a CSC pattern (two columns
   conflict when they share a row) -- what ArrayInterface.matrix_colors / SparseDiffTools hand to
   `colorvec` in the reference's tutorials (docs/src/tutorials.md:117-126).  Host-side, O(sum of
   row_degree * column_degree).  colorvec_out[N] receives colours 1..C (int64); *ncolors_out = C. */
int fd_color_columns_greedy(int64_t M, int64_t N, const void *colptr, const void *rowval, int idx_bytes,
                            int idx_base, int64_t *colorvec_out, int64_t *ncolors_out);
/* Closed-form colouring of a band: colorvec[j] = mod1(j, l+u+1) (valid for any matrix inside the band). */
int fd_color_banded(int64_t N, int64_t l, int64_t u, int64_t *colorvec_out, int64_t *ncolors_out);

/* ---- multi-GPU (one process per GPU): the exchange steps behind the C ABI -------------------------------------------
 * The reference is single-process; these have no counterpart there.  A plan with a column window
 * (fd_plan_opts.col_begin/col_end) fills a contiguous slice of nzval / band data; colour ownership
 * (color_begin/color_end) fills the values of the owned colours.  What remains is ONE exchange, provided here on top of
 * RCCL (loaded at run time with dlopen -- the instance a host framework already carries if there is one; libfdjac itself
 * does not link it): every call enqueues on the context's stream and returns.
 *   bootstrap: rank 0 calls fd_comm_unique_id and ships the FD_COMM_ID_BYTES bytes to the other ranks by whatever the
 *   host has (MPI.jl, a torch.distributed store, a file); every rank then calls fd_comm_create. */
typedef struct fd_comm fd_comm;
#define FD_COMM_ID_BYTES 128
int fd_comm_unique_id(void *id_out /* FD_COMM_ID_BYTES bytes, host */);
int fd_comm_create(fd_ctx *ctx, int nranks, int rank, const void *id, fd_comm **out);
int fd_comm_destroy(fd_comm *comm);
int fd_comm_info(const fd_comm *comm, int *nranks, int *rank, int *rccl_version);   /* any out pointer may be NULL */
const char *fd_comm_library(void);   /* which librccl was bound ("" if none could be) */
/* Assemble the Jacobian from column-range slices, every rank gets all of it: buf holds nranks slots of slot_elems
   elements, rank r computed its slice into slot r (pass outs[0] = buf + r*slot_elems to fd_jacobian_async); ONE in-place
   ncclAllGather.  elem_bytes: 8 (Float64), 4 (Float32) or 1. */
int fd_comm_allgather(fd_comm *comm, void *buf, int64_t slot_elems, int elem_bytes);
/* Assemble on one rank ("a single RCCL gather ... to assemble nzval"): rank r sends its send_elems == counts[r] values,
   the root receives slice r at recv + displs[r] (elements) -- one group of point-to-point transfers, each over its own
   xGMI link.  recv is only read on the root; the root's own slice is copied unless send already points there. */
int fd_comm_gatherv(fd_comm *comm, const void *send, int64_t send_elems, void *recv, const int64_t *counts,
                    const int64_t *displs, int elem_bytes, int root);
/* Assemble outputs computed under colour ownership (they start from zero; every value is non-zero on one rank only,
   so the sum is exact): in-place ncclAllReduce(sum).  elem_bytes 8 or 4. */
int fd_comm_allreduce_sum(fd_comm *comm, void *buf, int64_t n, int elem_bytes);
/* Colour ownership, assembled -- north_star's split ("each GPU owns a disjoint subset of colours ... a single RCCL gather to assemble
   nzval") as ONE entry point that is right call after call: a plan created with fd_plan_opts.color_begin / color_end writes only the
   stored values of its colours' columns (ext/FiniteDiffSparseArraysExt.jl:38-47 for those colours) and leaves the rest of outs
   untouched, so summing the ranks' outputs is exact only while the untouched entries are zero -- true for a fresh buffer, false from
   the second call on.  This call zero-fills outs, runs fd_jacobian_async on the plan's colours and sums every output over the
   communicator's ranks in place (one ncclAllReduce per output; every stored value is non-zero on exactly one rank: x + 0 == x).
   comm = NULL: no collective -- the caller sums (MPI.jl, a host transport). */
int fd_jacobian_owned_async(fd_plan *plan, fd_comm *comm, fd_f_launch f, void *fctx, const void *x, const void *f_in,
                            double relstep, double absstep, double dir, void *const *outs);
int fd_comm_broadcast(fd_comm *comm, void *buf, int64_t n, int elem_bytes, int root);   /* e.g. a new x from rank 0 */
/* Neighbour exchange of the boundary values of a vector sharded by contiguous ranges -- the x of a time-stepping loop whose
   rows are sharded like the Jacobian's columns (the sharded solve returns y by rows; the next Jacobian needs x plus the l + u
   values beside its range, NOT all of x).  buf is a device vector in GLOBAL indexing of which this rank owns
   [own_begin, own_end): it sends its first `halo` owned elements to rank - 1 and its last `halo` to rank + 1 and receives
   buf[own_begin - halo, own_begin) from rank - 1 and buf[own_end, own_end + halo) from rank + 1: one group of at most four
   point-to-point transfers (2 x halo elements per link).  The end ranks skip the neighbour they do not have. */
int fd_comm_halo_exchange(fd_comm *comm, void *buf, int64_t own_begin, int64_t own_end, int64_t halo, int elem_bytes);

/* ---- small messages by direct peer-to-peer stores (one node, one process per GPU) ----------------------------------------------
 * The per-step exchanges of a time-stepping loop -- the halo of x, the partial sums of the step-size reduction (62 KB), the
 * interface packets of the sharded solve (64 B per rank) -- are a few kilobytes each: as RCCL collectives they cost a
 * general-purpose machine's latency (10-25 us each), as stores into a peer's HBM over xGMI a few microseconds.  Every rank owns a
 * MAILBOX in its HBM and maps its peers' (hipIpc handles); an exchange is ONE kernel on the context's stream: every workgroup first puts (copies its
 * share into its peer's mailbox, system-scope fence, the last share raises a flag), then waits (polls the same peer's flag, bounded
 * by FDJAC_P2P_TIMEOUT_MS = 2000 by default -- a missing peer raises fd_p2p_status instead of hanging the GPU -- then copies out).  RCCL stays for the bulk assembly of nzval.
 *   bootstrap: fd_p2p_create on every rank; every rank's fd_p2p_local_handle (FD_P2P_HANDLE_BYTES bytes) reaches every rank by
 *   whatever the host has (MPI.jl Allgather, a torch.distributed store); fd_p2p_connect(all handles in rank order) -- or
 *   fd_comm_enable_p2p, which does all of that over the RCCL communicator and routes the communicator's small messages
 *   (fd_comm_allgather / fd_comm_halo_exchange up to slot_bytes, hence the sharded step-size reduction and the sharded solve)
 *   through the mailbox from then on. */
typedef struct fd_p2p fd_p2p;
#define FD_P2P_HANDLE_BYTES 64
int fd_p2p_create(fd_ctx *ctx, int nranks, int rank, int64_t slot_bytes /* per rank per exchange, <= 16 MiB */, fd_p2p **out);
int fd_p2p_local_handle(fd_p2p *p2p, void *handle_out /* FD_P2P_HANDLE_BYTES bytes, host */);
int fd_p2p_connect(fd_p2p *p2p, const void *handles /* nranks x FD_P2P_HANDLE_BYTES bytes, rank order, host */);
int fd_p2p_destroy(fd_p2p *p2p);
/* A LOOP-BACK mailbox: rank `rank` of an `nranks`-rank job on ONE device.  Every peer's mailbox is mapped to a local sink, no wait
   ever spins, and the senders' slots hold what fd_p2p_loopback_fill put there (`bytes` bytes of device or host memory at byte
   `offset` of `sender`'s slot: its group sums, then its halo of x -- the layout of the per-step exchange).  A plan with this mailbox
   attached runs one rank's share of the sharded step -- the same launches and stores -- with nobody else present: bench.py's
   `rank_share` measurement and the single-GPU tests of the sharded call (no connect step). */
int fd_p2p_create_loopback(fd_ctx *ctx, int nranks, int rank, int64_t slot_bytes, fd_p2p **out);
int fd_p2p_loopback_fill(fd_p2p *p2p, int sender, int64_t offset, const void *data, int64_t bytes);
/* ... and for the FUSED sharded step (one launch per Jacobian, every exchanged value its own flag): all 64 x 8 group sums of the
   reduction (doubles, group-major; the rank's own groups are ignored) and what the lower / upper neighbour would deliver as halo
   (halo_bytes each, <= 64; NULL: none). */
int fd_p2p_loopback_fill_fused(fd_p2p *p2p, const void *gsum64x8, const void *halo_lo, const void *halo_hi, int64_t halo_bytes);
int fd_p2p_info(const fd_p2p *p2p, int *nranks, int *rank, int64_t *slot_bytes, int *uncached);   /* any out pointer may be NULL */
/* 0 while every exchange completed; 1 + r once a wait for rank r timed out (sticky; no synchronisation needed to read it) */
int fd_p2p_status(const fd_p2p *p2p, int *timed_out_rank_plus_1);
/* in place, like fd_comm_allgather: buf holds nranks slots of `bytes` bytes (a multiple of 8, <= slot_bytes), this rank's in slot `rank` */
int fd_p2p_allgather(fd_p2p *p2p, void *buf, int64_t bytes);
/* like fd_comm_halo_exchange (halo * elem_bytes a multiple of 8, twice that <= slot_bytes) */
int fd_p2p_halo_exchange(fd_p2p *p2p, void *buf, int64_t own_begin, int64_t own_end, int64_t halo, int elem_bytes);
/* Create a mailbox for the communicator's ranks, exchange the handles over RCCL and route the communicator's small messages
   through it (collective: every rank calls it; blocking).  FD_ERR_COMM if the peers cannot be mapped (ranks on several nodes). */
int fd_comm_enable_p2p(fd_comm *comm, int64_t slot_bytes);
/* does the communicator route its small messages through a mailbox (*enabled), and has a wait of it ever timed out (as fd_p2p_status)? */
int fd_comm_p2p_status(const fd_comm *comm, int *enabled, int *timed_out_rank_plus_1);
/* drop the mailbox again: the communicator's small messages go back to RCCL (every rank calls it -- e.g. after a timed-out wait) */
int fd_comm_disable_p2p(fd_comm *comm);

/* Sharded step-size reduction (src/jacobians.jl:559-561 / 600-602 across ranks).  The reduction is DEFINED as a two-level sum that
   depends on N alone: x is cut into 64 contiguous groups, a group's blocks are added in block order, the 64 group sums in group order.
   By default every rank reduces the whole (replicated) x in one launch -- no communication, identical step sizes everywhere.  With a
   communicator (or a bare mailbox) attached, rank r of W reduces only ITS groups -- groups [r * ceil(64 / W), ...): the part of x
   that fd_plan_eps_shard_range reports, so cut the column ownership there -- and the ranks exchange their group sums (64 / W x 8
   doubles each: 512 B at W = 8), then add the 64 group sums in the same order: bit-identical step sizes on every rank, whatever W.
   With a mailbox (fd_comm_enable_p2p, or fd_plan_set_p2p for callers without RCCL) the exchange is ONE launch per call, whose last
   workgroup writes the step sizes; without, an in-place all-gather + one tiny launch.  comm / p2p = NULL detaches.  Both must live on
   the plan's context.  Plans whose reduction cannot be sharded (more than 8 colours, N <= 16384, dense arm, complex step: no reduction
   at all) keep the replicated reduction. */
int fd_plan_set_comm(fd_plan *plan, fd_comm *comm);
int fd_plan_set_p2p(fd_plan *plan, fd_p2p *p2p);
/* x is SHARDED too (a time-stepping loop: rank r holds x[own_begin, own_end) of the global vector it addresses, plus `halo` elements
   either side that f! reads): every following fd_jacobian* call of a plan with a communicator / mailbox exchanges the halo with the
   neighbour ranks itself -- it rides in the same launch as the group sums (mailbox) or as one grouped send / recv (RCCL) -- and
   WRITES x[own_begin - halo, own_begin) and x[own_end, own_end + halo) although x is declared const.  halo = 0 turns it off.
   FD_ERR_UNSUPPORTED for plans whose reduction cannot be sharded (exchange the halo with fd_comm_halo_exchange then). */
int fd_plan_set_halo(fd_plan *plan, int64_t own_begin, int64_t own_end, int64_t halo);
/* The same reduction in explicit pieces, for callers that exchange the group sums themselves (MPI.jl, tests):
   fd_plan_eps_partials enqueues shard `shard` of `nshards` (<= 64) and returns the device address of the group-sum buffer
   (nshards slots of *slot_doubles_out doubles; shard r fills slot r); after the exchange fd_plan_eps_finalize turns the
   complete buffer into step sizes; fd_plan_set_eps_mode(plan, FD_EPS_PRECOMPUTED) makes the following fd_jacobian* calls
   use them instead of reducing x again (FD_EPS_COMPUTE restores the default).  FD_ERR_UNSUPPORTED for plans whose
   reduction cannot be sharded. */
enum fd_eps_mode { FD_EPS_COMPUTE = 0, FD_EPS_PRECOMPUTED = 1 };
int fd_plan_eps_partials(fd_plan *plan, const void *x_dev, int shard, int nshards, void **partials_out,
                         int64_t *slot_doubles_out);
int fd_plan_eps_finalize(fd_plan *plan, double relstep, double absstep, double dir);
int fd_plan_set_eps_mode(fd_plan *plan, int mode);
/* The elements of x shard `shard` of `nshards` of the reduction reads, [*x_begin, *x_end): the shard's groups -- cut the column /
   row ownership of a multi-GPU run at these boundaries and every rank reduces exactly what it owns. */
int fd_plan_eps_shard_range(fd_plan *plan, int shard, int nshards, int64_t *x_begin, int64_t *x_end);

/* ---- the consumer: tridiagonal solve with the Jacobian where fd_jacobian_async left it -----------
 * Solves (alpha*I + beta*J) y = b on the device for a tridiagonal J -- the linear system of an implicit / Rosenbrock
 * step with a Tridiagonal jac_prototype (test/downstream/ordinarydiffeq_tridiagonal_solve.jl:18-30: W = I - gamma*J).
 * J is taken in the storage the Jacobian plans write:
 *   FD_TRI_DIAGONALS  J = {dl, d, du}: the outs of a fd_plan_create_tridiagonal plan (window-relative for a column window)
 *   FD_TRI_CSC        J = {nzval}: the stored values of a tridiagonal SparseMatrixCSC (3N-2 values; for a column window
 *                     the window's slice, as a fd_plan_create_csc plan with col_begin/col_end fills it)
 * rows [row_begin,row_end) = the columns this rank owns (0,0 = the whole matrix).  b and y hold the local rows.
 * Recursive partition method (Wang) + parallel cyclic reduction at the top, no pivoting: for diagonally dominant
 * systems.  With a communicator the ranks solve ONE global system: the Jacobian never leaves the GPUs that computed
 * it -- the exchange is one all-gather of 8 numbers per rank.  Everything is enqueued on the context's stream. */
typedef struct fd_tridiag_solver fd_tridiag_solver;
enum fd_tridiag_layout { FD_TRI_DIAGONALS = 0, FD_TRI_CSC = 1 };
int fd_tridiag_solver_create(fd_ctx *ctx, int64_t N, int64_t row_begin, int64_t row_end, int layout,
                             fd_tridiag_solver **out);
int fd_tridiag_solver_destroy(fd_tridiag_solver *solver);
/* The elimination does not pivot (LinearAlgebra's Tridiagonal \ does: test/downstream/ordinarydiffeq_tridiagonal_solve.jl:18-30).
   Every solve checks every row it fetches: a row that is not diagonally dominant (|alpha + beta J[i,i]| < |beta J[i,i-1]| +
   |beta J[i,i+1]|) raises bit 0 of the solver's status word, and -- the REFUSAL policy, default -- that solve writes NaN into y instead
   of a solution it cannot vouch for: a non-dominant I - gamma J never returns silently wrong numbers (a Rosenbrock / implicit step
   then fails as loudly as a NaN residual does, and fd_tridiag_solver_status says why).  fd_tridiag_solver_set_policy(solver, 1) turns
   the refusal off for callers who know their matrix is fine without dominance (symmetric positive definite, M-matrices): the flag is
   still raised, y is the elimination's result.  In a MULTI-RANK solve a rank that refuses does so for every rank: its interface packet
   carries NaN, every rank's y is poisoned and every rank's status has bit 1 set ("a peer refused"; bit 0 stays with the rank that owns the
   offending rows).  fd_tridiag_solver_status synchronises the context's stream. */
int fd_tridiag_solver_set_policy(fd_tridiag_solver *solver, int trust_non_dominant);
int fd_tridiag_solver_status(fd_tridiag_solver *solver, int *flags_out);
/* comm == NULL: the solver's rows are the whole system.  J: 3 (diagonals) or 1 (CSC) device pointers; b, y device. */
int fd_tridiag_solve_async(fd_tridiag_solver *solver, double alpha, double beta, const void *const *J, const void *b,
                           void *y, fd_comm *comm);
/* The same solve in its two local phases, for callers that exchange the 8-double packets themselves (MPI.jl, tests):
   _interface writes this rank's packet (device, 8 doubles); after the all-gather of all ranks' packets (nranks x 8
   doubles, device) _finish solves the interface system and the local rows. */
int fd_tridiag_solve_interface(fd_tridiag_solver *solver, double alpha, double beta, const void *const *J, const void *b,
                               void *packet_dev);
int fd_tridiag_solve_finish(fd_tridiag_solver *solver, double alpha, double beta, const void *const *J, const void *b,
                            const void *packets_dev, int rank, int nranks, void *y);

/* ---- the consumer for wider bands: (alpha*I + beta*J) y = b for a BANDED J, 0 <= l, u <= 4 (round 6) --------------------------------
 * The linear system of an implicit / Rosenbrock step whose jac_prototype is a BandedMatrix (pentadiagonal, ...), on J's own device
 * storage as the banded plans write it:
 *   FD_BAND_SOLVE_BANDED  J = BandedMatrix data, (l+u+1) x N column-major (the out of a fd_plan_create_banded plan)
 *   FD_BAND_SOLVE_CSC     J = nzval of the exact band as SparseMatrixCSC (the out of a fd_plan_create_csc plan of that pattern)
 * Block cyclic reduction with K x K blocks, K = max(l, u), no pivoting, Float64 arithmetic whatever the element type; level 0 is
 * read straight from J and b.  One GPU, the whole matrix (the tridiagonal case, l = u = 1, also has the faster and shardable
 * fd_tridiag_solve_async).  Policy and status as for the tridiagonal solver: every row of alpha*I + beta*J is checked for diagonal
 * dominance; a solve that meets a row that is not raises status bit 0 and REFUSES (y = NaN) unless
 * fd_banded_solver_set_policy(solver, 1).  Everything is enqueued on the context's stream; fd_banded_solver_status synchronises it. */
typedef struct fd_banded_solver fd_banded_solver;
enum fd_banded_solve_layout { FD_BAND_SOLVE_BANDED = 0, FD_BAND_SOLVE_CSC = 1 };
int fd_banded_solver_create(fd_ctx *ctx, int64_t N, int l, int u, int layout, fd_banded_solver **out);
int fd_banded_solver_destroy(fd_banded_solver *solver);
int fd_banded_solver_set_policy(fd_banded_solver *solver, int trust_non_dominant);
int fd_banded_solver_status(fd_banded_solver *solver, int *flags_out);
int fd_banded_solve_async(fd_banded_solver *solver, double alpha, double beta, const void *J, const void *b, void *y);

/* ---- the consumer for block-banded Jacobians: (alpha*I + beta*J) y = b for a BLOCK-TRIDIAGONAL J (round 6) ---------------------------
 * J = nblk x nblk dense blocks of block_size x block_size (<= 32), block bandwidths (1, 1), in BlockBandedMatrix data as a
 * fd_plan_create_blockbanded plan of uniform block sizes fills it (block column J's in-band blocks stacked into one column-major
 * panel, the panels one after the other: ext/FiniteDiffBlockBandedMatricesExt.jl:44-68) -- BASELINE's config 5.  Block cyclic
 * reduction, one workgroup per block row, Gauss-Jordan in LDS, no pivoting, Float64 arithmetic.  One GPU.  Policy and status as for the
 * other consumers: a row of alpha*I + beta*J without diagonal dominance raises status bit 0 and the solve REFUSES (y = NaN) unless
 * fd_blocktridiag_solver_set_policy(solver, 1). */
typedef struct fd_blocktridiag_solver fd_blocktridiag_solver;
int fd_blocktridiag_solver_create(fd_ctx *ctx, int64_t nblk, int block_size, fd_blocktridiag_solver **out);
int fd_blocktridiag_solver_destroy(fd_blocktridiag_solver *solver);
int fd_blocktridiag_solver_set_policy(fd_blocktridiag_solver *solver, int trust_non_dominant);
int fd_blocktridiag_solver_status(fd_blocktridiag_solver *solver, int *flags_out);
int fd_blocktridiag_solve_async(fd_blocktridiag_solver *solver, double alpha, double beta, const void *data, const void *b, void *y);

/* ---- runtime compilation of a row functor (hiprtc) --------------------------------------------------------------------------------
 * The reference accepts ANY callable as f! (src/jacobians.jl:541,563,605-606,634).  The one-launch call of this library needs f! as
 * device code; a caller without an offline toolchain (a Julia process) hands it over as SOURCE: a functor type
 *     struct MyF { <parameters>;  template <class P> __device__ real_t operator()(long long r, const P &X) const { ... } };
 * that returns row r of the residual at the point X (X(j) = coordinate j; `real_t` is the element type: double / float).  It is
 * compiled against include/fdjac_device.h (embedded in the library) with -O3 -ffp-contract=off for gfx950 and cached by content for
 * the life of the process.  `params` = the functor OBJECT byte for byte (params_bytes == sizeof(MyF); 0 for an empty struct).
 * Out: the plain launcher (rows at materialised points), the lazy launcher -- FD_LAZY_CAP_STORE_CSC | FD_LAZY_CAP_STORE_CSC_BASE: for a
 * plan created with FD_PLAN_STORE_CSC (| FD_PLAN_STORE_CSC_ALWAYS on banded patterns) the whole Jacobian is the step-size launch + ONE
 * launch of fd_csc_store_cols / fd_csc_store_cols_win instantiated for the functor -- and the context for both.
 * FD_ERR_ARG when the source does not compile (the compiler's messages: fd_f_compile_log(), this thread's last compilation);
 * FD_ERR_UNSUPPORTED when libhiprtc cannot be loaded.  Register the pair as
 *     fd_plan_set_lazy_f(plan, lazy); fd_plan_set_lazy_caps(plan, caps);          (fd32_* for elem_bytes == 4) */
int fd_f_compile_rows(fd_ctx *ctx, const char *source, const char *functor_type, const void *params, int64_t params_bytes, int64_t M,
                      int64_t N, int elem_bytes, fd_f_launch *fn_out, fd_f_launch_lazy *lazy_out, int *lazy_caps_out, void **fctx_out);
/* The same for a row function given as LLVM BITCODE -- what AMDGPU.jl / GPUCompiler emit for a Julia closure (the reference's "any
 * callable", src/jacobians.jl:563,634, on the one-launch path), or `hipcc --offload-arch=gfx950 -fgpu-rdc -emit-llvm
 * --offload-device-only -c` for a C function.  The bitcode defines, for the element type T of the call,
 *     extern "C" __device__ T    fdjac_user_row  (const void *params, long long r, const fd_cpoint *X);               (required)
 *     extern "C" __device__ void fdjac_user_row_c(const void *params, long long r, const fd_cpoint *X, T *re_im);   (complex step; optional)
 * and reads coordinate j of the point through the library's side of the link:
 *     extern "C" __device__ T    fdjac_point_get  (const fd_cpoint *X, long long j);
 *     extern "C" __device__ void fdjac_point_get_c(const fd_cpoint *X, long long j, T *re_im);
 * (`struct fd_cpoint { int kind; const void *obj; }` -- opaque to the caller).  The library compiles its kernels for a functor that
 * calls fdjac_user_row, links the two (hiprtcLink*, LLVM bitcode inputs; the row function is inlined into the kernels) and returns the
 * launchers of fd_f_compile_rows: the plain one, and the lazy one with the column store (any pattern), the band store (exact bands) and
 * -- if fdjac_user_row_c is defined -- the complex step.  `params`: up to 4096 bytes handed to the row function as they are.
 * FD_ERR_ARG when the link fails (fd_f_compile_log()), FD_ERR_UNSUPPORTED without libhiprtc. */
int fd_f_link_rows_bitcode(fd_ctx *ctx, const void *bitcode, int64_t bitcode_bytes, const void *params, int64_t params_bytes, int64_t M,
                           int64_t N, int elem_bytes, fd_f_launch *fn_out, fd_f_launch_lazy *lazy_out, int *lazy_caps_out, void **fctx_out);
/* A SEPARABLE residual from its TERM alone (round 6; include/fdjac_device.h, "SEPARABLE residuals"): row r of f is the left-to-right
 * sum, over the stored entries (r, j) of the Jacobian's own pattern in ascending j, of a term of ONE coordinate.  `source` defines
 *     struct MyTerms { <parameters>;  template <class T> __device__ T term(long long r, long long j, T v) const { ... } };
 * (generic in T: the complex step instantiates it on fd_cplx<real_t>).  row_ptr_dev / row_col_dev: the pattern by rows on the device --
 * the lists of the plan the functor will be used with (fd_plan_row_lists of a plan created with FD_PLAN_STORE_CSC |
 * FD_PLAN_STORE_CSC_ROWS), which must outlive the functor; plan_serial: that plan's serial.  The library wraps the terms into
 * fd_sep_rows<MyTerms> and returns the launchers of fd_f_compile_rows -- plain evaluation, column store, complex step, all from `term`
 * -- plus the ROW-WISE store: on the plan the lists came from (verified colouring, reach <= 700, M == N) the Jacobian is ONE launch of
 * fd_csc_store_rows, 2 L term evaluations per row of L entries instead of L^2, bit-identical to the column store.  Any other plan takes
 * the column store.  `params` = the MyTerms object byte for byte (0 bytes for an empty struct). */
int fd_f_compile_terms(fd_ctx *ctx, const char *source, const char *terms_type, const void *params, int64_t params_bytes, int64_t M, int64_t N,
                       int elem_bytes, const void *row_ptr_dev, const void *row_col_dev, uint64_t plan_serial, fd_f_launch *fn_out,
                       fd_f_launch_lazy *lazy_out, int *lazy_caps_out, void **fctx_out);
int fd_f_compiled_destroy(void *fctx);
int fd_f_compiled_counts(void *fctx, int64_t *launches);
/* launches of the ROW-WISE store among them (fd_f_compile_terms functors; 0 for the others) */
int fd_f_compiled_row_stores(void *fctx, int64_t *row_stores);
const char *fd_f_compile_log(void);

/* Device stream-copy ceiling probe: copies `bytes` device-to-device `iters` times with a
   16 B/lane kernel and returns the achieved GB/s (read + write bytes) -- the measured roofline
   the achieved figures are quoted against. */
int fd_stream_copy_gbps(fd_ctx *ctx, int64_t bytes, int iters, double *gbps_out);

/* ---- Float32 instantiation -------------------------------------------------------------------------------------
 * The reference is generic in eltype(x) (JacobianCache{...,returntype}); every function above that touches values
 * exists a second time with the prefix fd32_ for Float32 problems: x, f_in, the f! arrays and the outputs are
 * `float` (complex step: (re,im) float pairs), step sizes follow the same rules evaluated in Float32
 * (default_relstep = sqrt / cbrt of eps(Float32), src/epsilons.jl:133-144), relstep / absstep / dir and the reported
 * epsilons stay `double` arguments.  Contexts (fd_ctx_*), fd_last_error, the colouring helpers, fd_plan_opts,
 * fd_lazy_points and the launcher types are shared.  Masked norms are accumulated in Float64 in both instantiations.
 */
typedef struct fd32_plan fd32_plan;
typedef struct fd32_jvp_plan fd32_jvp_plan;
int fd32_plan_create_csc(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr, const void *rowval,
                       int idx_bytes, int idx_base, const void *colorvec, int color_bytes,
                       const fd_plan_opts *opts, fd32_plan **out);
int fd32_plan_create_csc_device(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr_dev, const void *rowval_dev,
                                int idx_bytes, int idx_base, const void *colorvec_dev, int color_bytes,
                                const fd_plan_opts *opts, fd32_plan **out);
int fd32_plan_checksum(fd32_plan *plan, uint64_t *checksum_out);
int fd32_plan_create_csc_dense(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr,
                             const void *rowval, int idx_bytes, int idx_base, const void *colorvec,
                             int color_bytes, const fd_plan_opts *opts, fd32_plan **out);
int fd32_plan_create_coo_dense(fd_ctx *ctx, int64_t M, int64_t N, const void *rows_index,
                             const void *cols_index, int64_t nnz, int idx_bytes, int idx_base,
                             const void *colorvec, int color_bytes, const fd_plan_opts *opts,
                             fd32_plan **out);
int fd32_plan_create_entries(fd_ctx *ctx, int64_t M, int64_t N, const void *rows_index,
                           const void *cols_index, const int64_t *dest, int64_t nnz,
                           int64_t out_len, int idx_bytes, int idx_base, const void *colorvec,
                           int color_bytes, const fd_plan_opts *opts, fd32_plan **out);
int fd32_plan_create_dense(fd_ctx *ctx, int64_t M, int64_t N, int64_t ncols, const fd_plan_opts *opts,
                         fd32_plan **out);
int fd32_plan_create_tridiagonal(fd_ctx *ctx, int64_t N, const void *colorvec, int color_bytes,
                               const fd_plan_opts *opts, fd32_plan **out);
int fd32_plan_create_banded(fd_ctx *ctx, int64_t M, int64_t N, int64_t l, int64_t u,
                          const void *colorvec, int color_bytes, const fd_plan_opts *opts,
                          fd32_plan **out);
int fd32_plan_create_blockbanded(fd_ctx *ctx, int64_t nblk, const void *blk_sizes, int64_t bl,
                               int64_t bu, const void *block_starts, const void *block_strides,
                               int idx_bytes, int idx_base, const void *colorvec, int color_bytes,
                               const fd_plan_opts *opts, fd32_plan **out);
int fd32_plan_create_bandedblockbanded(fd_ctx *ctx, int64_t nblk, const void *blk_sizes, int64_t bl, int64_t bu, int64_t lam, int64_t mu,
                                       const void *block_starts, const void *block_strides, int64_t data_len, int idx_bytes, int idx_base,
                                       const void *colorvec, int color_bytes, const fd_plan_opts *opts, fd32_plan **out);
int fd32_plan_destroy(fd32_plan *plan);
int fd32_plan_matches(fd32_plan *plan, const fd_pattern_arrays *now, int *matches_out);
int fd32_plan_info(const fd32_plan *plan, int key, int64_t *value);
int fd32_plan_row_lists(const fd32_plan *plan, const void **row_ptr_dev, const void **row_col_dev, const void **row_slot_dev, int64_t *entries,
                        uint64_t *plan_serial);
int fd32_jacobian(fd32_plan *plan, fd_f_launch f, void *fctx, const void *x, int x_kind,
                const void *f_in, int f_in_kind, double relstep, double absstep, double dir,
                void *const *outs, int out_kind);
int fd32_jacobian_async(fd32_plan *plan, fd_f_launch f, void *fctx, const void *x, const void *f_in,
                      double relstep, double absstep, double dir, void *const *outs);
int fd32_jacobian_owned_async(fd32_plan *plan, fd_comm *comm, fd_f_launch f, void *fctx, const void *x, const void *f_in,
                              double relstep, double absstep, double dir, void *const *outs);
int fd32_plan_set_lazy_f(fd32_plan *plan, fd_f_launch_lazy lazy);
int fd32_plan_set_lazy_caps(fd32_plan *plan, int caps);
int fd32_plan_get_epsilons(fd32_plan *plan, double *eps_out);
int fd32_plan_fused_trace(fd32_plan *plan, long long *marks16);
typedef struct fd32_blocktridiag_solver fd32_blocktridiag_solver;
int fd32_blocktridiag_solver_create(fd_ctx *ctx, int64_t nblk, int block_size, fd32_blocktridiag_solver **out);
int fd32_blocktridiag_solver_destroy(fd32_blocktridiag_solver *solver);
int fd32_blocktridiag_solver_set_policy(fd32_blocktridiag_solver *solver, int trust_non_dominant);
int fd32_blocktridiag_solver_status(fd32_blocktridiag_solver *solver, int *flags_out);
int fd32_blocktridiag_solve_async(fd32_blocktridiag_solver *solver, double alpha, double beta, const void *data, const void *b, void *y);
typedef struct fd32_banded_solver fd32_banded_solver;
int fd32_banded_solver_create(fd_ctx *ctx, int64_t N, int l, int u, int layout, fd32_banded_solver **out);
int fd32_banded_solver_destroy(fd32_banded_solver *solver);
int fd32_banded_solver_set_policy(fd32_banded_solver *solver, int trust_non_dominant);
int fd32_banded_solver_status(fd32_banded_solver *solver, int *flags_out);
int fd32_banded_solve_async(fd32_banded_solver *solver, double alpha, double beta, const void *J, const void *b, void *y);
typedef struct fd32_tridiag_solver fd32_tridiag_solver;
int fd32_tridiag_solver_create(fd_ctx *ctx, int64_t N, int64_t row_begin, int64_t row_end, int layout,
                               fd32_tridiag_solver **out);
int fd32_tridiag_solver_destroy(fd32_tridiag_solver *solver);
int fd32_tridiag_solver_set_policy(fd32_tridiag_solver *solver, int trust_non_dominant);
int fd32_tridiag_solver_status(fd32_tridiag_solver *solver, int *flags_out);
int fd32_tridiag_solve_async(fd32_tridiag_solver *solver, double alpha, double beta, const void *const *J, const void *b,
                             void *y, fd_comm *comm);
int fd32_tridiag_solve_interface(fd32_tridiag_solver *solver, double alpha, double beta, const void *const *J,
                                 const void *b, void *packet_dev);
int fd32_tridiag_solve_finish(fd32_tridiag_solver *solver, double alpha, double beta, const void *const *J, const void *b,
                              const void *packets_dev, int rank, int nranks, void *y);
int fd32_plan_set_comm(fd32_plan *plan, fd_comm *comm);
int fd32_plan_matches_async(fd32_plan *plan, const fd_pattern_arrays *now);
int fd32_plan_stale(fd32_plan *plan, int *stale_out);
int fd32_plan_set_p2p(fd32_plan *plan, fd_p2p *p2p);
int fd32_plan_set_halo(fd32_plan *plan, int64_t own_begin, int64_t own_end, int64_t halo);
int fd32_plan_eps_partials(fd32_plan *plan, const void *x_dev, int shard, int nshards, void **partials_out,
                           int64_t *slot_doubles_out);
int fd32_plan_eps_finalize(fd32_plan *plan, double relstep, double absstep, double dir);
int fd32_plan_set_eps_mode(fd32_plan *plan, int mode);
int fd32_plan_eps_shard_range(fd32_plan *plan, int shard, int nshards, int64_t *x_begin, int64_t *x_end);
int fd32_plan_enable_timing(fd32_plan *plan, int on);
int fd32_plan_get_timings(fd32_plan *plan, double *ms_sum /*[FD_NSTAGES]*/, int64_t *launches /*[FD_NSTAGES]*/);
int fd32_plan_set_timing_stride(fd32_plan *plan, int stride);
int fd32_plan_get_timing_samples(fd32_plan *plan, int stage, double *ms_out, int64_t cap, int64_t *n_out);
int fd32_builtin_f_create(fd_ctx *ctx, int family, const int64_t *params, int nparams,
                        fd_f_launch *fn_out, void **fctx_out);
int fd32_builtin_f_create_sparse(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr, const void *rowval, int idx_bytes, int idx_base,
                                 fd_f_launch *fn_out, void **fctx_out);
int fd32_builtin_f_destroy(void *fctx);
int fd32_builtin_f_counts(void *fctx, int64_t *launches, int64_t *points);
int fd32_builtin_f_info(void *fctx, int key, int64_t *value);
int fd32_builtin_f_lazy(void *fctx, fd_f_launch_lazy *fn_out);
int fd32_builtin_f_lazy_caps(void *fctx, int *caps_out);
int fd32_jvp_plan_create(fd_ctx *ctx, int64_t M, int64_t N, int fdtype, fd32_jvp_plan **out);
int fd32_jvp_plan_destroy(fd32_jvp_plan *plan);
int fd32_jvp(fd32_jvp_plan *plan, fd_f_launch f, void *fctx, const void *x, const void *v, int xv_kind,
           const void *f_in, int f_in_kind, double relstep, double absstep, double dir, void *jvp_out,
           int out_kind);
int fd32_jvp_async(fd32_jvp_plan *plan, fd_f_launch f, void *fctx, const void *x, const void *v, const void *f_in,
                 double relstep, double absstep, double dir, void *jvp_out);
int fd32_jvp_get_epsilon(fd32_jvp_plan *plan, double *eps_out);
int fd32_jvp_plan_set_lazy_f(fd32_jvp_plan *plan, fd_f_launch_lazy_jvp lazy);
int fd32_builtin_f_lazy_jvp(void *fctx, fd_f_launch_lazy_jvp *fn_out);
int fd32_jvp_plan_set_lazy_caps(fd32_jvp_plan *plan, int caps);
int fd32_builtin_f_lazy_jvp_caps(void *fctx, int *caps_out);

#ifdef __cplusplus
}
#endif
#endif /* FDJAC_H */
