/*
 * fd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Single-threaded CPU restatement of FiniteDiff.jl's coloured sparse-Jacobian
 * path, pass-for-pass faithful to the reference's loop structure (separate
 * mask / norm / perturb / f! / diff / decompress / un-perturb passes, Int64
 * indices, add-then-subtract perturbation, IEEE division).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 *
 * Parity status: PINNED -- the reference (Julia) cannot run in this image, so
 * the restatement is pinned against every known-answer the reference's own
 * tests hold for the path (tests/test_oracle_golden.py; fixtures listed in
 * SURVEY.md section 8c).
 *
 * Reference lines followed (relative to /root/reference):
 *   src/jacobians.jl:504-653   cached in-place finite_difference_jacobian!
 *   src/jacobians.jl:446-471   cache-less wrapper (f(fx,x) once for forward)
 *   src/jacobians.jl:473-488   _findstructralnz(::DenseMatrix)
 *   src/jacobians.jl:277-331   out-of-place dense forward (config 1 plumbing)
 *   src/epsilons.jl:26-29,50-53,104-107,133-144   step-size rules
 *   src/iteration_utils.jl:25-32                  generic COO decompression
 *   src/jvp.jl:238-274                            finite_difference_jvp!
 *   ext/FiniteDiffSparseArraysExt.jl:20-28,38-47  CSC decompression (general / common pattern)
 *   ext/FiniteDiffBandedMatricesExt.jl:13-27      banded decompression
 *   ext/FiniteDiffBlockBandedMatricesExt.jl:44-68 block-banded decompression
 *
 * Third-party arithmetic not in /root/reference: LinearAlgebra.norm (Julia
 * stdlib; src/jacobians.jl:560,601).  Restated as sqrt(sum of squares) with a
 * plain sequential double accumulation -- differs from BLAS nrm2 at the ulp
 * level only, which moves epsilon by <= ~1e-16 relative.
 *
 * All indices crossing this API are 1-based Int64, as in Julia.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* element type: the same source is compiled twice, -DFDO_F32 gives the Float32 oracle (libfd_oracle32.so) whose
   arithmetic, step rules and eps(T) are those of a Float32 problem in the reference */
#ifdef FDO_F32
typedef float fdo_real;
typedef float complex cplx;
#define FDO_EPS 1.1920928955078125e-07f
#else
typedef double fdo_real;
typedef double complex cplx;
#define FDO_EPS 2.220446049250313e-16
#endif
/* -DFDO_OMP -fopenmp builds libfd_oracle_omp.so: the same passes with their loops split over the host's cores.  The
   reference is single-threaded (no Threads / @simd anywhere in src/), so that build is NOT the reference's speed: it is
   the generous "what a parallel CPU port could reach" upper bound bench.py reports next to the 1-core number
   (SURVEY 8d).  Its masked norm is an OpenMP reduction (different summation order, eps moves by ulps). */
#if defined(FDO_OMP) && defined(_OPENMP)
#include <omp.h>
void fdo_set_num_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int fdo_max_threads(void) { return omp_get_max_threads(); }
#define FDO_PAR _Pragma("omp parallel for schedule(static)")
#define FDO_PAR_SUM(v) _Pragma("omp parallel for schedule(static) reduction(+ : s)")
#else
void fdo_set_num_threads(int n) { (void)n; }
int fdo_max_threads(void) { return 1; }
#define FDO_PAR
#define FDO_PAR_SUM(v)
#endif
typedef void (*fdo_f_real)(void *ctx, fdo_real *fx, const fdo_real *x);
typedef void (*fdo_f_cplx)(void *ctx, cplx *fx, const cplx *x);

enum { FDO_FORWARD = 0, FDO_CENTRAL = 1, FDO_COMPLEX = 2 };

/* sparsity / J kinds */
enum {
    FDO_PAT_NONE = 0,        /* sparsity === nothing: colour index == column index (jacobians.jl:548-557) */
    FDO_PAT_CSC_COMMON = 1,  /* J and sparsity share colptr/rowval: nzval[p] = vfx[rowval[p]] (ext/SparseArrays:38-47) */
    FDO_PAT_CSC_DENSEJ = 2,  /* sparse pattern, dense J: J[row,col] = vfx[row] (ext/SparseArrays:20-28) */
    FDO_PAT_COO_DENSEJ = 3,  /* rows_index/cols_index, dense J (iteration_utils.jl:25-32) */
    FDO_PAT_COO_TRIDIAG = 4, /* rows_index/cols_index, Tridiagonal J (dl,d,du) via setindex! */
    FDO_PAT_BANDED = 5,      /* BandedMatrix data (l+u+1) x N (ext/Banded:13-27) */
    FDO_PAT_BLOCKBANDED = 6, /* BlockBandedMatrix flat data + block_starts/strides (ext/BlockBanded:44-68) */
    FDO_PAT_BANDEDBLOCKBANDED = 7, /* BandedBlockBandedMatrix: per-block banded data (ext/BlockBanded:16-42) */
    FDO_PAT_COO_DENSEJ_ACCUM = 8  /* rows_index/cols_index, dense J, the BROADCAST arm taken when x1 has no fast scalar
                                     indexing (a GPU array): fast_jacobian_setindex!, src/jacobians.jl:574-581,665-671 --
                                     J[r,c] = J[r,c] + (color[c] == color_i) * vfx[r] over ALL listed entries, every colour */
};

typedef struct {
    int kind;
    int64_t M, N;
    /* CSC */
    const int64_t *colptr; /* N+1, 1-based */
    const int64_t *rowval; /* nnz, 1-based */
    /* COO */
    const int64_t *rows_index, *cols_index;
    int64_t ncoo;
    /* banded */
    int64_t l, u;
    /* block banded: nblk square block structure (rows sizes reused for columns,
       as ext/FiniteDiffBlockBandedMatricesExt.jl:47-48 does), block bandwidths
       bl,bu; block_starts is the (bl+bu+1) x nblk band storage of BlockBandedMatrices'
       block_starts BandedMatrix{Int}: block_starts[(bu+K-J) + (bl+bu+1)*(J-1)] = 1-based
       start of block (K,J) in data; block_strides[J-1] = column stride of block-col J */
    int64_t nblk;
    const int64_t *blk_sizes;
    int64_t bl, bu;
    const int64_t *block_starts;
    const int64_t *block_strides;
    /* outputs */
    fdo_real *out0; /* nzval | dense J (col-major M x N) | banded data | block data | d  */
    fdo_real *out1; /* dl (tridiag) */
    fdo_real *out2; /* du (tridiag) */
    int64_t out_len; /* number of stored values in out0 for fill_matrix! */
    /* banded-block-banded: sub-block bandwidths (lambda, mu); block_starts then holds the 1-based start of the
       BANDED data of block (K,J) (the `pointer(bandeddata(view(Jac,K,J)))` of ext/BlockBanded:29-31) and
       block_strides[J-1] its column stride `stride(data,2)` (:32) */
    int64_t lam, mu;
} fdo_pattern;

/* ---- src/epsilons.jl:26-29, 50-53 ---- */
static inline fdo_real eps_forward(fdo_real x, fdo_real relstep, fdo_real absstep, fdo_real dir)
{
    fdo_real a = relstep * fabs(x);
    return (a > absstep ? a : absstep) * dir;
}
static inline fdo_real eps_central(fdo_real x, fdo_real relstep, fdo_real absstep)
{
    fdo_real a = relstep * fabs(x);
    return (a > absstep ? a : absstep);
}

/* src/epsilons.jl:133-144 */
fdo_real fdo_default_relstep(int fdtype)
{
    const fdo_real e = FDO_EPS;
    if (fdtype == FDO_FORWARD) return sqrt(e);
    if (fdtype == FDO_CENTRAL) return cbrt(e);
    return 1.0;
}

/* LinearAlgebra.norm restated (see header) */
static fdo_real norm2(const fdo_real *v, int64_t n)
{
    fdo_real s = 0.0;
    FDO_PAR_SUM(s)
    for (int64_t i = 0; i < n; ++i) s += v[i] * v[i];
    return sqrt(s);
}

static int64_t max_color(const int64_t *colorvec, int64_t n)
{
    int64_t m = colorvec[0];
    for (int64_t i = 1; i < n; ++i)
        if (colorvec[i] > m) m = colorvec[i];
    return m;
}

/* J[row,col] = v for a Tridiagonal (1-based row/col); off-band stores are an
   error in Julia -- never reached for a structurally tridiagonal index list. */
static void tridiag_setindex(const fdo_pattern *p, int64_t row, int64_t col, fdo_real v)
{
    if (row == col) p->out0[row - 1] = v;            /* d  */
    else if (row == col + 1) p->out1[col - 1] = v;   /* dl[col] = J[col+1,col] */
    else if (row + 1 == col) p->out2[row - 1] = v;   /* du[row] = J[row,row+1] */
}

/* fill_matrix!(J,false): src/jacobians.jl:530-532,663 ; ext/SparseArrays:30 */
static void fill_matrix_zero(const fdo_pattern *p)
{
    if (p->kind == FDO_PAT_COO_TRIDIAG) {
        memset(p->out0, 0, sizeof(fdo_real) * (size_t)p->N);
        if (p->N > 1) {
            memset(p->out1, 0, sizeof(fdo_real) * (size_t)(p->N - 1));
            memset(p->out2, 0, sizeof(fdo_real) * (size_t)(p->N - 1));
        }
    } else {
        memset(p->out0, 0, sizeof(fdo_real) * (size_t)p->out_len);
    }
}

/* One colour's decompression of vfx into J. */
static void colored_iteration(const fdo_pattern *p, const fdo_real *vfx, const int64_t *colorvec,
                              int64_t color_i)
{
    const int64_t M = p->M, N = p->N;
    switch (p->kind) {
    case FDO_PAT_CSC_COMMON: /* ext/FiniteDiffSparseArraysExt.jl:38-47 */
        FDO_PAR
        for (int64_t col = 1; col <= N; ++col)
            if (colorvec[col - 1] == color_i)
                for (int64_t sp = p->colptr[col - 1]; sp <= p->colptr[col] - 1; ++sp) {
                    int64_t row = p->rowval[sp - 1];
                    p->out0[sp - 1] = vfx[row - 1];
                }
        break;
    case FDO_PAT_CSC_DENSEJ: /* ext/FiniteDiffSparseArraysExt.jl:20-28 */
        for (int64_t col = 1; col <= N; ++col)
            if (colorvec[col - 1] == color_i)
                for (int64_t sp = p->colptr[col - 1]; sp <= p->colptr[col] - 1; ++sp) {
                    int64_t row = p->rowval[sp - 1];
                    p->out0[(row - 1) + M * (col - 1)] = vfx[row - 1];
                }
        break;
    case FDO_PAT_COO_DENSEJ: /* src/iteration_utils.jl:25-32 */
        for (int64_t i = 0; i < p->ncoo; ++i)
            if (colorvec[p->cols_index[i] - 1] == color_i)
                p->out0[(p->rows_index[i] - 1) + M * (p->cols_index[i] - 1)] = vfx[p->rows_index[i] - 1];
        break;
    case FDO_PAT_COO_DENSEJ_ACCUM: /* src/jacobians.jl:665-671; Bool * Float64 is Julia's "strong zero":
                                      false * v == copysign(0.0, v) even for v = NaN / Inf (base/bool.jl) */
        for (int64_t i = 0; i < p->ncoo; ++i) {
            const fdo_real v = vfx[p->rows_index[i] - 1];
            const fdo_real masked = (colorvec[p->cols_index[i] - 1] == color_i) ? v : copysign((fdo_real)0.0, v);
            fdo_real *dst = &p->out0[(p->rows_index[i] - 1) + M * (p->cols_index[i] - 1)];
            *dst = *dst + masked;
        }
        break;
    case FDO_PAT_COO_TRIDIAG: /* src/iteration_utils.jl:25-32 with Tridiagonal setindex! */
        for (int64_t i = 0; i < p->ncoo; ++i)
            if (colorvec[p->cols_index[i] - 1] == color_i)
                tridiag_setindex(p, p->rows_index[i], p->cols_index[i], vfx[p->rows_index[i] - 1]);
        break;
    case FDO_PAT_BANDED: { /* ext/FiniteDiffBandedMatricesExt.jl:13-27 ; storage per its line 22 */
        const int64_t l = p->l, u = p->u, ld = l + u + 1;
        int64_t c0 = (1 - l > 1) ? 1 - l : 1;         /* max(1,1-l) */
        int64_t c1 = (N + u < N) ? N + u : N;         /* min(ncols,ncols+u) */
        for (int64_t col = c0; col <= c1; ++col)
            if (colorvec[col - 1] == color_i) {
                int64_t r0 = (col - u > 1) ? col - u : 1;
                int64_t r1 = (col + l < M) ? col + l : M;
                for (int64_t row = r0; row <= r1; ++row)
                    p->out0[(u + row - col) + ld * (col - 1)] = vfx[row - 1]; /* data[u+row-col+1, col] */
            }
        break;
    }
    case FDO_PAT_BLOCKBANDED: { /* ext/FiniteDiffBlockBandedMatricesExt.jl:44-68 */
        const int64_t nb = p->nblk;
        int64_t colbase = 0; /* first global column of block-column J, 0-based */
        for (int64_t J = 1; J <= nb; ++J) {
            int64_t ncolsJ = p->blk_sizes[J - 1];
            int64_t K0 = (J - p->bu > 1) ? J - p->bu : 1;   /* blockcolrange */
            int64_t K1 = (J + p->bl < nb) ? J + p->bl : nb;
            for (int64_t j = 1; j <= ncolsJ; ++j) {
                if (colorvec[colbase + j - 1] == color_i) {
                    int64_t rowbase = 0;
                    for (int64_t K = 1; K < K0; ++K) rowbase += p->blk_sizes[K - 1];
                    for (int64_t K = K0; K <= K1; ++K) {
                        int64_t m = p->blk_sizes[K - 1];
                        int64_t start = p->block_starts[(p->bu + K - J) + (p->bl + p->bu + 1) * (J - 1)];
                        int64_t st = p->block_strides[J - 1];
                        for (int64_t k = 1; k <= m; ++k)
                            p->out0[(start - 1) + (j - 1) * st + (k - 1)] = vfx[rowbase + k - 1];
                        rowbase += m;
                    }
                }
            }
            colbase += ncolsJ;
        }
        break;
    }
    case FDO_PAT_BANDEDBLOCKBANDED: { /* ext/FiniteDiffBlockBandedMatricesExt.jl:16-42 */
        const int64_t nb = p->nblk, lam = p->lam, mu = p->mu;
        int64_t colbase = 0;
        for (int64_t J = 1; J <= nb; ++J) {
            int64_t n = p->blk_sizes[J - 1];
            int64_t K0 = (J - p->bu > 1) ? J - p->bu : 1;   /* blockcolrange */
            int64_t K1 = (J + p->bl < nb) ? J + p->bl : nb;
            int64_t rowbase = 0;
            for (int64_t K = 1; K < K0; ++K) rowbase += p->blk_sizes[K - 1];
            for (int64_t K = K0; K <= K1; ++K) {
                int64_t m = p->blk_sizes[K - 1];
                int64_t start = p->block_starts[(p->bu + K - J) + (p->bl + p->bu + 1) * (J - 1)];
                int64_t st = p->block_strides[J - 1];
                for (int64_t j = 1; j <= n; ++j)
                    if (colorvec[colbase + j - 1] == color_i) {
                        int64_t k0 = (j - mu > 1) ? j - mu : 1, k1 = (j + lam < m) ? j + lam : m;
                        for (int64_t k = k0; k <= k1; ++k)   /* unsafe_store!(p, b_v[k], (j-1)*st + mu + k - j + 1) */
                            p->out0[(start - 1) + (j - 1) * st + mu + k - j] = vfx[rowbase + k - 1];
                    }
                rowbase += m;
            }
            colbase += n;
        }
        break;
    }
    default: break;
    }
}

/*
 * Cached in-place finite_difference_jacobian!  (src/jacobians.jl:504-653).
 *
 * x is mutable: the central arm perturbs and restores it in place exactly as
 * the reference does (:604,:620).  x1,x2,fx,fx1 are the cache arrays (caller
 * allocated, may hold garbage -- test/cache_reuse_tests.jl).  For the complex
 * arm cx1/cfx are the complex cache arrays (x1, fx) and fx1 is unused.
 * f_in may be NULL.  fcalls (optional) counts f! evaluations.
 * Returns 0, or 1 for an unsupported fdtype.
 */
int fdo_jacobian_cached(int fdtype, fdo_f_real f, fdo_f_cplx fc, void *ctx, fdo_real *x, fdo_real *x1,
                        fdo_real *x2, fdo_real *fx, fdo_real *fx1, cplx *cx1, cplx *cfx,
                        const fdo_real *f_in, const int64_t *colorvec, fdo_real relstep,
                        fdo_real absstep, fdo_real dir, const fdo_pattern *pat, int64_t *fcalls)
{
    const int64_t M = pat->M, N = pat->N;
    int64_t nf = 0;
    const int has_sparsity = pat->kind != FDO_PAT_NONE;

    if (fdtype == FDO_COMPLEX) {
        for (int64_t i = 0; i < N; ++i) cx1[i] = x[i]; /* copyto!(x1,x)  :519 */
    } else {
        memcpy(x1, x, sizeof(fdo_real) * (size_t)N);     /* :519 */
    }
    if (has_sparsity) fill_matrix_zero(pat);           /* :530-532 */

    const int64_t ncolors = max_color(colorvec, N);    /* 1:maximum(colorvec) */

    if (fdtype == FDO_FORWARD) {
        const fdo_real *vfx;
        if (f_in == NULL) { f(ctx, fx, x); ++nf; vfx = fx; } /* :540-545 */
        else vfx = f_in;
        for (int64_t color_i = 1; color_i <= ncolors; ++color_i) {
            if (!has_sparsity) { /* :548-557 */
                fdo_real x1_save = x1[color_i - 1];
                fdo_real epsilon = eps_forward(x1_save, relstep, absstep, dir);
                x1[color_i - 1] = x1_save + epsilon;
                f(ctx, fx1, x1); ++nf;
                for (int64_t r = 0; r < M; ++r)
                    pat->out0[r + M * (color_i - 1)] = (fx1[r] - vfx[r]) / epsilon;
                x1[color_i - 1] = x1_save;
            } else { /* :558-585 */
                FDO_PAR
                for (int64_t i = 0; i < N; ++i) x2[i] = x1[i] * (fdo_real)(colorvec[i] == color_i);
                fdo_real tmp = norm2(x2, N);
                fdo_real epsilon = eps_forward(sqrt(tmp), relstep, absstep, dir);
                FDO_PAR
                for (int64_t i = 0; i < N; ++i) x1[i] = x1[i] + epsilon * (fdo_real)(colorvec[i] == color_i);
                f(ctx, fx1, x1); ++nf;
                FDO_PAR
                for (int64_t r = 0; r < M; ++r) fx1[r] = (fx1[r] - vfx[r]) / epsilon;
                colored_iteration(pat, fx1, colorvec, color_i);
                FDO_PAR
                for (int64_t i = 0; i < N; ++i) x1[i] = x1[i] - epsilon * (fdo_real)(colorvec[i] == color_i);
            }
        }
    } else if (fdtype == FDO_CENTRAL) {
        for (int64_t color_i = 1; color_i <= ncolors; ++color_i) {
            if (!has_sparsity) { /* :590-598 */
                fdo_real x_save = x[color_i - 1];
                fdo_real epsilon = eps_central(x_save, relstep, absstep);
                x1[color_i - 1] = x_save + epsilon;
                f(ctx, fx1, x1); ++nf;
                x1[color_i - 1] = x_save - epsilon;
                f(ctx, fx, x1); ++nf;
                for (int64_t r = 0; r < M; ++r)
                    pat->out0[r + M * (color_i - 1)] = (fx1[r] - fx[r]) / (2 * epsilon);
                x1[color_i - 1] = x_save;
            } else { /* :599-621 */
                FDO_PAR
                for (int64_t i = 0; i < N; ++i) x2[i] = x1[i] * (fdo_real)(colorvec[i] == color_i);
                fdo_real tmp = norm2(x2, N);
                fdo_real epsilon = eps_central(sqrt(tmp), relstep, absstep);
                FDO_PAR
                for (int64_t i = 0; i < N; ++i) x1[i] = x1[i] + epsilon * (fdo_real)(colorvec[i] == color_i);
                FDO_PAR
                for (int64_t i = 0; i < N; ++i) x[i] = x[i] - epsilon * (fdo_real)(colorvec[i] == color_i);
                f(ctx, fx1, x1); ++nf;
                f(ctx, fx, x); ++nf;
                FDO_PAR
                for (int64_t r = 0; r < M; ++r) fx1[r] = (fx1[r] - fx[r]) / (2 * epsilon);
                colored_iteration(pat, fx1, colorvec, color_i);
                FDO_PAR
                for (int64_t i = 0; i < N; ++i) x1[i] = x1[i] - epsilon * (fdo_real)(colorvec[i] == color_i);
                FDO_PAR
                for (int64_t i = 0; i < N; ++i) x[i] = x[i] + epsilon * (fdo_real)(colorvec[i] == color_i);
            }
        }
    } else if (fdtype == FDO_COMPLEX) { /* :623-648 */
        const fdo_real epsilon = FDO_EPS; /* eps(T), src/epsilons.jl:104-107 */
        fdo_real *vre = (fdo_real *)malloc(sizeof(fdo_real) * (size_t)(M > 0 ? M : 1));
        for (int64_t color_i = 1; color_i <= ncolors; ++color_i) {
            if (!has_sparsity) {
                cplx x1_save = cx1[color_i - 1];
                cx1[color_i - 1] = x1_save + I * epsilon;
                fc(ctx, cfx, cx1); ++nf;
                for (int64_t r = 0; r < M; ++r) pat->out0[r + M * (color_i - 1)] = cimag(cfx[r]) / epsilon;
                cx1[color_i - 1] = x1_save;
            } else {
                for (int64_t i = 0; i < N; ++i)
                    cx1[i] = cx1[i] + CMPLX(0.0, epsilon * (fdo_real)(colorvec[i] == color_i));
                fc(ctx, cfx, cx1); ++nf;
                for (int64_t r = 0; r < M; ++r) { cfx[r] = cimag(cfx[r]) / epsilon; vre[r] = creal(cfx[r]); }
                colored_iteration(pat, vre, colorvec, color_i);
                for (int64_t i = 0; i < N; ++i)
                    cx1[i] = cx1[i] - CMPLX(0.0, epsilon * (fdo_real)(colorvec[i] == color_i));
            }
        }
        free(vre);
    } else {
        return 1; /* fdtype_error, src/epsilons.jl:159-167 */
    }
    if (fcalls) *fcalls = nf;
    return 0;
}

/* finite_difference_jvp!  (src/jvp.jl:238-274).  x1, fx1 are the cache arrays; jvp is the output.
   Returns 0, or 1 for :complex (rejected by the reference, :248-250 / :270-271). */
int fdo_jvp(int fdtype, fdo_f_real f, void *ctx, const fdo_real *x, const fdo_real *v, int64_t M, int64_t N,
            const fdo_real *f_in, fdo_real relstep, fdo_real absstep, fdo_real dir, fdo_real *x1, fdo_real *fx1, fdo_real *jvp,
            fdo_real *eps_out)
{
    if (fdtype != FDO_FORWARD && fdtype != FDO_CENTRAL) return 1;
    fdo_real dot = 0.0;
    for (int64_t i = 0; i < N; ++i) dot += x[i] * v[i];
    const fdo_real tmp = sqrt(fabs(dot));                         /* :253 */
    fdo_real epsilon;
    if (fdtype == FDO_FORWARD) {
        epsilon = eps_forward(tmp, relstep, absstep, dir);      /* :254 */
        const fdo_real *b;
        if (f_in == NULL) { f(ctx, fx1, x); b = fx1; } else b = f_in;
        for (int64_t i = 0; i < N; ++i) x1[i] = x[i] + epsilon * v[i];
        f(ctx, jvp, x1);
        for (int64_t r = 0; r < M; ++r) jvp[r] = (jvp[r] - b[r]) / epsilon;
    } else {
        epsilon = eps_central(tmp, relstep, absstep);
        for (int64_t i = 0; i < N; ++i) x1[i] = x[i] - epsilon * v[i];
        f(ctx, fx1, x1);
        for (int64_t i = 0; i < N; ++i) x1[i] = x[i] + epsilon * v[i];
        f(ctx, jvp, x1);
        for (int64_t r = 0; r < M; ++r) jvp[r] = (jvp[r] - fx1[r]) / (2 * epsilon);
    }
    if (eps_out) *eps_out = epsilon;
    return 0;
}

/* src/jacobians.jl:473-488: column-major scan of a dense pattern matrix.
   rows/cols must hold count(A != 0) entries; returns that count. */
int64_t fdo_findstructralnz_dense(const fdo_real *A, int64_t m, int64_t n, int64_t *rows, int64_t *cols)
{
    int64_t idx = 0;
    for (int64_t j = 1; j <= n; ++j)
        for (int64_t i = 1; i <= m; ++i)
            if (A[(i - 1) + m * (j - 1)] != 0) {
                if (rows) { rows[idx] = i; cols[idx] = j; }
                ++idx;
            }
    return idx;
}

/* Out-of-place dense forward Jacobian (config 1 plumbing), src/jacobians.jl:319-331:
   per-element epsilon, J[:,i] = (f(x with x[i]+eps) - f(x)) / eps.  J is M x N col-major. */
void fdo_jacobian_oop_dense_forward(fdo_f_real f, void *ctx, const fdo_real *x, int64_t M, int64_t N,
                                    fdo_real relstep, fdo_real absstep, fdo_real dir, fdo_real *J)
{
    fdo_real *vecfx = (fdo_real *)malloc(sizeof(fdo_real) * (size_t)M);
    fdo_real *fx1 = (fdo_real *)malloc(sizeof(fdo_real) * (size_t)M);
    fdo_real *x1 = (fdo_real *)malloc(sizeof(fdo_real) * (size_t)N);
    f(ctx, vecfx, x);
    for (int64_t i = 0; i < N; ++i) {
        fdo_real x_save = x[i];
        fdo_real epsilon = eps_forward(x_save, relstep, absstep, dir);
        /* setindex(vecx, x_save+epsilon, i): x .* (i .!== 1:n) .+ v .* (i .== 1:n)  (src/FiniteDiff.jl:99-102) */
        for (int64_t k = 0; k < N; ++k) x1[k] = x[k] * (fdo_real)(k != i) + (x_save + epsilon) * (fdo_real)(k == i);
        f(ctx, fx1, x1);
        for (int64_t r = 0; r < M; ++r) J[r + M * i] = (fx1[r] - vecfx[r]) / epsilon;
    }
    free(vecfx); free(fx1); free(x1);
}

/* ===================================================================== */
/* Fixture functions (the reference's test f!s and the benchmark families) */
/* ===================================================================== */

/* test/coloring_tests.jl:5-13 : second difference, zero Dirichlet ends */
void fdo_f_tridiag(void *ctx, fdo_real *dx, const fdo_real *x)
{
    int64_t n = *(const int64_t *)ctx;
    if (n == 1) { dx[0] = -2 * x[0]; return; }
    FDO_PAR
    for (int64_t i = 1; i < n - 1; ++i) dx[i] = x[i - 1] - 2 * x[i] + x[i + 1];
    dx[0] = -2 * x[0] + x[1];
    dx[n - 1] = x[n - 2] - 2 * x[n - 1];
}
void fdo_fc_tridiag(void *ctx, cplx *dx, const cplx *x)
{
    int64_t n = *(const int64_t *)ctx;
    if (n == 1) { dx[0] = -2 * x[0]; return; }
    for (int64_t i = 1; i < n - 1; ++i) dx[i] = x[i - 1] - 2 * x[i] + x[i + 1];
    dx[0] = -2 * x[0] + x[1];
    dx[n - 1] = x[n - 2] - 2 * x[n - 1];
}

/* nonlinear tridiagonal variant: dx[i] = x[i-1] - 2x[i] + x[i+1] + x[i]^2 * x[i+1]
   (x[n] treated as 0 beyond the end) so that J depends on x. */
void fdo_f_tridiag_nl(void *ctx, fdo_real *dx, const fdo_real *x)
{
    int64_t n = *(const int64_t *)ctx;
    for (int64_t i = 0; i < n; ++i) {
        fdo_real xm = i > 0 ? x[i - 1] : 0.0, xp = i + 1 < n ? x[i + 1] : 0.0;
        dx[i] = xm - 2 * x[i] + xp + x[i] * x[i] * xp;
    }
}
void fdo_fc_tridiag_nl(void *ctx, cplx *dx, const cplx *x)
{
    int64_t n = *(const int64_t *)ctx;
    for (int64_t i = 0; i < n; ++i) {
        cplx xm = i > 0 ? x[i - 1] : 0.0, xp = i + 1 < n ? x[i + 1] : 0.0;
        dx[i] = xm - 2 * x[i] + xp + x[i] * x[i] * xp;
    }
}

/* 2-D 5-point stencils on an nx (fast index) x ny grid; ctx = {nx, ny} */
/* zero-Dirichlet Laplacian */
void fdo_f_lap5(void *ctx, fdo_real *out, const fdo_real *x)
{
    const int64_t nx = ((const int64_t *)ctx)[0], ny = ((const int64_t *)ctx)[1];
    for (int64_t j = 0; j < ny; ++j)
        for (int64_t i = 0; i < nx; ++i) {
            int64_t k = i + nx * j;
            fdo_real w = i > 0 ? x[k - 1] : 0.0, e = i + 1 < nx ? x[k + 1] : 0.0;
            fdo_real s = j > 0 ? x[k - nx] : 0.0, n = j + 1 < ny ? x[k + nx] : 0.0;
            out[k] = w + e + s + n - 4 * x[k];
        }
}
void fdo_fc_lap5(void *ctx, cplx *out, const cplx *x)
{
    const int64_t nx = ((const int64_t *)ctx)[0], ny = ((const int64_t *)ctx)[1];
    for (int64_t j = 0; j < ny; ++j)
        for (int64_t i = 0; i < nx; ++i) {
            int64_t k = i + nx * j;
            cplx w = i > 0 ? x[k - 1] : 0.0, e = i + 1 < nx ? x[k + 1] : 0.0;
            cplx s = j > 0 ? x[k - nx] : 0.0, n = j + 1 < ny ? x[k + nx] : 0.0;
            out[k] = w + e + s + n - 4 * x[k];
        }
}
/* nonlinear variant of the zero-Dirichlet Laplacian: + x[k]^2 * x[k+1] (east neighbour, 0 beyond the edge); same
   5-point pattern, J depends on x -- the full-size parity fixture of BASELINE config 3 */
void fdo_f_lap5_nl(void *ctx, fdo_real *out, const fdo_real *x)
{
    const int64_t nx = ((const int64_t *)ctx)[0], ny = ((const int64_t *)ctx)[1];
    for (int64_t j = 0; j < ny; ++j)
        for (int64_t i = 0; i < nx; ++i) {
            int64_t k = i + nx * j;
            fdo_real w = i > 0 ? x[k - 1] : 0.0, e = i + 1 < nx ? x[k + 1] : 0.0;
            fdo_real s = j > 0 ? x[k - nx] : 0.0, n = j + 1 < ny ? x[k + nx] : 0.0;
            out[k] = (w + e + s + n - 4 * x[k]) + x[k] * x[k] * e;
        }
}
void fdo_fc_lap5_nl(void *ctx, cplx *out, const cplx *x)
{
    const int64_t nx = ((const int64_t *)ctx)[0], ny = ((const int64_t *)ctx)[1];
    for (int64_t j = 0; j < ny; ++j)
        for (int64_t i = 0; i < nx; ++i) {
            int64_t k = i + nx * j;
            cplx w = i > 0 ? x[k - 1] : 0.0, e = i + 1 < nx ? x[k + 1] : 0.0;
            cplx s = j > 0 ? x[k - nx] : 0.0, n = j + 1 < ny ? x[k + nx] : 0.0;
            out[k] = (w + e + s + n - 4 * x[k]) + x[k] * x[k] * e;
        }
}
/* clamped-edge sum stencil, test/coloring_tests.jl:99-108 */
void fdo_f_clamp5(void *ctx, fdo_real *out, const fdo_real *x)
{
    const int64_t nx = ((const int64_t *)ctx)[0], ny = ((const int64_t *)ctx)[1];
    for (int64_t j = 0; j < ny; ++j)
        for (int64_t i = 0; i < nx; ++i) {
            int64_t im = i > 0 ? i - 1 : 0, ip = i + 1 < nx ? i + 1 : nx - 1;
            int64_t jm = j > 0 ? j - 1 : 0, jp = j + 1 < ny ? j + 1 : ny - 1;
            out[i + nx * j] = x[i + nx * j] + x[im + nx * j] + x[ip + nx * j] + x[i + nx * jm] + x[i + nx * jp];
        }
}
void fdo_fc_clamp5(void *ctx, cplx *out, const cplx *x)
{
    const int64_t nx = ((const int64_t *)ctx)[0], ny = ((const int64_t *)ctx)[1];
    for (int64_t j = 0; j < ny; ++j)
        for (int64_t i = 0; i < nx; ++i) {
            int64_t im = i > 0 ? i - 1 : 0, ip = i + 1 < nx ? i + 1 : nx - 1;
            int64_t jm = j > 0 ? j - 1 : 0, jp = j + 1 < ny ? j + 1 : ny - 1;
            out[i + nx * j] = x[i + nx * j] + x[im + nx * j] + x[ip + nx * j] + x[i + nx * jm] + x[i + nx * jp];
        }
}

/* block-coupled dense-block f; ctx = {nblk, bs}:
   f_b[k] = x_b[k]*(sig_{b-1}+sig_b+sig_{b+1}) + sin(x_b[k]),  sig_b = sum_j w_j x_b[j], w_j=(j+1)/bs (j 0-based) */
void fdo_f_blockcoupled(void *ctx, fdo_real *out, const fdo_real *x)
{
    const int64_t nb = ((const int64_t *)ctx)[0], bs = ((const int64_t *)ctx)[1];
    fdo_real *sig = (fdo_real *)malloc(sizeof(fdo_real) * (size_t)nb);
    for (int64_t b = 0; b < nb; ++b) {
        fdo_real s = 0.0;
        for (int64_t j = 0; j < bs; ++j) s += ((fdo_real)(j + 1) / (fdo_real)bs) * x[b * bs + j];
        sig[b] = s;
    }
    for (int64_t b = 0; b < nb; ++b) {
        fdo_real sm = b > 0 ? sig[b - 1] : 0.0, sp = b + 1 < nb ? sig[b + 1] : 0.0;
        fdo_real S = sm + sig[b] + sp;
        for (int64_t k = 0; k < bs; ++k) out[b * bs + k] = x[b * bs + k] * S + sin(x[b * bs + k]);
    }
    free(sig);
}
void fdo_fc_blockcoupled(void *ctx, cplx *out, const cplx *x)
{
    const int64_t nb = ((const int64_t *)ctx)[0], bs = ((const int64_t *)ctx)[1];
    cplx *sig = (cplx *)malloc(sizeof(cplx) * (size_t)nb);
    for (int64_t b = 0; b < nb; ++b) {
        cplx s = 0.0;
        for (int64_t j = 0; j < bs; ++j) s += ((fdo_real)(j + 1) / (fdo_real)bs) * x[b * bs + j];
        sig[b] = s;
    }
    for (int64_t b = 0; b < nb; ++b) {
        cplx sm = b > 0 ? sig[b - 1] : 0.0, sp = b + 1 < nb ? sig[b + 1] : 0.0;
        cplx S = sm + sig[b] + sp;
        for (int64_t k = 0; k < bs; ++k) out[b * bs + k] = x[b * bs + k] * S + csin(x[b * bs + k]);
    }
    free(sig);
}

/* test/coloring_tests.jl:124-133 : y = (x1-3)^2 + x1*x2 + (x2+4)^2 - 3 ; ctx = n (length y) */
void fdo_f_nonsquare(void *ctx, fdo_real *y, const fdo_real *x)
{
    int64_t n = *(const int64_t *)ctx;
    for (int64_t k = 0; k < n; ++k) {
        fdo_real a = x[k], b = x[n + k];
        y[k] = (a - 3) * (a - 3) + a * b + (b + 4) * (b + 4) - 3;
    }
}
void fdo_fc_nonsquare(void *ctx, cplx *y, const cplx *x)
{
    int64_t n = *(const int64_t *)ctx;
    for (int64_t k = 0; k < n; ++k) {
        cplx a = x[k], b = x[n + k];
        y[k] = (a - 3) * (a - 3) + a * b + (b + 4) * (b + 4) - 3;
    }
}

/* config 1: f(x) = sin.(x) ; ctx = n */
void fdo_f_sin(void *ctx, fdo_real *y, const fdo_real *x)
{
    int64_t n = *(const int64_t *)ctx;
    for (int64_t k = 0; k < n; ++k) y[k] = sin(x[k]);
}

/* ------------------------------------------------------------------ */
/* Convenience for the cpu_baseline leg of bench.py: tridiagonal CSC    */
/* pattern builder (1-based Int64, as Julia's SparseMatrixCSC stores).   */
/* ------------------------------------------------------------------ */
void fdo_build_tridiag_csc(int64_t n, int64_t *colptr, int64_t *rowval)
{
    int64_t p = 1;
    for (int64_t j = 1; j <= n; ++j) {
        colptr[j - 1] = p;
        if (j > 1) rowval[p++ - 1] = j - 1;
        rowval[p++ - 1] = j;
        if (j < n) rowval[p++ - 1] = j + 1;
    }
    colptr[n] = p;
}
