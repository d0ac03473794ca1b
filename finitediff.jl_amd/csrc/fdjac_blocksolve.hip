// The consumer for block-banded Jacobians: (alpha I + beta J) y = b for a BLOCK-TRIDIAGONAL J of nblk dense
// b x b blocks (b <= 32) on the storage fd_plan_create_blockbanded fills -- BlockBandedMatrix data with block bandwidths (1, 1) and
// uniform block sizes (BlockSkylineSizes: block column J's in-band blocks stacked into one column-major panel, panels one after the
// other) -- the implicit step behind BASELINE's config 5 (10^4 blocks of 32 x 32).
//
// Method: block cyclic reduction without pivoting, ONE WORKGROUP per kept block row.  The workgroup brings a neighbour's row
// [B | A | C | d] into LDS, reduces it to [I | B^-1 A | B^-1 C | B^-1 d] by Gauss-Jordan (one barrier per pivot: the pivot row is
// left unscaled until the end, so a step reads row p and column p and writes neither), multiplies by its own coupling block and
// accumulates B', d' in LDS; A', C' go straight to the next level (rows of 3 b^2 + b doubles, column-major blocks, one contiguous piece per row).  The
// back-substitution solves an eliminated row from its two known neighbours the same way.  Level 0 is read from the caller's data and
// b; all arithmetic in Float64.  Rows of alpha I + beta J without diagonal dominance: the solve refuses (NaN, status bit 0) unless
// the caller vouches for the matrix -- the policy of the other two consumers.
#include "fdjac_internal.h"
#include <cstring>
#include <new>

namespace fdjac {

constexpr int kBtdMaxB = 32;
constexpr int kBtdPitch = 3 * kBtdMaxB + 1;      // [B | A | C | d]: 97 doubles per row of the augmented matrix (odd: column walks hit every bank)
constexpr int kBtdMaxLevels = 40;
constexpr int kBtdThreads = 256;

struct BtdLevels {
    int nlev;
    long long n[kBtdMaxLevels];
    long long off[kBtdMaxLevels];     // rows of level l >= 1 in `pool` (doubles)
    long long xoff[kBtdMaxLevels];    // solution of level l >= 1 in `xpool`
};
struct BtdSrc {
    const real_t *data, *rhs;
    long long nblk;
    int b;
    double alpha, beta;
    int *status;
};

// block (K, J) of level 0, element (r, c): where it lies in BlockBandedMatrix data (uniform blocks, block bandwidths (1, 1))
__device__ __forceinline__ long long btd_pos(const BtdSrc &s, long long K, long long J, int r, int c)
{
    const long long b = s.b, nb = s.nblk;
    const long long K0 = J > 0 ? J - 1 : 0, K1 = J + 1 < nb ? J + 1 : nb - 1;
    const long long stride = (K1 - K0 + 1) * b;
    const long long pstart = J == 0 ? 0 : 2 * b * b + (J - 1) * 3 * b * b;      // (panel 0 holds two blocks -- or one, then there is no panel 1)
    return pstart + (K - K0) * b + (long long)c * stride + r;
}

// block `which` (0: A = (row, row-1), 1: B = (row, row), 2: C = (row, row+1)) of block row `row` of level l: this thread's (up to) four
// elements e = t, t + 256, ... of the column-major block into registers -- ALL loads of a phase are issued before the first value is
// used (a load inside the loop that stores it to LDS is waited for trip by trip) -- and from there into LDS: dst[r * pitch + c]
struct BtdFrag { double v[4]; };
__device__ __forceinline__ BtdFrag btd_fetch_block(const BtdSrc &s, const double *pool, const BtdLevels &lv, int l, long long row, int which)
{
    const int b = s.b, t = threadIdx.x;
    BtdFrag f;
    if (l == 0) {
        const long long J = row + which - 1;
        const bool exists = J >= 0 && J < s.nblk;
        const long long Jc = exists ? J : row;
        real_t raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = t + u * kBtdThreads, ec = e < b * b ? e : 0;
            const int c = ec / b, r = ec - c * b;                 // (column-major in the source: lane-consecutive addresses)
            raw[u] = s.data[btd_pos(s, row, Jc, r, c)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = t + u * kBtdThreads, c = e / b, r = e - c * b;
            f.v[u] = exists ? s.beta * (double)raw[u] + ((which == 1 && r == c) ? s.alpha : 0.0) : 0.0;
        }
    } else {
        const double *p = pool + lv.off[l] + row * (long long)(3 * b * b + b) + (long long)which * b * b;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = t + u * kBtdThreads; f.v[u] = p[e < b * b ? e : 0]; }
    }
    return f;
}
__device__ __forceinline__ void btd_put_block(const BtdFrag &f, int b, double *dst, int pitch)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = threadIdx.x + u * kBtdThreads;
        if (e < b * b) { const int c = e / b, r = e - c * b; dst[r * pitch + c] = f.v[u]; }      // (lanes walk down a column: with the odd pitch every lane hits its own bank)
    }
}
__device__ __forceinline__ void btd_load_block(const BtdSrc &s, const double *pool, const BtdLevels &lv, int l, long long row, int which,
                                               double *dst, int pitch)
{
    const BtdFrag f = btd_fetch_block(s, pool, lv, l, row, which);
    btd_put_block(f, s.b, dst, pitch);
}
__device__ __forceinline__ void btd_load_rhs(const BtdSrc &s, const double *pool, const BtdLevels &lv, int l, long long row, double *dst, int pitch)
{
    const int b = s.b, t = threadIdx.x;
    if (t < b) dst[t * pitch] = l == 0 ? (double)s.rhs[row * b + t] : pool[lv.off[l] + row * (long long)(3 * b * b + b) + 3LL * b * b + t];
}

// Gauss-Jordan without pivoting on M = [B | R] (b rows, W = b + nr columns, row pitch kBtdPitch): afterwards the R part holds B^-1 R.
// The matrix lives in REGISTERS while it is reduced -- thread (q, g) = (t & 31, t >> 5) keeps row q's columns g, g + 8, ... (NE of
// them) -- and a pivot step exchanges only what it must through LDS: the pivot row (its owners' NE values each) and column p (one value
// per row), double-buffered so that ONE barrier per pivot suffices; every thread then updates its NE elements from NE + 2 independent
// LDS reads (profiles/r06_blocksolve.md).  The pivot row stays unscaled until the end.  s_x: 2 x (32 + 8 NE) doubles.  Every thread calls.
template <int NE>
__device__ __forceinline__ void btd_gauss_jordan(double *M, double *s_x, int b, int nr)
{
    const int t = threadIdx.x, q = t & 31, g = t >> 5;
    const int W = b + nr;
    constexpr int kBuf = 32 + 8 * NE;
    double m[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) { const int c = g + 8 * j; m[j] = (q < b && c < W) ? M[q * kBtdPitch + c] : 0.0; }
    double inv_q = 0.0;                      // 1 / pivot of this thread's row, met at step p == q
    // (the pivot loop is unrolled completely: p >> 3 -- which of a thread's elements lies in column p -- and the comparisons
    //  "column > p" are then compile-time, the select chains and half of the updates disappear)
#pragma unroll
    for (int p = 0; p < kBtdMaxB; ++p) {
        if (p < b) {
            double *fcol = s_x + (p & 1) * kBuf, *prow = fcol + 32;
            if (g == (p & 7)) fcol[q] = m[p >> 3];            // this column group owns column p
            if (q == p) {
#pragma unroll
                for (int j = p >> 3; j < NE; ++j) prow[g + 8 * j] = m[j];
            }
            __syncthreads();
            // 1 / pivot: the hardware's reciprocal and two Newton steps (full Float64 accuracy; the IEEE division sequence is three
            // times as long and sits on the critical path of every pivot step)
            const double pv = prow[p];
            double inv = __builtin_amdgcn_rcp(pv);
            inv = __builtin_fma(__builtin_fma(-pv, inv, 1.0), inv, inv);
            inv = __builtin_fma(__builtin_fma(-pv, inv, 1.0), inv, inv);
            if (q == p) inv_q = inv;
            const double f = fcol[q] * inv;
            double pr[NE];
#pragma unroll
            for (int j = p >> 3; j < NE; ++j) pr[j] = prow[g + 8 * j];
            if (q != p) {
#pragma unroll
                for (int j = p >> 3; j < NE; ++j) m[j] = (g + 8 * j > p) ? m[j] - f * pr[j] : m[j];
            }
        }
    }
    // the reduced right-hand sides back into LDS, scaled by the row's pivot
#pragma unroll
    for (int j = 0; j < NE; ++j) { const int c = g + 8 * j; if (q < b && c >= b && c < W) M[q * kBtdPitch + c] = m[j] * inv_q; }
    __syncthreads();
}

// one reduction step: kept row m of level l + 1 from rows 2m, 2m + 1, 2m + 2 of level l
__global__ void __launch_bounds__(kBtdThreads) k_btd_reduce(BtdSrc src, double *pool, BtdLevels lv, int l)
{
    __shared__ double s_M[kBtdMaxB * kBtdPitch];                  // a neighbour's [B | A | C | d]
    __shared__ double s_P[kBtdMaxB * (kBtdMaxB + 1)];             // this row's coupling block to that neighbour
    __shared__ double s_B[kBtdMaxB * (kBtdMaxB + 1)];             // B' (accumulated)
    __shared__ double s_d[kBtdMaxB], s_x[2 * (32 + 8 * 13)];
    const int b = src.b, t = threadIdx.x, r = t & 31, c8 = t >> 5;
    const long long m = blockIdx.x, i = 2 * m + 1, n = lv.n[l];
    double *out = pool + lv.off[l + 1] + m * (long long)(3 * b * b + b);
    btd_load_block(src, pool, lv, l, i, 1, s_B, kBtdMaxB + 1);
    if (t < b) s_d[t] = l == 0 ? (double)src.rhs[i * b + t] : pool[lv.off[l] + i * (long long)(3 * b * b + b) + 3LL * b * b + t];
    for (int side = 0; side < 2; ++side) {
        const long long nb_row = side == 0 ? i - 1 : i + 1;
        if (nb_row >= n) {       // no upper neighbour: C' = 0
            for (int e = t; e < b * b; e += kBtdThreads) out[2LL * b * b + e] = 0.0;
            break;
        }
        {   // (sixteen loads in flight per thread, then the LDS writes)
            const BtdFrag fB = btd_fetch_block(src, pool, lv, l, nb_row, 1), fA = btd_fetch_block(src, pool, lv, l, nb_row, 0),
                          fC = btd_fetch_block(src, pool, lv, l, nb_row, 2), fP = btd_fetch_block(src, pool, lv, l, i, side == 0 ? 0 : 2);
            btd_load_rhs(src, pool, lv, l, nb_row, s_M + 3 * b, kBtdPitch);
            btd_put_block(fB, b, s_M, kBtdPitch);
            btd_put_block(fA, b, s_M + b, kBtdPitch);
            btd_put_block(fC, b, s_M + 2 * b, kBtdPitch);
            btd_put_block(fP, b, s_P, kBtdMaxB + 1);
        }
        __syncthreads();
        btd_gauss_jordan<13>(s_M, s_x, b, 2 * b + 1);
        // Z_A = s_M[:, b .. 2b), Z_C = s_M[:, 2b .. 3b), Z_d = s_M[:, 3b]
        if (b == 32) {
            // 32 x 32 blocks: the two 32^3 products on the Float64 matrix cores -- v_mfma_f64_16x16x4, one 16 x 16 tile of both
            // products per wavefront, 16 instructions (A: lane l holds P[i = l & 15][k = l >> 4], B: Z[k = l >> 4][j = l & 15], D: column
            // l & 15, rows (l >> 4) + 4 reg).  The products are staged in s_P (free once every wavefront has read it) and leave as
            // dense column-major pieces.
            typedef double d4_t __attribute__((ext_vector_type(4)));
            const int lane = t & 63, w = t >> 6, ti = w >> 1, tj = w & 1, li = lane & 15, lk = lane >> 4;
            d4_t accA = {0, 0, 0, 0}, accC = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const double a = s_P[(16 * ti + li) * (kBtdMaxB + 1) + 4 * ks + lk];
                const double zA = s_M[(4 * ks + lk) * kBtdPitch + 32 + 16 * tj + li], zC = s_M[(4 * ks + lk) * kBtdPitch + 64 + 16 * tj + li];
                accA = __builtin_amdgcn_mfma_f64_16x16x4f64(a, zA, accA, 0, 0, 0);
                accC = __builtin_amdgcn_mfma_f64_16x16x4f64(a, zC, accC, 0, 0, 0);
            }
            if (c8 == 0) {       // (the matrix-vector product P Z_d stays on the vector unit)
                const double *Pr = s_P + r * (kBtdMaxB + 1);
                double sd = 0.0;
                for (int k = 0; k < 32; ++k) sd += Pr[k] * s_M[k * kBtdPitch + 96];
                s_d[r] -= sd;
            }
            __syncthreads();     // every wavefront has read s_P
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rr = 16 * ti + lk + 4 * reg, cc = 16 * tj + li;
                const double keep = side == 0 ? accA[reg] : accC[reg], acc = side == 0 ? accC[reg] : accA[reg];
                s_P[rr * (kBtdMaxB + 1) + cc] = -keep;                  // A' = -A Z_A  /  C' = -C Z_C
                s_B[rr * (kBtdMaxB + 1) + cc] -= acc;                   // B' -= A Z_C  /  B' -= C Z_A
            }
            __syncthreads();
            double *o = out + (side == 0 ? 0 : 2LL * b * b);
            for (int e = t; e < 32 * 32; e += kBtdThreads) o[e] = s_P[(e & 31) * (kBtdMaxB + 1) + (e >> 5)];
        } else if (r < b) {
            const double *Pr = s_P + r * (kBtdMaxB + 1);
            for (int c = c8; c < b; c += 8) {
                double sa = 0.0, sc = 0.0;
                for (int k = 0; k < b; ++k) { const double pk = Pr[k]; sa += pk * s_M[k * kBtdPitch + b + c]; sc += pk * s_M[k * kBtdPitch + 2 * b + c]; }
                if (side == 0) { out[c * b + r] = -sa; s_B[r * (kBtdMaxB + 1) + c] -= sc; }             // A' = -A Z_A;  B' -= A Z_C
                else { out[2LL * b * b + c * b + r] = -sc; s_B[r * (kBtdMaxB + 1) + c] -= sa; }         // C' = -C Z_C;  B' -= C Z_A
            }
            if (c8 == 0) {
                double sd = 0.0;
                for (int k = 0; k < b; ++k) sd += Pr[k] * s_M[k * kBtdPitch + 3 * b];
                s_d[r] -= sd;
            }
        }
        __syncthreads();
    }
    for (int e = t; e < b * b; e += kBtdThreads) { const int cc = e / b, rr = e - cc * b; out[(long long)b * b + e] = s_B[rr * (kBtdMaxB + 1) + cc]; }
    if (t < b) out[3LL * b * b + t] = s_d[t];
}

// back-substitution: the eliminated row 2m of level l from its neighbours' solution (level l + 1); copies the kept row 2m + 1 along.
// l == 0 writes y (NaN if the solve is refused); the last level (one row) has no neighbours.
__global__ void __launch_bounds__(kBtdThreads) k_btd_back(BtdSrc src, double *pool, double *xpool, BtdLevels lv, int l, real_t *y, int refuse)
{
    __shared__ double s_M[kBtdMaxB * kBtdPitch];
    __shared__ double s_x[2 * kBtdMaxB], s_gj[2 * (32 + 8 * 5)];
    const int b = src.b, t = threadIdx.x;
    const long long m = blockIdx.x, i = 2 * m, n = lv.n[l];
    const bool last = l == lv.nlev - 1;
    const bool hl = !last && m >= 1, hh = !last && i + 1 < n;
    const double *xn = last ? nullptr : xpool + lv.xoff[l + 1];
    if (t < b) {
        s_x[t] = hl ? xn[(m - 1) * b + t] : 0.0;
        s_x[kBtdMaxB + t] = hh ? xn[m * b + t] : 0.0;
    }
    {
        const BtdFrag fB = btd_fetch_block(src, pool, lv, l, i, 1), fA = btd_fetch_block(src, pool, lv, l, i, 0), fC = btd_fetch_block(src, pool, lv, l, i, 2);
        btd_load_rhs(src, pool, lv, l, i, s_M + 3 * b, kBtdPitch);
        btd_put_block(fB, b, s_M, kBtdPitch);
        btd_put_block(fA, b, s_M + b, kBtdPitch);
        btd_put_block(fC, b, s_M + 2 * b, kBtdPitch);
    }
    __syncthreads();
    if (t < b) {       // rhs = d - A x_lo - C x_hi, into column b of [B | rhs]
        double v = s_M[t * kBtdPitch + 3 * b];
        for (int k = 0; k < b; ++k) v -= s_M[t * kBtdPitch + b + k] * s_x[k] + s_M[t * kBtdPitch + 2 * b + k] * s_x[kBtdMaxB + k];
        s_M[t * kBtdPitch + 3 * b] = v;
    }
    __syncthreads();
    if (t < b) s_M[t * kBtdPitch + b] = s_M[t * kBtdPitch + 3 * b];
    __syncthreads();
    btd_gauss_jordan<5>(s_M, s_gj, b, 1);
    if (t < b) {
        const double x = s_M[t * kBtdPitch + b];
        if (l == 0) {
            const bool poison = refuse && (*(volatile int *)src.status & 1);
            const double qn = __longlong_as_double(0x7FF8000000000000ll);
            y[i * b + t] = (real_t)(poison ? qn : x);
            if (hh) y[(i + 1) * b + t] = (real_t)(poison ? qn : s_x[kBtdMaxB + t]);
        } else {
            double *xo = xpool + lv.xoff[l];
            xo[i * b + t] = x;
            if (hh) xo[(i + 1) * b + t] = s_x[kBtdMaxB + t];
        }
    }
}

// diagonal dominance of every row of alpha I + beta J (thread = matrix row; lane-consecutive addresses within a panel column)
__global__ void __launch_bounds__(256) k_btd_check(BtdSrc src)
{
    const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long N = src.nblk * src.b;
    bool bad = false;
    if (row < N) {
        const long long K = row / src.b;
        const int r = (int)(row - K * src.b);
        double diag = 0.0, offd = 0.0;
        for (long long J = K > 0 ? K - 1 : 0; J <= K + 1 && J < src.nblk; ++J)
            for (int c = 0; c < src.b; ++c) {
                const double v = src.beta * (double)src.data[btd_pos(src, K, J, r, c)] + ((J == K && c == r) ? src.alpha : 0.0);
                if (J == K && c == r) diag = fabs(v); else offd += fabs(v);
            }
        bad = !(diag >= offd) || !(diag > 0.0);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(src.status, 1);
}

}  // namespace fdjac

struct fd_blocktridiag_solver {
    fd_ctx *ctx = nullptr;
    int64_t nblk = 0;
    int b = 0;
    fdjac::BtdLevels lv;
    double *pool = nullptr, *xpool = nullptr;
    int *status = nullptr;
    int refuse = 1;
};

using namespace fdjac;

int fd_blocktridiag_solver_create(fd_ctx *ctx, int64_t nblk, int block_size, fd_blocktridiag_solver **out)
{
    FD_REQUIRE(ctx && out, FD_ERR_ARG, "NULL argument");
    *out = nullptr;
    FD_REQUIRE(nblk >= 1, FD_ERR_ARG, "nblk = %lld", (long long)nblk);
    FD_REQUIRE(block_size >= 1 && block_size <= kBtdMaxB, FD_ERR_UNSUPPORTED, "block size %d: the block-tridiagonal solver takes 1 .. %d", block_size, kBtdMaxB);
    FD_HIP_CHECK(hipSetDevice(ctx->device));
    fd_blocktridiag_solver *s = new (std::nothrow) fd_blocktridiag_solver();
    FD_REQUIRE(s != nullptr, FD_ERR_NOMEM, "out of host memory");
    s->ctx = ctx; s->nblk = nblk; s->b = block_size;
    std::memset(&s->lv, 0, sizeof s->lv);
    const int64_t b = block_size, rowd = 3 * b * b + b;
    int64_t n = nblk, off = 0, xoff = 0;
    int nl = 0;
    for (;;) {
        if (nl >= kBtdMaxLevels) { delete s; FD_REQUIRE(false, FD_ERR_UNSUPPORTED, "too many blocks for the block-tridiagonal solver"); }
        s->lv.n[nl] = n; s->lv.off[nl] = off; s->lv.xoff[nl] = xoff;
        if (nl >= 1) { off += rowd * n; xoff += b * n; }
        ++nl;
        if (n == 1) break;
        n /= 2;
    }
    s->lv.nlev = nl;
    hipError_t e = hipMalloc((void **)&s->pool, sizeof(double) * (size_t)(off > 0 ? off : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&s->xpool, sizeof(double) * (size_t)(xoff > 0 ? xoff : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&s->status, sizeof(int));
    if (e != hipSuccess) {
        if (s->pool) (void)hipFree(s->pool);
        if (s->xpool) (void)hipFree(s->xpool);
        delete s;
        set_error("block-tridiagonal solver: %s", hipGetErrorString(e));
        return FD_ERR_NOMEM;
    }
    FD_HIP_CHECK(hipMemsetAsync(s->status, 0, sizeof(int), ctx->stream));
    *out = s;
    return FD_OK;
}

int fd_blocktridiag_solver_destroy(fd_blocktridiag_solver *s)
{
    if (!s) return FD_OK;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->pool) (void)hipFree(s->pool);
    if (s->xpool) (void)hipFree(s->xpool);
    if (s->status) (void)hipFree(s->status);
    delete s;
    return FD_OK;
}

int fd_blocktridiag_solver_set_policy(fd_blocktridiag_solver *s, int trust_non_dominant)
{
    FD_REQUIRE(s != nullptr, FD_ERR_ARG, "solver is NULL");
    s->refuse = trust_non_dominant ? 0 : 1;
    return FD_OK;
}

int fd_blocktridiag_solver_status(fd_blocktridiag_solver *s, int *flags_out)
{
    FD_REQUIRE(s && flags_out, FD_ERR_ARG, "NULL argument");
    FD_HIP_CHECK(hipStreamSynchronize(s->ctx->stream));
    FD_HIP_CHECK(hipMemcpy(flags_out, s->status, sizeof(int), hipMemcpyDeviceToHost));
    return FD_OK;
}

int fd_blocktridiag_solve_async(fd_blocktridiag_solver *s, double alpha, double beta, const void *data, const void *b, void *y)
{
    FD_REQUIRE(s && data && b && y, FD_ERR_ARG, "NULL argument");
    FD_HIP_CHECK(hipSetDevice(s->ctx->device));
    hipStream_t st = s->ctx->stream;
    FD_HIP_CHECK(hipMemsetAsync(s->status, 0, sizeof(int), st));
    BtdSrc src;
    src.data = (const real_t *)data; src.rhs = (const real_t *)b; src.nblk = s->nblk; src.b = s->b;
    src.alpha = alpha; src.beta = beta; src.status = s->status;
    const int64_t N = s->nblk * s->b;
    hipLaunchKernelGGL(k_btd_check, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, src);
    const BtdLevels &lv = s->lv;
    for (int l = 0; l + 1 < lv.nlev; ++l)
        hipLaunchKernelGGL(k_btd_reduce, dim3((unsigned)lv.n[l + 1]), dim3(kBtdThreads), 0, st, src, s->pool, lv, l);
    for (int l = lv.nlev - 1; l >= 0; --l)
        hipLaunchKernelGGL(k_btd_back, dim3((unsigned)((lv.n[l] + 1) / 2)), dim3(kBtdThreads), 0, st, src, s->pool, s->xpool, lv, l, (real_t *)y, s->refuse);
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}
