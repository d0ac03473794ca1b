// The consumer for wider bands ("banded solve"): (alpha I + beta J) y = b on the device for a BANDED J with
// l, u <= 4 in the storage the banded plans write (BandedMatrix data, or the nzval of the exact band as SparseMatrixCSC) -- the linear
// system of an implicit / Rosenbrock step whose jac_prototype is a BandedMatrix (the tridiagonal case has its own, faster solver:
// fdjac_solve.hip).
//
// Method: the band is block-tridiagonal with K x K blocks, K = max(l, u); BLOCK CYCLIC REDUCTION without pivoting.  Level 0 is read
// straight from J and b (nothing is materialised for it); a reduction step keeps the odd block rows,
//     P = A_i B_{i-1}^-1, Q = C_i B_{i+1}^-1:   A' = -P A_{i-1},  B' = B_i - P C_{i-1} - Q A_{i+1},  C' = -Q C_{i+1},  d' = d_i - P d_{i-1} - Q d_{i+1},
// one thread per kept row (three block rows in, one out: every level is a dense stream, structure-of-arrays so that lanes read
// consecutive addresses); levels of at most kBcrTopRows block rows are finished -- reduced to one row, solved, substituted back --
// inside ONE workgroup; the back-substitution of the large levels solves the even rows from their two known neighbours and, at
// level 0, writes y.  All arithmetic in Float64 whatever the element type.  The elimination does not pivot: every row of
// alpha I + beta J is checked for diagonal dominance while level 0 is read, and a solve that meets a row that is not REFUSES (NaN in
// y, status bit 0) unless the caller vouches for the matrix -- the tridiagonal solver's policy.
#include "fdjac_internal.h"
#include "fdjac_device.h"
#include <cstring>
#include <new>

namespace fdjac {

constexpr int kBcrMaxLevels = 48;
constexpr int kBcrTopRows = 1024;      // block rows the single-workgroup top takes
constexpr int kBcrTopThreads = 256;

// level l >= 1 in the pool: [A: K*K*n][B: K*K*n][C: K*K*n][d: K*n][x: K*n] doubles; element (r, c) of block row i at [(r*K + c)*n + i]
struct BcrLevels {
    int nlev;                            // levels 0 .. nlev-1; level nlev-1 has one block row
    int top;                             // first level the top kernel owns (n <= kBcrTopRows)
    long long n[kBcrMaxLevels];
    long long off[kBcrMaxLevels];        // offset of level l (l >= 1) in the pool, in doubles
};
struct BcrSrc {                          // level 0: the caller's arrays
    const real_t *J, *b;
    long long N;
    int l, u, layout;                    // FD_BAND_CSC / FD_BAND_BANDED (include/fdjac_device.h)
    double alpha, beta;
    int *status;
};

template <int K> struct BcrRow { double A[K][K], B[K][K], C[K][K], d[K]; };

template <int K>
__device__ __forceinline__ void bcr_load(const double *pool, const BcrLevels &lv, int l, long long i, BcrRow<K> &r)
{
    const long long n = lv.n[l];
    const double *p = pool + lv.off[l];
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
        for (int c = 0; c < K; ++c) {
            r.A[a][c] = p[(long long)(a * K + c) * n + i];
            r.B[a][c] = p[(long long)(K * K + a * K + c) * n + i];
            r.C[a][c] = p[(long long)(2 * K * K + a * K + c) * n + i];
        }
#pragma unroll
    for (int a = 0; a < K; ++a) r.d[a] = p[(long long)(3 * K * K + a) * n + i];
}
template <int K>
__device__ __forceinline__ void bcr_store(double *pool, const BcrLevels &lv, int l, long long i, const BcrRow<K> &r)
{
    const long long n = lv.n[l];
    double *p = pool + lv.off[l];
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
        for (int c = 0; c < K; ++c) {
            p[(long long)(a * K + c) * n + i] = r.A[a][c];
            p[(long long)(K * K + a * K + c) * n + i] = r.B[a][c];
            p[(long long)(2 * K * K + a * K + c) * n + i] = r.C[a][c];
        }
#pragma unroll
    for (int a = 0; a < K; ++a) p[(long long)(3 * K * K + a) * n + i] = r.d[a];
}

// block row I of level 0 from the caller's J and b: rows I K .. I K + K - 1 of alpha I + beta J (rows past N: identity, d = 0);
// `bad` is raised by a row that is not diagonally dominant
// where level 0 finds J: the caller's array, or the workgroup's LDS copy of the span its rows read
struct BcrGlobalJ {
    const real_t *J;
    __device__ __forceinline__ double operator()(long long pos) const { return (double)J[pos]; }
};
struct BcrLdsJ {
    const FD_LDS_PTR(real_t) J;
    long long pos0;
    __device__ __forceinline__ double operator()(long long pos) const { return (double)J[pos - pos0]; }
};
__device__ __forceinline__ long long bcr_colpos(const BcrSrc &s, long long j)      // where column j's stored values start
{
    if (s.layout == FD_BAND_BANDED) return (long long)(s.l + s.u + 1) * j;
    fd_band_store bd;
    bd.M = bd.N = s.N; bd.l = s.l; bd.u = s.u; bd.entry_begin = 0; bd.col_begin = 0; bd.col_end = s.N;
    return fd_band_colptr(&bd, j);
}
// the span of J the block rows [I0, I1) read, staged in LDS with lane-consecutive loads (every value of J once per workgroup instead of
// once per row that touches it, at 8 B per lane per request instead of 8 B per 40-72 B); returns the accessor
template <int K>
__device__ __forceinline__ BcrLdsJ bcr_stage(const BcrSrc &s, long long I0, long long I1, FD_LDS_PTR(real_t) lds)
{
    long long r0 = I0 * K, r1 = I1 * K;
    if (r1 > s.N) r1 = s.N;
    const long long c0 = r0 - s.l > 0 ? r0 - s.l : 0, c1 = r1 + s.u < s.N ? r1 + s.u : s.N;
    const long long p0 = bcr_colpos(s, c0), p1 = c1 > c0 ? bcr_colpos(s, c1) : p0;
    for (long long p = threadIdx.x; p < p1 - p0; p += blockDim.x) lds[p] = s.J[p0 + p];
    __syncthreads();
    return BcrLdsJ{lds, p0};
}
template <int K> constexpr int bcr_wg() { return K <= 2 ? 256 : (K == 3 ? 128 : 64); }     // threads (kept rows) per workgroup at level 0
template <int K> constexpr int bcr_stage_elems() { return (2 * bcr_wg<K>() * K + 4 * K + 8) * (2 * K + 1); }

template <int K, typename JA>
__device__ __forceinline__ void bcr_source(const BcrSrc &s, const JA &Jat, long long I, BcrRow<K> &r, bool &bad)
{
    fd_band_store bd;
    bd.M = bd.N = s.N; bd.l = s.l; bd.u = s.u; bd.entry_begin = 0; bd.col_begin = 0; bd.col_end = s.N;
    const int w = s.l + s.u + 1;
#pragma unroll
    for (int a = 0; a < K; ++a) {
#pragma unroll
        for (int c = 0; c < K; ++c) { r.A[a][c] = 0.0; r.B[a][c] = 0.0; r.C[a][c] = 0.0; }
        const long long i = I * K + a;
        if (i >= s.N) { r.B[a][a] = 1.0; r.d[a] = 0.0; continue; }
        r.d[a] = (double)s.b[i];
        double diag = 0.0, offd = 0.0;
#pragma unroll
        for (int t = -K; t <= K; ++t) {
            const long long j = i + t;
            if (t < -s.l || t > s.u || j < 0 || j >= s.N) continue;
            long long pos;
            if (s.layout == FD_BAND_BANDED) pos = (long long)(s.u + i - j) + (long long)w * j;
            else { const long long first = j - s.u > 0 ? j - s.u : 0; pos = fd_band_colptr(&bd, j) + (i - first); }
            const double v = s.beta * Jat(pos) + (t == 0 ? s.alpha : 0.0);
            if (t == 0) diag = fabs(v); else offd += fabs(v);
            const int cb = a + t;                                   // column relative to the block row's first column
            if (cb < 0) r.A[a][cb + K] = v;
            else if (cb < K) r.B[a][cb] = v;
            else r.C[a][cb - K] = v;
        }
        bad = bad || !(diag >= offd) || !(diag > 0.0);
    }
}

// Z = B^-1 R for the K x NR right-hand sides in R (in place), Gauss-Jordan without pivoting, one reciprocal per pivot
template <int K, int NR>
__device__ __forceinline__ void bcr_solve(double (&B)[K][K], double (&R)[K][NR])
{
#pragma unroll
    for (int p = 0; p < K; ++p) {
        const double inv = 1.0 / B[p][p];
#pragma unroll
        for (int c = p + 1; c < K; ++c) B[p][c] *= inv;
#pragma unroll
        for (int c = 0; c < NR; ++c) R[p][c] *= inv;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            if (q == p) continue;
            const double f = B[q][p];
#pragma unroll
            for (int c = p + 1; c < K; ++c) B[q][c] -= f * B[p][c];
#pragma unroll
            for (int c = 0; c < NR; ++c) R[q][c] -= f * R[p][c];
        }
    }
}

// the kept row (odd i) of a reduction step from its two neighbours
template <int K>
__device__ __forceinline__ void bcr_reduce_row(BcrRow<K> &me, BcrRow<K> &lo, BcrRow<K> *hi)
{
    BcrRow<K> out;
    {   // left neighbour: Z = B_lo^-1 [A_lo | C_lo | d_lo]
        double R[K][2 * K + 1];
#pragma unroll
        for (int a = 0; a < K; ++a) {
#pragma unroll
            for (int c = 0; c < K; ++c) { R[a][c] = lo.A[a][c]; R[a][K + c] = lo.C[a][c]; }
            R[a][2 * K] = lo.d[a];
        }
        bcr_solve<K, 2 * K + 1>(lo.B, R);
#pragma unroll
        for (int a = 0; a < K; ++a) {
#pragma unroll
            for (int c = 0; c < K; ++c) {
                double sa = 0.0, sb = 0.0;
#pragma unroll
                for (int t = 0; t < K; ++t) { sa += me.A[a][t] * R[t][c]; sb += me.A[a][t] * R[t][K + c]; }
                out.A[a][c] = -sa;
                out.B[a][c] = me.B[a][c] - sb;
            }
            double sd = 0.0;
#pragma unroll
            for (int t = 0; t < K; ++t) sd += me.A[a][t] * R[t][2 * K];
            out.d[a] = me.d[a] - sd;
        }
    }
    if (hi) {
        double R[K][2 * K + 1];
#pragma unroll
        for (int a = 0; a < K; ++a) {
#pragma unroll
            for (int c = 0; c < K; ++c) { R[a][c] = hi->A[a][c]; R[a][K + c] = hi->C[a][c]; }
            R[a][2 * K] = hi->d[a];
        }
        bcr_solve<K, 2 * K + 1>(hi->B, R);
#pragma unroll
        for (int a = 0; a < K; ++a) {
#pragma unroll
            for (int c = 0; c < K; ++c) {
                double sa = 0.0, sc = 0.0;
#pragma unroll
                for (int t = 0; t < K; ++t) { sa += me.C[a][t] * R[t][c]; sc += me.C[a][t] * R[t][K + c]; }
                out.B[a][c] -= sa;
                out.C[a][c] = -sc;
            }
            double sd = 0.0;
#pragma unroll
            for (int t = 0; t < K; ++t) sd += me.C[a][t] * R[t][2 * K];
            out.d[a] -= sd;
        }
    } else {
#pragma unroll
        for (int a = 0; a < K; ++a)
#pragma unroll
            for (int c = 0; c < K; ++c) out.C[a][c] = 0.0;
    }
    me = out;
}

// x_i = B_i^-1 (d_i - A_i xl - C_i xh) for an eliminated (even) row
template <int K>
__device__ __forceinline__ void bcr_back_row(BcrRow<K> &me, const double *xl, const double *xh, double *x)
{
    double R[K][1];
#pragma unroll
    for (int a = 0; a < K; ++a) {
        double v = me.d[a];
#pragma unroll
        for (int t = 0; t < K; ++t) {
            if (xl) v -= me.A[a][t] * xl[t];
            if (xh) v -= me.C[a][t] * xh[t];
        }
        R[a][0] = v;
    }
    bcr_solve<K, 1>(me.B, R);
#pragma unroll
    for (int a = 0; a < K; ++a) x[a] = R[a][0];
}

// one reduction step: level l >= 1 -> level l + 1
template <int K>
__global__ void __launch_bounds__(256) k_bcr_reduce(double *pool, BcrLevels lv, int l)
{
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= lv.n[l + 1]) return;
    const long long i = 2 * m + 1, n = lv.n[l];
    BcrRow<K> me, lo, hi;
    bcr_load<K>(pool, lv, l, i, me);
    bcr_load<K>(pool, lv, l, i - 1, lo);
    if (i + 1 < n) bcr_load<K>(pool, lv, l, i + 1, hi);
    bcr_reduce_row<K>(me, lo, i + 1 < n ? &hi : nullptr);
    bcr_store<K>(pool, lv, l + 1, m, me);
}
// ... and from the caller's arrays (level 0 -> 1): the workgroup's span of J through LDS
template <int K>
__global__ void __launch_bounds__(bcr_wg<K>()) k_bcr_reduce0(BcrSrc src, double *pool, BcrLevels lv)
{
    __shared__ __attribute__((aligned(16))) real_t s_J[bcr_stage_elems<K>()];
    const long long m0 = (long long)blockIdx.x * bcr_wg<K>(), m = m0 + threadIdx.x;
    const long long n = lv.n[0];
    long long I1 = 2 * (m0 + bcr_wg<K>()) + 1;
    if (I1 > n) I1 = n;
    const BcrLdsJ Jat = bcr_stage<K>(src, 2 * m0, I1, (FD_LDS_PTR(real_t))s_J);
    if (m >= lv.n[1]) return;
    const long long i = 2 * m + 1;
    BcrRow<K> me, lo, hi;
    bool bad = false;
    bcr_source<K>(src, Jat, i, me, bad);
    bcr_source<K>(src, Jat, i - 1, lo, bad);
    if (i + 1 < n) bcr_source<K>(src, Jat, i + 1, hi, bad);
    if (bad) atomicOr(src.status, 1);
    bcr_reduce_row<K>(me, lo, i + 1 < n ? &hi : nullptr);
    bcr_store<K>(pool, lv, 1, m, me);
}

// one back-substitution step: level l from the solution of level l + 1 (l == 0: from the caller's arrays, staged, into y)
template <int K, bool L0>
__global__ void __launch_bounds__(L0 ? bcr_wg<K>() : 256) k_bcr_back(BcrSrc src, double *pool, BcrLevels lv, int l, real_t *y, int refuse)
{
    __shared__ __attribute__((aligned(16))) real_t s_J[L0 ? bcr_stage_elems<K>() : 1];
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = lv.n[l];
    const long long i = 2 * m;
    BcrLdsJ Jat{(FD_LDS_PTR(real_t))s_J, 0};
    if (L0) {
        const long long m0 = (long long)blockIdx.x * blockDim.x;
        long long I1 = 2 * (m0 + blockDim.x);
        if (I1 > n) I1 = n;
        Jat = bcr_stage<K>(src, 2 * m0, I1, (FD_LDS_PTR(real_t))s_J);
    }
    if (i >= n) return;
    const long long nn = lv.n[l + 1];
    const double *xn = pool + lv.off[l + 1] + (long long)(3 * K * K + K) * nn;
    double xl[K], xh[K], x[K];
    const bool hl = m >= 1, hh = i + 1 < n;
#pragma unroll
    for (int a = 0; a < K; ++a) {
        xl[a] = hl ? xn[(long long)a * nn + (m - 1)] : 0.0;
        xh[a] = hh ? xn[(long long)a * nn + m] : 0.0;
    }
    BcrRow<K> me;
    bool bad = false;
    if (L0) bcr_source<K>(src, Jat, i, me, bad);
    else bcr_load<K>(pool, lv, l, i, me);
    bcr_back_row<K>(me, hl ? xl : nullptr, hh ? xh : nullptr, x);
    if (L0) {
        const bool poison = refuse && (*(volatile int *)src.status & 1);
        const double qn = __longlong_as_double(0x7FF8000000000000ll);
#pragma unroll
        for (int a = 0; a < K; ++a) {
            const long long r0 = i * K + a, r1 = (i + 1) * K + a;
            if (r0 < src.N) y[r0] = (real_t)(poison ? qn : x[a]);
            if (hh && r1 < src.N) y[r1] = (real_t)(poison ? qn : xh[a]);
        }
    } else {
        double *xo = pool + lv.off[l] + (long long)(3 * K * K + K) * n;
#pragma unroll
        for (int a = 0; a < K; ++a) {
            xo[(long long)a * n + i] = x[a];
            if (hh) xo[(long long)a * n + i + 1] = xh[a];
        }
    }
}

// levels lv.top .. nlev-1 inside one workgroup: reduce to one block row, solve it, substitute back down to level lv.top
// (lv.top == 0 -- a small system -- reads the caller's arrays and writes y itself)
template <int K>
__global__ void __launch_bounds__(kBcrTopThreads) k_bcr_top(BcrSrc src, double *pool, BcrLevels lv, real_t *y, int refuse)
{
    const int t = threadIdx.x;
    const BcrGlobalJ Jg{src.J};
    __shared__ int s_bad;
    if (t == 0) s_bad = 0;
    __syncthreads();
    for (int l = lv.top; l + 1 < lv.nlev; ++l) {
        const long long n = lv.n[l];
        for (long long m = t; m < lv.n[l + 1]; m += kBcrTopThreads) {
            const long long i = 2 * m + 1;
            BcrRow<K> me, lo, hi;
            bool bad = false;
            if (l == 0) {
                bcr_source<K>(src, Jg, i, me, bad);
                bcr_source<K>(src, Jg, i - 1, lo, bad);
                if (i + 1 < n) bcr_source<K>(src, Jg, i + 1, hi, bad);
                if (bad) s_bad = 1;
            } else {
                bcr_load<K>(pool, lv, l, i, me);
                bcr_load<K>(pool, lv, l, i - 1, lo);
                if (i + 1 < n) bcr_load<K>(pool, lv, l, i + 1, hi);
            }
            bcr_reduce_row<K>(me, lo, i + 1 < n ? &hi : nullptr);
            bcr_store<K>(pool, lv, l + 1, m, me);
        }
        __syncthreads();
    }
    const int last = lv.nlev - 1;
    if (t == 0) {      // the last level: one block row
        BcrRow<K> me;
        bool bad = false;
        double x[K];
        if (last == 0) { bcr_source<K>(src, Jg, 0, me, bad); if (bad) s_bad = 1; }
        else bcr_load<K>(pool, lv, last, 0, me);
        bcr_back_row<K>(me, nullptr, nullptr, x);
        if (last == 0) {
            const bool poison = refuse && s_bad;
#pragma unroll
            for (int a = 0; a < K; ++a) if (a < src.N) y[a] = (real_t)(poison ? __longlong_as_double(0x7FF8000000000000ll) : x[a]);
        } else {
            double *xo = pool + lv.off[last] + (long long)(3 * K * K + K) * lv.n[last];
#pragma unroll
            for (int a = 0; a < K; ++a) xo[a] = x[a];
        }
    }
    __syncthreads();
    if (lv.top == 0 && s_bad && t == 0) atomicOr(src.status, 1);
    for (int l = last - 1; l >= lv.top; --l) {
        const long long n = lv.n[l], nn = lv.n[l + 1];
        const double *xn = pool + lv.off[l + 1] + (long long)(3 * K * K + K) * nn;
        for (long long m = t; 2 * m < n; m += kBcrTopThreads) {
            const long long i = 2 * m;
            double xl[K], xh[K], x[K];
            const bool hl = m >= 1, hh = i + 1 < n;
#pragma unroll
            for (int a = 0; a < K; ++a) {
                xl[a] = hl ? xn[(long long)a * nn + (m - 1)] : 0.0;
                xh[a] = hh ? xn[(long long)a * nn + m] : 0.0;
            }
            BcrRow<K> me;
            bool bad = false;
            if (l == 0) bcr_source<K>(src, Jg, i, me, bad);
            else bcr_load<K>(pool, lv, l, i, me);
            bcr_back_row<K>(me, hl ? xl : nullptr, hh ? xh : nullptr, x);
            if (l == 0) {
                const bool poison = refuse && s_bad;
                const double qn = __longlong_as_double(0x7FF8000000000000ll);
#pragma unroll
                for (int a = 0; a < K; ++a) {
                    const long long r0 = i * K + a, r1 = (i + 1) * K + a;
                    if (r0 < src.N) y[r0] = (real_t)(poison ? qn : x[a]);
                    if (hh && r1 < src.N) y[r1] = (real_t)(poison ? qn : xh[a]);
                }
            } else {
                double *xo = pool + lv.off[l] + (long long)(3 * K * K + K) * n;
#pragma unroll
                for (int a = 0; a < K; ++a) {
                    xo[(long long)a * n + i] = x[a];
                    if (hh) xo[(long long)a * n + i + 1] = xh[a];
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace fdjac

struct fd_banded_solver {
    fd_ctx *ctx = nullptr;
    int64_t N = 0;
    int l = 0, u = 0, K = 1, layout = 0;
    fdjac::BcrLevels lv;
    double *pool = nullptr;
    int *status = nullptr;
    int refuse = 1;
};

using namespace fdjac;

int fd_banded_solver_create(fd_ctx *ctx, int64_t N, int l, int u, int layout, fd_banded_solver **out)
{
    FD_REQUIRE(ctx && out, FD_ERR_ARG, "NULL argument");
    *out = nullptr;
    FD_REQUIRE(N >= 1, FD_ERR_ARG, "N = %lld", (long long)N);
    FD_REQUIRE(l >= 0 && u >= 0 && l <= 4 && u <= 4 && l + u >= 1, FD_ERR_UNSUPPORTED, "bandwidths (%d, %d): the banded solver takes 0 <= l, u <= 4, l + u >= 1", l, u);
    FD_REQUIRE(layout == FD_BAND_SOLVE_CSC || layout == FD_BAND_SOLVE_BANDED, FD_ERR_ARG, "layout %d", layout);
    FD_HIP_CHECK(hipSetDevice(ctx->device));
    fd_banded_solver *s = new (std::nothrow) fd_banded_solver();
    FD_REQUIRE(s != nullptr, FD_ERR_NOMEM, "out of host memory");
    s->ctx = ctx; s->N = N; s->l = l; s->u = u; s->K = l > u ? l : u; s->layout = layout;
    const int K = s->K;
    std::memset(&s->lv, 0, sizeof s->lv);
    int64_t n = (N + K - 1) / K, off = 0;
    int nl = 0;
    s->lv.top = -1;
    for (;;) {
        if (nl >= kBcrMaxLevels) { delete s; FD_REQUIRE(false, FD_ERR_UNSUPPORTED, "N too large for the banded solver"); }
        s->lv.n[nl] = n;
        s->lv.off[nl] = off;
        if (nl >= 1) off += (int64_t)(3 * K * K + 2 * K) * n;
        if (s->lv.top < 0 && n <= kBcrTopRows) s->lv.top = nl;
        ++nl;
        if (n == 1) break;
        n = n / 2;
    }
    s->lv.nlev = nl;
    hipError_t e = hipMalloc((void **)&s->pool, sizeof(double) * (size_t)(off > 0 ? off : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&s->status, sizeof(int));
    if (e != hipSuccess) {
        if (s->pool) (void)hipFree(s->pool);
        delete s;
        set_error("banded solver: %s", hipGetErrorString(e));
        return FD_ERR_NOMEM;
    }
    FD_HIP_CHECK(hipMemsetAsync(s->status, 0, sizeof(int), ctx->stream));
    *out = s;
    return FD_OK;
}

int fd_banded_solver_destroy(fd_banded_solver *s)
{
    if (!s) return FD_OK;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->pool) (void)hipFree(s->pool);
    if (s->status) (void)hipFree(s->status);
    delete s;
    return FD_OK;
}

int fd_banded_solver_set_policy(fd_banded_solver *s, int trust_non_dominant)
{
    FD_REQUIRE(s != nullptr, FD_ERR_ARG, "solver is NULL");
    s->refuse = trust_non_dominant ? 0 : 1;
    return FD_OK;
}

int fd_banded_solver_status(fd_banded_solver *s, int *flags_out)
{
    FD_REQUIRE(s && flags_out, FD_ERR_ARG, "NULL argument");
    FD_HIP_CHECK(hipStreamSynchronize(s->ctx->stream));
    FD_HIP_CHECK(hipMemcpy(flags_out, s->status, sizeof(int), hipMemcpyDeviceToHost));
    return FD_OK;
}

template <int K>
static int banded_solve_k(fd_banded_solver *s, const BcrSrc &src, real_t *y)
{
    hipStream_t st = s->ctx->stream;
    const BcrLevels &lv = s->lv;
    constexpr int T0 = bcr_wg<K>();
    for (int l = 0; l < lv.top; ++l) {
        if (l == 0) hipLaunchKernelGGL((k_bcr_reduce0<K>), dim3((unsigned)((lv.n[1] + T0 - 1) / T0)), dim3(T0), 0, st, src, s->pool, lv);
        else hipLaunchKernelGGL((k_bcr_reduce<K>), dim3((unsigned)((lv.n[l + 1] + 255) / 256)), dim3(256), 0, st, s->pool, lv, l);
    }
    hipLaunchKernelGGL((k_bcr_top<K>), dim3(1), dim3(kBcrTopThreads), 0, st, src, s->pool, lv, y, s->refuse);
    for (int l = lv.top - 1; l >= 0; --l) {
        const int64_t half = (lv.n[l] + 1) / 2;
        if (l == 0) hipLaunchKernelGGL((k_bcr_back<K, true>), dim3((unsigned)((half + T0 - 1) / T0)), dim3(T0), 0, st, src, s->pool, lv, l, y, s->refuse);
        else hipLaunchKernelGGL((k_bcr_back<K, false>), dim3((unsigned)((half + 255) / 256)), dim3(256), 0, st, src, s->pool, lv, l, y, s->refuse);
    }
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

int fd_banded_solve_async(fd_banded_solver *s, double alpha, double beta, const void *J, const void *b, void *y)
{
    FD_REQUIRE(s && J && b && y, FD_ERR_ARG, "NULL argument");
    FD_HIP_CHECK(hipSetDevice(s->ctx->device));
    FD_HIP_CHECK(hipMemsetAsync(s->status, 0, sizeof(int), s->ctx->stream));
    BcrSrc src;
    src.J = (const real_t *)J; src.b = (const real_t *)b; src.N = s->N; src.l = s->l; src.u = s->u;
    src.layout = s->layout == FD_BAND_SOLVE_BANDED ? FD_BAND_BANDED : FD_BAND_CSC;
    src.alpha = alpha; src.beta = beta; src.status = s->status;
    switch (s->K) {
    case 1: return banded_solve_k<1>(s, src, (real_t *)y);
    case 2: return banded_solve_k<2>(s, src, (real_t *)y);
    case 3: return banded_solve_k<3>(s, src, (real_t *)y);
    default: return banded_solve_k<4>(s, src, (real_t *)y);
    }
}
