/*
 * A USER's f! with a block-banded Jacobian that stores it itself (complex step) -- compiled apart from libfdjac, against the two
 * public headers only:
 *
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -Iinclude examples/user_bb_store.hip -o libuser_bb.so
 *
 * The problem: nb blocks of bs unknowns,  f_k(x) = x_k * (s_{b-1} + s_b + s_{b+1}) + sin(x_k)  for k in block b, with the weighted block
 * sums s_b = sum_j w_j x_j, w_j = 1 / (1 + (j mod bs)).  Its Jacobian is block-tridiagonal with dense blocks -- a BlockBandedMatrix in the
 * reference (ext/FiniteDiffBlockBandedMatricesExt.jl:44-68), differentiated with the complex step (src/jacobians.jl:624-637).
 *
 *   user_bb_launch        fd_f_launch: f! on `nbatch` materialised COMPLEX points ((re, im) pairs, is_complex = 1) -- what the library
 *                         calls when the storing launcher is absent or declines
 *   user_bb_launch_lazy   fd_f_launch_lazy registered with FD_LAZY_CAP_STORE: given a `fd_colrange_store` (store_kind =
 *                         FD_STORE_COLRANGE: the plan verified that colorvec is a valid colouring) it evaluates, for every stored
 *                         (row k, column j), imag(f_k(x + i eps_c e_j)) / eps_c at column j's own point and stores it with
 *                         fd_colrange_emit (include/fdjac_device.h) -- no f! output arrays, no decompression launch.
 *
 * examples/user_bb_client.c drives both through the C ABI and checks them against the analytic Jacobian and each other.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fdjac.h"
#include "fdjac_device.h"

namespace {

constexpr int kBlock = 256;
struct cplx { double re, im; };
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx csin(cplx a) { return {sin(a.re) * cosh(a.im), cos(a.re) * sinh(a.im)}; }
__device__ __forceinline__ double weight(long long j, int bs) { return 1.0 / (1.0 + (double)(j % bs)); }

/* plain launcher: one thread per (point, row); the three block sums are re-formed per row (an example, not a fast kernel) */
__global__ void __launch_bounds__(kBlock) k_user_bb_f(double *__restrict__ fx, const double *__restrict__ x, long long nb, int bs, long long xs,
                                                      long long fs, long long r0, long long r1)
{
    const long long k = r0 + (long long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= r1) return;
    const cplx *xb = reinterpret_cast<const cplx *>(x) + (long long)blockIdx.y * xs;
    cplx *fb = reinterpret_cast<cplx *>(fx) + (long long)blockIdx.y * fs;
    const long long b = k / bs, j0 = (b > 0 ? b - 1 : 0) * bs, j1 = (b + 1 < nb ? b + 2 : nb) * bs;
    cplx s = {0.0, 0.0};
    for (long long j = j0; j < j1; ++j) { const cplx xj = xb[j]; const double w = weight(j, bs); s = cadd(s, cplx{w * xj.re, w * xj.im}); }
    fb[k] = cadd(cmul(xb[k], s), csin(xb[k]));
}

/* block sums of the real base point: S[b] = sum of w_j x_j over the blocks b-1 .. b+1 */
__global__ void __launch_bounds__(kBlock) k_user_bb_sums(const double *__restrict__ x, long long nb, int bs, double *__restrict__ S)
{
    const long long b = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (b >= nb) return;
    const long long j0 = (b > 0 ? b - 1 : 0) * bs, j1 = (b + 1 < nb ? b + 2 : nb) * bs;
    double s = 0.0;
    for (long long j = j0; j < j1; ++j) s += weight(j, bs) * x[j];
    S[b] = s;
}

/* storing launcher: one thread per (column j of the batch's colours, row k of the column's three blocks).  At column j's point only
   x_j is perturbed among the columns row k depends on (valid colouring), so  x~_k = x_k + i eps [k == j],  s~ = S_{b(k)} + i eps w_j  and
   imag(f_k) = imag(x~_k * s~ + sin(x~_k)). */
__global__ void __launch_bounds__(kBlock) k_user_bb_store(const double *__restrict__ x, const double *__restrict__ eps, const double *__restrict__ S,
                                                          long long nb, int bs, fd_colrange_store st, int c_lo, int ncolors)
{
    const long long j = st.col_begin + blockIdx.x;                       /* one workgroup per local column */
    const int c = st.color_bytes == 1 ? (int)((const unsigned char *)st.color)[j] : ((const int *)st.color)[j];
    if ((st.color_bytes == 1 && c == 0xFF) || c < c_lo || c >= c_lo + ncolors) return;       /* no colour / another batch */
    const double e = eps[c], wj = weight(j, bs);
    const long long jj = j - st.col_begin;
    for (int t = threadIdx.x; t < st.row_count[jj]; t += kBlock) {
        const long long k = st.row_first[jj] + t;
        const cplx xk = {x[k], k == j ? e : 0.0};
        const cplx s = {S[k / bs], e * wj};
        const cplx v = cadd(cmul(xk, s), csin(xk));
        fd_colrange_emit<double>(&st, j, k, v.im / e);
    }
}

struct UserBB {
    long long nb = 0;
    int bs = 0;
    double *S = nullptr;
    int64_t points = 0;
} g_bb;

}  // namespace

extern "C" {

int user_bb_init(int64_t nb, int bs)
{
    if (g_bb.S) (void)hipFree(g_bb.S);
    g_bb.nb = nb; g_bb.bs = bs; g_bb.points = 0;
    return hipMalloc((void **)&g_bb.S, sizeof(double) * (size_t)nb) == hipSuccess ? 0 : 21;
}
int64_t user_bb_points(void) { return g_bb.points; }

int user_bb_launch(void *fctx, void *fx, const void *x, int64_t nbatch, int64_t x_stride, int64_t fx_stride, int64_t row_begin,
                   int64_t row_end, int is_complex, void *stream)
{
    (void)fctx;
    if (!is_complex) return 22;                                           /* this example serves the complex step only */
    const long long n = g_bb.nb * g_bb.bs, r0 = row_begin < 0 ? 0 : row_begin, r1 = row_end > n ? n : row_end;
    if (nbatch <= 0 || r1 <= r0) return 0;
    hipLaunchKernelGGL(k_user_bb_f, dim3((unsigned)((r1 - r0 + kBlock - 1) / kBlock), (unsigned)nbatch), dim3(kBlock), 0, (hipStream_t)stream,
                       (double *)fx, (const double *)x, g_bb.nb, g_bb.bs, (long long)x_stride, (long long)fx_stride, r0, r1);
    g_bb.points += nbatch;
    return hipGetLastError() == hipSuccess ? 0 : 23;
}

int user_bb_launch_lazy(void *fctx, void *fx, const fd_lazy_points *lp, int64_t fx_stride, int64_t row_begin, int64_t row_end, void *stream)
{
    (void)fctx; (void)fx; (void)fx_stride; (void)row_begin; (void)row_end;
    if (!lp->store || lp->store_kind != FD_STORE_COLRANGE || !lp->is_complex) return FD_LAZY_DECLINED;   /* only the storing form is offered */
    const fd_colrange_store st = *(const fd_colrange_store *)lp->store;
    if (st.elem_bytes != 8 || st.col_end <= st.col_begin) return FD_LAZY_DECLINED;
    const hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_user_bb_sums, dim3((unsigned)((g_bb.nb + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, (const double *)lp->x, g_bb.nb, g_bb.bs, g_bb.S);
    hipLaunchKernelGGL(k_user_bb_store, dim3((unsigned)(st.col_end - st.col_begin)), dim3(kBlock), 0, s, (const double *)lp->x, (const double *)lp->eps,
                       g_bb.S, g_bb.nb, g_bb.bs, st, lp->c_lo, lp->ncolors);
    g_bb.points += lp->ncolors;
    return hipGetLastError() == hipSuccess ? 0 : 23;
}

}  // extern "C"
