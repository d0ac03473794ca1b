/*
 * A USER's f! that stores the Jacobian itself -- compiled apart from libfdjac, against the two public headers only:
 *
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -Iinclude examples/user_f_store.hip -o libuser_f.so
 *
 * The problem: a convection-diffusion residual on a line,
 *     f_i(x) = (x[i-1] - 2 x[i]) + x[i+1] + 0.25 * x[i] * (x[i+1] - x[i-1]),        x[-1] = x[N] = 0,
 * whose Jacobian is tridiagonal and depends on x:  df_i/dx_{i-1} = 1 - 0.25 x_i,  df_i/dx_i = -2 + 0.25 (x_{i+1} - x_{i-1}),
 * df_i/dx_{i+1} = 1 + 0.25 x_i.  In the reference this is `f!(dx, x)` handed to FiniteDiff.finite_difference_jacobian!
 * (src/jacobians.jl:504); here it is three launchers behind include/fdjac.h:
 *
 *   user_f_launch        fd_f_launch: f! on `nbatch` materialised points (always required; also what the library falls back to)
 *   user_f_launch_lazy   fd_f_launch_lazy registered with FD_LAZY_CAP_STORE: when the plan hands over a `fd_band_store`
 *                        (fd_lazy_points.store) the launch evaluates f! at the lazily perturbed points, forms the difference
 *                        quotients (src/jacobians.jl:565 / 607) and stores them where the Jacobian keeps them
 *                        (ext/FiniteDiffSparseArraysExt.jl:38-47; BandedMatrix data; Tridiagonal dl / d / du) through
 *                        include/fdjac_device.h -- no decompression launch follows.  Any other request is declined
 *                        (FD_LAZY_DECLINED): the library then materialises the points and calls user_f_launch.
 *   user_f_set_mode      0: row-centric kernel, one fd_band_emit per (row, colour) -- the ten-line version;
 *                        1: column-centric kernel + fd_band_emit_wave (dense 16-byte stores through a wave-private LDS
 *                        window) -- the bandwidth-bound version, the structure of the library's built-in launcher.
 *
 * examples/user_store_client.c drives it through the C ABI; tests/test_gpu_edge.py runs that client and compares the
 * stored values with the CPU oracle.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fdjac.h"
#include "fdjac_device.h"

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int kBlock = 256;

__device__ __forceinline__ double f_row(double xm, double xi, double xp)
{
    return ((xm - 2.0 * xi) + xp) + (0.25 * xi) * (xp - xm);
}

__device__ __forceinline__ double x_at(const double *__restrict__ x, long long i, long long n) { return (i >= 0 && i < n) ? x[i] : 0.0; }

/* plain launcher: nbatch independent points */
__global__ void __launch_bounds__(kBlock) k_user_f(double *__restrict__ fx, const double *__restrict__ x, long long n, long long xs,
                                                   long long fs, long long r0, long long r1)
{
    const double *xb = x + (long long)blockIdx.y * xs;
    double *fb = fx + (long long)blockIdx.y * fs;
    for (long long i = r0 + (long long)blockIdx.x * kBlock + threadIdx.x; i < r1; i += (long long)gridDim.x * kBlock)
        fb[i] = f_row(x_at(xb, i - 1, n), xb[i], x_at(xb, i + 1, n));
}

/* mode 0 -- row-centric: thread i owns row i, evaluates f_i at the base point and at the point of every colour of the batch
   (x~[j] = x[j] +- eps_c * [colour(j) == c]) and emits the quotient of (row i, colour c) */
template <bool CENTRAL>
__global__ void __launch_bounds__(kBlock) k_user_store_rows(const double *__restrict__ x, const double *__restrict__ eps, long long n,
                                                            fd_band_store st, int c_lo, int ncolors, long long r0, long long r1)
{
    const long long i = r0 + (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= r1) return;
    const double xm = x_at(x, i - 1, n), xi = x[i], xp = x_at(x, i + 1, n);
    const int cm = i - 1 >= 0 ? fd_band_color(&st, i - 1) : -1, ci = fd_band_color(&st, i), cp = i + 1 < n ? fd_band_color(&st, i + 1) : -1;
    const double base = f_row(xm, xi, xp);
    for (int c = c_lo; c < c_lo + ncolors; ++c) {
        const double e = eps[c];
        const double dm = cm == c ? e : 0.0, di = ci == c ? e : 0.0, dp = cp == c ? e : 0.0;
        const double plus = f_row(xm + dm, xi + di, xp + dp);
        const double q = CENTRAL ? (plus - f_row(xm - dm, xi - di, xp - dp)) / (2 * e) : (plus - base) / e;
        fd_band_emit<double>(&st, i, c, q);
    }
}

/* mode 1 -- column-centric: lane t of a wavefront owns the columns j = jw + 2t, j + 1, loads x[j-2 .. j+3] as three aligned
   pairs, evaluates the three rows each column touches at x and at x +- eps e_j (what the colour's point looks like from those
   rows), and hands its six quotients to fd_band_emit_wave */
__device__ __forceinline__ d2 ld_pair(const double *__restrict__ x, long long a, long long n)
{
    if (a >= 0 && a + 1 < n) return *reinterpret_cast<const d2 *>(x + a);
    d2 v = {0.0, 0.0};
    if (a >= 0 && a < n) v.x = x[a];
    return v;
}
template <bool CENTRAL>
__global__ void __launch_bounds__(kBlock) k_user_store_wave(const double *__restrict__ x, const double *__restrict__ eps, long long n,
                                                            fd_band_store st, long long jstart)
{
    __shared__ __attribute__((aligned(16))) double win[kBlock / 64][FD_BAND_WAVE_LDS(3)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long jw = jstart + ((long long)blockIdx.x * (kBlock / 64) + wave) * 128;
    if (jw >= st.col_end) return;
    const long long j = jw + 2 * lane;
    const d2 C = ld_pair(x, j, n), L = ld_pair(x, j - 2, n), R = ld_pair(x, j + 2, n);
    const double xv[6] = {L.x, L.y, C.x, C.y, R.x, R.y};
    double q[6];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const double e = eps[fd_band_color(&st, j + o)];
#pragma unroll
        for (int k = 0; k < 3; ++k) {                              /* row j + o - 1 + k sees column j + o at position 2 - k */
            double p[3], m[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) { const double d = (k + t == 2) ? e : 0.0; p[t] = xv[o + k + t] + d; m[t] = xv[o + k + t] - d; }
            const double plus = f_row(p[0], p[1], p[2]);
            q[3 * o + k] = CENTRAL ? (plus - f_row(m[0], m[1], m[2])) / (2 * e)
                                   : (plus - f_row(xv[o + k], xv[o + k + 1], xv[o + k + 2])) / e;
        }
    }
    fd_band_emit_wave<double, 3>(&st, win[wave], jw, q);
}

struct UserF {
    long long n;
    int mode;
    long long launches, points;
};
UserF g_user = {0, 1, 0, 0};

}  // namespace

extern "C" {

void user_f_init(int64_t n, int mode) { g_user.n = n; g_user.mode = mode; g_user.launches = g_user.points = 0; }
void user_f_set_mode(int mode) { g_user.mode = mode; }
int64_t user_f_points(void) { return g_user.points; }

int user_f_launch(void *fctx, void *fx, const void *x, int64_t nbatch, int64_t x_stride, int64_t fx_stride, int64_t row_begin,
                  int64_t row_end, int is_complex, void *stream)
{
    (void)fctx;
    if (is_complex) return 21;                       /* (this example has no complex-step evaluation) */
    if (nbatch <= 0) return 0;
    const long long r0 = row_begin < 0 ? 0 : row_begin, r1 = row_end > g_user.n ? g_user.n : row_end;
    g_user.launches += 1;
    g_user.points += nbatch;
    if (r1 <= r0) return 0;
    const unsigned g = (unsigned)((r1 - r0 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_user_f, dim3(g < 65535u * 16u ? g : 65535u * 16u, (unsigned)nbatch), dim3(kBlock), 0, (hipStream_t)stream, (double *)fx,
                       (const double *)x, g_user.n, (long long)x_stride, (long long)fx_stride, r0, r1);
    return hipGetLastError() == hipSuccess ? 0 : 22;
}

int user_f_launch_lazy(void *fctx, void *fx, const fd_lazy_points *lp, int64_t fx_stride, int64_t row_begin, int64_t row_end, void *stream)
{
    (void)fctx; (void)fx; (void)fx_stride;
    if (!lp->store || lp->is_complex) return FD_LAZY_DECLINED;        /* only the storing form is offered */
    const fd_band_store st = *(const fd_band_store *)lp->store;
    if (st.elem_bytes != 8 || st.l != 1 || st.u != 1 || st.M != st.N) return FD_LAZY_DECLINED;
    const long long r0 = row_begin < 0 ? 0 : row_begin, r1 = row_end > g_user.n ? g_user.n : row_end;
    const bool central = lp->pts == 2;
    const hipStream_t s = (hipStream_t)stream;
    const bool all_colours = lp->c_lo == 0 && lp->ncolors == st.C;
    if (g_user.mode == 1 && all_colours && (((uintptr_t)lp->x) & 15) == 0) {
        const long long jstart = st.col_begin & ~1ll;
        const long long nwaves = (st.col_end - jstart + 127) / 128;
        const unsigned g = (unsigned)((nwaves + kBlock / 64 - 1) / (kBlock / 64));
        if (central) hipLaunchKernelGGL(k_user_store_wave<true>, dim3(g), dim3(kBlock), 0, s, (const double *)lp->x, (const double *)lp->eps, g_user.n, st, jstart);
        else hipLaunchKernelGGL(k_user_store_wave<false>, dim3(g), dim3(kBlock), 0, s, (const double *)lp->x, (const double *)lp->eps, g_user.n, st, jstart);
    } else {
        if (r1 <= r0) return 0;
        const unsigned g = (unsigned)((r1 - r0 + kBlock - 1) / kBlock);
        if (central) hipLaunchKernelGGL(k_user_store_rows<true>, dim3(g), dim3(kBlock), 0, s, (const double *)lp->x, (const double *)lp->eps, g_user.n, st, lp->c_lo, lp->ncolors, r0, r1);
        else hipLaunchKernelGGL(k_user_store_rows<false>, dim3(g), dim3(kBlock), 0, s, (const double *)lp->x, (const double *)lp->eps, g_user.n, st, lp->c_lo, lp->ncolors, r0, r1);
    }
    g_user.launches += 1;
    g_user.points += (int64_t)lp->ncolors * lp->pts + (lp->diff == 2 ? 1 : 0);   /* f(x) rides in the same launch (forward) */
    return hipGetLastError() == hipSuccess ? 0 : 23;
}

}  // extern "C"
