/*
 * A USER's f! on a pattern with NO closed-form layout, storing the Jacobian itself, column by column, through the plan's compact
 * copy of the pattern -- compiled apart from libfdjac, against the two public headers only:
 *
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -Iinclude examples/user_csc_store.hip -o libuser_rl.so
 *
 * The problem: a reaction-diffusion residual on an nx x ny grid with the NINE-point (Moore) neighbourhood,
 *     f_k(x) = sum over the 8 neighbours n of k (in the order SW, S, SE, W, E, NW, N, NE; outside the grid: 0) of 0.5 x_n
 *              - 4 x_k + x_k^3,
 * whose Jacobian has 9 entries per row: neither a band nor the 5-point stencil, so the library has no closed-form store
 * descriptor for it.  The user writes the residual ONCE, as a device functor  f(r, X)  that reads coordinate j as X(j):
 *   user_rl_launch        fd_f_launch: the functor on `nbatch` materialised points (X(j) = x[j])
 *   user_rl_launch_lazy   fd_f_launch_lazy registered with FD_LAZY_CAP_STORE_CSC: the SAME functor inside
 *                         fd_csc_store_cols (include/fdjac_device.h) -- X(j) = x[j] + eps_c (color[j] == c) -- evaluates, for
 *                         every stored entry of every column, the entry's row at the column's colour point, subtracts f(x)
 *                         (computed once by user_rl_launch), divides and stores into nzval in storage order
 *                         (src/jacobians.jl:562-568 + ext/FiniteDiffSparseArraysExt.jl:38-47).  Any other request is declined.
 * examples/user_csc_client.c drives it through the C ABI (tests/test_gpu_storetable.py runs that client).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fdjac.h"
#include "fdjac_device.h"

namespace {

constexpr int kBlock = 256;

struct Moore9 {
    long long nx, ny;
    template <class P> __device__ double operator()(long long k, const P &X) const
    {
        const long long j = k / nx, i = k - j * nx;
        double s = 0.0;
        bool first = true;
        for (int dj = -1; dj <= 1; ++dj)
            for (int di = -1; di <= 1; ++di) {
                if (di == 0 && dj == 0) continue;
                const long long ii = i + di, jj = j + dj;
                // (the coordinate is fetched unconditionally, from a clamped index, and selected away outside the grid: a load inside a
                //  per-lane conditional is waited for on its own -- eight serial memory round trips per row instead of one)
                const bool in = ii >= 0 && ii < nx && jj >= 0 && jj < ny;
                const double xv = X(in ? jj * nx + ii : k);
                const double v = in ? 0.5 * xv : 0.0;
                s = first ? v : s + v;
                first = false;
            }
        const double c = X(k);
        return (s - 4.0 * c) + (c * c) * c;
    }
};
struct PlainPoint {
    const double *x;
    __device__ double operator()(long long j) const { return x[j]; }
};

__global__ void __launch_bounds__(kBlock) k_user_rl_f(double *__restrict__ fx, const double *__restrict__ x, Moore9 f, long long xs, long long fs,
                                                      long long r0, long long r1)
{
    const long long k = r0 + (long long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= r1) return;
    const PlainPoint P = {x + (long long)blockIdx.y * xs};
    fx[(long long)blockIdx.y * fs + k] = f(k, P);
}

Moore9 g_f = {0, 0};
long long g_points = 0;

}  // namespace

extern "C" {

void user_rl_init(int64_t nx, int64_t ny) { g_f.nx = nx; g_f.ny = ny; g_points = 0; }
int64_t user_rl_points(void) { return g_points; }

int user_rl_launch(void *fctx, void *fx, const void *x, int64_t nbatch, int64_t x_stride, int64_t fx_stride, int64_t row_begin, int64_t row_end,
                   int is_complex, void *stream)
{
    (void)fctx;
    if (is_complex) return 21;
    if (nbatch <= 0) return 0;
    const long long n = g_f.nx * g_f.ny, r0 = row_begin < 0 ? 0 : row_begin, r1 = row_end > n ? n : row_end;
    g_points += nbatch;
    if (r1 <= r0) return 0;
    hipLaunchKernelGGL(k_user_rl_f, dim3((unsigned)((r1 - r0 + kBlock - 1) / kBlock), (unsigned)nbatch), dim3(kBlock), 0, (hipStream_t)stream, (double *)fx,
                       (const double *)x, g_f, (long long)x_stride, (long long)fx_stride, r0, r1);
    return hipGetLastError() == hipSuccess ? 0 : 22;
}

int user_rl_launch_lazy(void *fctx, void *fx, const fd_lazy_points *lp, int64_t fx_stride, int64_t row_begin, int64_t row_end, void *stream)
{
    (void)fctx; (void)fx; (void)fx_stride; (void)row_begin; (void)row_end;
    if (!lp->store || lp->store_kind != FD_STORE_CSC || lp->is_complex) return FD_LAZY_DECLINED;
    const fd_csc_store st = *(const fd_csc_store *)lp->store;
    if (st.elem_bytes != 8 || st.col_end <= st.col_begin || (lp->pts == 1 && !st.fx_base)) return FD_LAZY_DECLINED;
    const unsigned g = fd_xcd_grid((st.col_end - st.col_begin + kBlock - 1) / kBlock);
    const hipStream_t s = (hipStream_t)stream;
    const double *x = (const double *)lp->x, *eps = (const double *)lp->eps;
    const int c_lo = lp->c_lo, c_hi = lp->c_lo + lp->ncolors;
#define LAUNCH(CT, MODE) hipLaunchKernelGGL((fd_csc_store_cols<double, CT, MODE, Moore9>), dim3(g), dim3(kBlock), 0, s, g_f, x, eps, c_lo, c_hi, st)
    if (st.color_bytes == 1) { if (lp->pts == 2) LAUNCH(unsigned char, 1); else LAUNCH(unsigned char, 0); }
    else { if (lp->pts == 2) LAUNCH(int, 1); else LAUNCH(int, 0); }
#undef LAUNCH
    g_points += (int64_t)lp->ncolors * lp->pts;      /* (f(x) of a forward difference was one call of user_rl_launch) */
    return hipGetLastError() == hipSuccess ? 0 : 23;
}

}  // extern "C"
