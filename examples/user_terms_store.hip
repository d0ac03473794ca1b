/*
 * A USER's SEPARABLE residual storing the Jacobian of a general pattern ROW BY ROW -- compiled apart from libfdjac, against the two
 * public headers only (the offline counterpart of fd_f_compile_terms):
 *
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -Iinclude examples/user_terms_store.hip -o libuser_tm.so
 *
 * The problem: the nine-point (Moore) reaction-diffusion residual on an nx x ny grid, written as the ordered sum over the stored
 * entries (r, j) of row r -- ascending j, i.e. SW, S, SE, W, the point itself, E, NW, N, NE -- of ONE-coordinate terms:
 *     term(r, j, v) = 0.5 v   for a neighbour,      v^3 - 4 v   for j == r.
 * That is the contract of include/fdjac_device.h, "SEPARABLE residuals": the user writes `term` once;
 *   user_tm_bind          takes the row lists of the plan the functor will be used with (fd_plan_row_lists of a plan created with
 *                         FD_PLAN_STORE_CSC | FD_PLAN_STORE_CSC_ROWS) -- fd_sep_rows<MooreTerms> evaluates whole rows from them
 *   user_tm_launch        fd_f_launch: rows at materialised points (fd_sep_rows' call operator)
 *   user_tm_launch_lazy   fd_f_launch_lazy registered with FD_LAZY_CAP_STORE_CSC | FD_LAZY_CAP_STORE_CSC_BASE: on the bound plan (verified
 *                         colouring, locally banded square pattern) the Jacobian is ONE launch of fd_csc_store_rows -- every row's
 *                         plain terms once, for entry k the row's sum with term k perturbed: 2 L term evaluations per row of L
 *                         entries instead of the column store's L^2, the same additions in the same order (same bits); any other
 *                         plan: fd_csc_store_cols with the same functor (src/jacobians.jl:562-568 + ext/FiniteDiffSparseArraysExt.jl:38-47).
 * examples/user_terms_client.c drives it through the C ABI (tests/test_gpu_storetable.py runs that client).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fdjac.h"
#include "fdjac_device.h"

namespace {

constexpr int kBlock = 256;

struct MooreTerms {
    template <class T> __device__ T term(long long r, long long j, T v) const { return j == r ? (v * v) * v - (T)4 * v : (T)0.5 * v; }
};
typedef fd_sep_rows<MooreTerms> Rows;
struct PlainPoint {
    typedef double value_type;
    const double *x;
    __device__ double operator()(long long j) const { return x[j]; }
};

__global__ void __launch_bounds__(kBlock) k_user_tm_f(double *__restrict__ fx, const double *__restrict__ x, Rows f, long long xs, long long fs, long long r0,
                                                      long long r1)
{
    const long long k = r0 + (long long)blockIdx.x * kBlock + threadIdx.x;
    if (k >= r1) return;
    const PlainPoint P = {x + (long long)blockIdx.y * xs};
    fx[(long long)blockIdx.y * fs + k] = f(k, P);
}

Rows g_f = {MooreTerms(), nullptr, nullptr};
long long g_M = 0, g_entries = 0, g_points = 0, g_row_stores = 0;
unsigned long long g_serial = 0;

}  // namespace

extern "C" {

void user_tm_bind(int64_t M, const void *row_ptr_dev, const void *row_col_dev, int64_t entries, uint64_t plan_serial)
{
    g_f.row_ptr = (const int *)row_ptr_dev; g_f.row_col = (const int *)row_col_dev;
    g_M = M; g_entries = entries; g_serial = plan_serial; g_points = 0; g_row_stores = 0;
}
int64_t user_tm_points(void) { return g_points; }
int64_t user_tm_row_stores(void) { return g_row_stores; }

int user_tm_launch(void *fctx, void *fx, const void *x, int64_t nbatch, int64_t x_stride, int64_t fx_stride, int64_t row_begin, int64_t row_end,
                   int is_complex, void *stream)
{
    (void)fctx;
    if (is_complex || !g_f.row_ptr) return 21;
    if (nbatch <= 0) return 0;
    const long long r0 = row_begin < 0 ? 0 : row_begin, r1 = row_end > g_M ? g_M : row_end;
    g_points += nbatch;
    if (r1 <= r0) return 0;
    hipLaunchKernelGGL(k_user_tm_f, dim3((unsigned)((r1 - r0 + kBlock - 1) / kBlock), (unsigned)nbatch), dim3(kBlock), 0, (hipStream_t)stream, (double *)fx,
                       (const double *)x, g_f, (long long)x_stride, (long long)fx_stride, r0, r1);
    return hipGetLastError() == hipSuccess ? 0 : 22;
}

int user_tm_launch_lazy(void *fctx, void *fx, const fd_lazy_points *lp, int64_t fx_stride, int64_t row_begin, int64_t row_end, void *stream)
{
    (void)fctx; (void)fx; (void)fx_stride; (void)row_begin; (void)row_end;
    if (!lp->store || lp->store_kind != FD_STORE_CSC || lp->is_complex || !g_f.row_ptr) return FD_LAZY_DECLINED;
    const fd_csc_store st = *(const fd_csc_store *)lp->store;
    if (st.elem_bytes != 8 || st.col_end <= st.col_begin || st.M != g_M) return FD_LAZY_DECLINED;
    const hipStream_t s = (hipStream_t)stream;
    const double *x = (const double *)lp->x, *eps = (const double *)lp->eps;
    const int c_lo = lp->c_lo, c_hi = lp->c_lo + lp->ncolors;
    const int reach = (int)st.reach;
    const int cap = (int)(256.0 * 1.25 * (double)g_entries / (double)st.M) + 64 < 3072 ? (int)(256.0 * 1.25 * (double)g_entries / (double)st.M) + 64 : 3072;
    const size_t lds = fd_csc_rows_lds_bytes<double>(st.reach, lp->ncolors, cap);
    /* the row-wise store: the plan the lists came from, a colouring it has verified, a square locally banded pattern, every column local */
    if (st.row_ptr && st.row_pack && st.row_tile && st.plan_serial == g_serial && st.valid_coloring && st.reach > 0 && st.reach <= 700 && st.M == st.N && st.N >= 2 &&
        st.col_begin == 0 && st.col_end == st.N && lds <= 64 * 1024) {
        const unsigned gr = fd_xcd_grid((st.M + kBlock - 1) / kBlock);
#define LAUNCH(CT, MODE) hipLaunchKernelGGL((fd_csc_store_rows<double, CT, MODE, Rows>), dim3(gr), dim3(kBlock), lds, s, g_f, x, eps, c_lo, c_hi, st, reach, cap)
        if (st.color_bytes == 1) { if (lp->pts == 2) LAUNCH(unsigned char, 1); else LAUNCH(unsigned char, 0); }
        else { if (lp->pts == 2) LAUNCH(int, 1); else LAUNCH(int, 0); }
#undef LAUNCH
        g_row_stores += 1;
    } else {
        const unsigned g = fd_xcd_grid((st.col_end - st.col_begin + kBlock - 1) / kBlock);
#define LAUNCH(CT, MODE) hipLaunchKernelGGL((fd_csc_store_cols<double, CT, MODE, Rows>), dim3(g), dim3(kBlock), 0, s, g_f, x, eps, c_lo, c_hi, st)
        if (st.color_bytes == 1) { if (lp->pts == 2) LAUNCH(unsigned char, 1); else LAUNCH(unsigned char, 0); }
        else { if (lp->pts == 2) LAUNCH(int, 1); else LAUNCH(int, 0); }
#undef LAUNCH
    }
    g_points += (int64_t)lp->ncolors * lp->pts + ((lp->pts == 1 && !st.fx_base) ? 1 : 0);      /* (f(x) formed inside the launch: one more evaluation) */
    return hipGetLastError() == hipSuccess ? 0 : 23;
}

}  // extern "C"
