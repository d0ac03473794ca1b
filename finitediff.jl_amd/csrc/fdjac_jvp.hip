// Jacobian-vector products by finite differences on the device: FiniteDiff.finite_difference_jvp!
// (reference: src/jvp.jl:238-274).  A coloured-Jacobian colour step IS a JVP with v = colour mask,
// so this reuses the same pieces: a deterministic reduction for the step size, a perturbation pass
// written from the pristine x, the fd_f_launch boundary, and a fused difference.
//   eps     = compute_epsilon(fdtype, sqrt(abs(dot(x, v))), relstep, absstep, dir)   (src/jvp.jl:253-254)
//   forward : jvp = (f(x + eps v) - f(x)) / eps                                     (:255-263)
//   central : jvp = (f(x + eps v) - f(x - eps v)) / (2 eps)                         (:264-269)
#include <cmath>
#include <limits>
#include <new>

#include "fdjac_internal.h"

namespace fdjac {

int balanced_grid(int64_t tiles, int64_t cap);

__device__ __forceinline__ double wave_sum_j(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

typedef r2_t d2j_t;

// VEC: x and v are 16-B aligned -> one 16-B load per array per lane (2 elements); else scalar.  Each variant is
// deterministic (fixed per-thread order, fixed shuffle tree, partials reduced in fixed order by k_jvp_eps).
template <bool VEC>
__global__ void __launch_bounds__(kBlock)
k_dot_partial(const real_t *__restrict__ x, const real_t *__restrict__ v, int64_t n, double *__restrict__ partial)
{
    double acc = 0.0;   // Float64 accumulation for either element type
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    if (VEC) {
        const int64_t n2 = n >> 1;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
            const d2j_t a = reinterpret_cast<const d2j_t *>(x)[i], b = reinterpret_cast<const d2j_t *>(v)[i];
            acc += (double)a.x * (double)b.x;
            acc += (double)a.y * (double)b.y;
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc += (double)x[n - 1] * (double)v[n - 1];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc += (double)x[i] * (double)v[i];
    }
    __shared__ double red[kBlock / 64];
    acc = wave_sum_j(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) t += red[w];
        partial[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(kBlock)
k_jvp_eps(const double *__restrict__ partial, int nparts, double relstep, double absstep, double dir, int is_forward,
          real_t *__restrict__ eps)
{
    double acc = 0.0;
    for (int k = threadIdx.x; k < nparts; k += kBlock) acc += partial[k];
    __shared__ double red[kBlock / 64];
    acc = wave_sum_j(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) t += red[w];
        const real_t tmp = sqrt(fabs((real_t)t));      // the dot product in Float64, the step rule in the element type
        const real_t a = (real_t)relstep * fabs(tmp);
        real_t e = (a > (real_t)absstep) ? a : (real_t)absstep;   // src/epsilons.jl:26-29,50-53
        if (is_forward) e = e * (real_t)dir;
        eps[0] = e;
    }
}

// forward: X[0] = x + eps v ; central: X[0] = x - eps v, X[1] = x + eps v   (src/jvp.jl:260,265,267)
template <bool VEC>
__global__ void __launch_bounds__(kBlock)
k_jvp_points(const real_t *__restrict__ x, const real_t *__restrict__ v, const real_t *__restrict__ eps, int central,
             int64_t n, real_t *__restrict__ X, int64_t ld)
{
    const real_t e = eps[0];
    if (VEC) {   // one pair of elements per thread (ld is a multiple of 32, X is hipMalloc'ed: 16-B aligned rows)
        const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 2;
        if (i + 1 < n) {
            const d2j_t xi = *reinterpret_cast<const d2j_t *>(x + i), vi = *reinterpret_cast<const d2j_t *>(v + i);
            const d2j_t ev = {e * vi.x, e * vi.y};
            if (central) {
                *reinterpret_cast<d2j_t *>(X + i) = d2j_t{xi.x - ev.x, xi.y - ev.y};
                *reinterpret_cast<d2j_t *>(X + ld + i) = d2j_t{xi.x + ev.x, xi.y + ev.y};
            } else {
                *reinterpret_cast<d2j_t *>(X + i) = d2j_t{xi.x + ev.x, xi.y + ev.y};
            }
        } else if (i < n) {
            const real_t xi = x[i], ev = e * v[i];
            if (central) { X[i] = xi - ev; X[ld + i] = xi + ev; } else { X[i] = xi + ev; }
        }
        return;
    }
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const real_t xi = x[i], ev = e * v[i];
        if (central) {
            X[i] = xi - ev;
            X[ld + i] = xi + ev;
        } else {
            X[i] = xi + ev;
        }
    }
}

// Small problems (N <= kSmallN): dot product, step size and the perturbed point(s) in ONE single-workgroup launch, with
// x itself as one more batch member (forward arm without f_in) so that f(x) and f(x + eps v) come out of one f! launch:
// three launches per JVP instead of six -- the inner loop of a Newton-Krylov solver is launch-latency bound.
constexpr int kJvpSmallBlock = 1024;
__global__ void __launch_bounds__(kJvpSmallBlock)
k_jvp_small(const real_t *__restrict__ x, const real_t *__restrict__ v, int64_t n, double relstep, double absstep,
            double dir, int central, int base_row, real_t *__restrict__ eps, real_t *__restrict__ X, int64_t ld)
{
    __shared__ double red[kJvpSmallBlock / 64];
    __shared__ real_t s_e;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += kJvpSmallBlock) acc += (double)x[i] * (double)v[i];
    acc = wave_sum_j(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kJvpSmallBlock / 64; ++w) t += red[w];
        const real_t tmp = sqrt(fabs((real_t)t));
        const real_t a = (real_t)relstep * fabs(tmp);
        real_t e = (a > (real_t)absstep) ? a : (real_t)absstep;   // src/epsilons.jl:26-29,50-53
        if (!central) e = e * (real_t)dir;
        eps[0] = e;
        s_e = e;
    }
    __syncthreads();
    const real_t e = s_e;
    for (int64_t i = threadIdx.x; i < n; i += kJvpSmallBlock) {
        const real_t xi = x[i], ev = e * v[i];
        if (central) {
            X[i] = xi - ev;
            X[ld + i] = xi + ev;
        } else {
            X[i] = xi + ev;
            if (base_row >= 0) X[(int64_t)base_row * ld + i] = xi;
        }
    }
}

template <bool VEC>
__global__ void __launch_bounds__(kBlock)
k_jvp_diff(const real_t *__restrict__ a, const real_t *__restrict__ b, const real_t *__restrict__ eps, int central,
           int64_t m, real_t *__restrict__ out)
{
    const real_t e = central ? 2 * eps[0] : eps[0];
    if (VEC) {
        const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 2;
        if (i + 1 < m) {
            const d2j_t av = *reinterpret_cast<const d2j_t *>(a + i), bv = *reinterpret_cast<const d2j_t *>(b + i);
            *reinterpret_cast<d2j_t *>(out + i) = d2j_t{(av.x - bv.x) / e, (av.y - bv.y) / e};
        } else if (i < m) {
            out[i] = (a[i] - b[i]) / e;
        }
        return;
    }
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += stride) out[i] = (a[i] - b[i]) / e;
}

}  // namespace fdjac

struct fd_jvp_plan {
    fd_ctx *ctx = nullptr;
    int fdtype = 0;
    int64_t M = 0, N = 0, ldx = 0, ldf = 0;
    fdjac::real_t *d_X = nullptr, *d_FX = nullptr, *d_fx = nullptr, *d_eps = nullptr;
    double *d_partial = nullptr;
    fdjac::real_t *d_xs = nullptr, *d_vs = nullptr, *d_fin = nullptr, *d_out = nullptr;
    int nparts = 1;
    bool small_ok = true;          // fused single-workgroup launch of small problems (FDJAC_SMALL, read at plan creation)
    fd_f_launch_lazy_jvp lazy_fn = nullptr;
    int lazy_caps = 0;             // FD_LAZY_JVP_CAP_* of lazy_fn
    bool lazy_diff = true;         // ask a FD_LAZY_JVP_CAP_QUOTIENT launcher for the finished quotient (FDJAC_LAZY_DIFF=0: never)
};

using namespace fdjac;

extern "C" {

int fd_jvp_plan_create(fd_ctx *ctx, int64_t M, int64_t N, int fdtype, fd_jvp_plan **out)
{
    FD_REQUIRE(ctx && out, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(M >= 1 && N >= 1, FD_ERR_SHAPE, "bad shape");
    if (fdtype != FD_FORWARD && fdtype != FD_CENTRAL) {
        // src/jvp.jl:248-250, :270-271
        set_error("finite_difference_jvp doesn't support :complex-mode finite diff");
        return FD_ERR_UNSUPPORTED;
    }
    FD_HIP_CHECK(hipSetDevice(ctx->device));
    fd_jvp_plan *p = new (std::nothrow) fd_jvp_plan();
    FD_REQUIRE(p, FD_ERR_NOMEM, "out of host memory");
    p->ctx = ctx; p->fdtype = fdtype; p->M = M; p->N = N;
    { const char *e = fdjac::test_switch("FDJAC_SMALL"); p->small_ok = !(e && *e && atoi(e) == 0); }
    { const char *e = fdjac::test_switch("FDJAC_LAZY_DIFF"); p->lazy_diff = !(e && *e && atoi(e) == 0); }
    p->ldx = (N + 31) / 32 * 32; p->ldf = (M + 31) / 32 * 32;
    p->nparts = balanced_grid((N + kBlock - 1) / kBlock, (int64_t)ctx->num_cus * 8);
    const int pts = fdtype == FD_CENTRAL ? 2 : 1;
    void **slots[] = {(void **)&p->d_X, (void **)&p->d_FX, (void **)&p->d_fx, (void **)&p->d_eps, (void **)&p->d_partial,
                      (void **)&p->d_xs, (void **)&p->d_vs, (void **)&p->d_fin, (void **)&p->d_out};
    // (+1 row of X / FX: small forward problems evaluate f(x) as a second member of the batch)
    const int64_t sizes[] = {(pts + 1) * p->ldx, (pts + 1) * p->ldf, p->ldf, 1, p->nparts, p->ldx, p->ldx, p->ldf, p->ldf};
    for (int k = 0; k < 9; ++k)
        if (hipMalloc(slots[k], sizeof(double) * (size_t)sizes[k]) != hipSuccess) {   // (partials are Float64)
            set_error("hipMalloc failed in fd_jvp_plan_create");
            for (int q = 0; q < k; ++q) (void)hipFree(*slots[q]);
            delete p;
            return FD_ERR_NOMEM;
        }
    *out = p;
    return FD_OK;
}

int fd_jvp_plan_destroy(fd_jvp_plan *p)
{
    if (!p) return FD_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    void *ptrs[] = {p->d_X, p->d_FX, p->d_fx, p->d_eps, p->d_partial, p->d_xs, p->d_vs, p->d_fin, p->d_out};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete p;
    return FD_OK;
}

static int jvp_enqueue(fd_jvp_plan *p, fd_f_launch f, void *fctx, const real_t *xd, const real_t *vd, const real_t *fin,
                       double relstep, double absstep, double dir, real_t *out)
{
    hipStream_t s = p->ctx->stream;
    const int central = p->fdtype == FD_CENTRAL;
    if (!(relstep > 0)) {   // default_relstep(fdtype, eltype(x)), src/epsilons.jl:133-144
        const real_t e = std::numeric_limits<real_t>::epsilon();
        relstep = central ? (double)std::cbrt(e) : (double)std::sqrt(e);
    }
    if (absstep < 0) absstep = relstep;
    const int g = balanced_grid((p->N + kBlock - 1) / kBlock, (int64_t)p->ctx->num_cus * 8);
    const int gm = balanced_grid((p->M + kBlock - 1) / kBlock, (int64_t)p->ctx->num_cus * 8);
    const bool small = p->small_ok && p->N <= kSmallN;   // (FDJAC_SMALL is read at plan creation)
    const bool base_in_batch = small && !central && !fin;
    const bool vx = ((((uintptr_t)xd) | ((uintptr_t)vd)) & kPairMask) == 0;
    // Lazy-point launcher (large problems; small ones are launch-latency bound and already fused): dot product and
    // step as usual, then ONE f! launch that perturbs while loading -- the points pass and, in the forward arm, the
    // separate f(x) launch disappear.  A launcher that declines falls through to the materialised points.
    if (!small && p->lazy_fn) {
        if (vx) hipLaunchKernelGGL(k_dot_partial<true>, dim3(p->nparts), dim3(kBlock), 0, s, xd, vd, p->N, p->d_partial);
        else hipLaunchKernelGGL(k_dot_partial<false>, dim3(p->nparts), dim3(kBlock), 0, s, xd, vd, p->N, p->d_partial);
        hipLaunchKernelGGL(k_jvp_eps, dim3(1), dim3(kBlock), 0, s, p->d_partial, p->nparts, relstep, absstep, dir,
                           central ? 0 : 1, p->d_eps);
        FD_HIP_CHECK(hipGetLastError());
        fd_lazy_jvp_points lp = {};
        lp.x = xd;
        lp.v = vd;
        lp.eps = p->d_eps;
        lp.base_out = (!central && !fin) ? p->d_fx : nullptr;
        lp.central = central;
        // a launcher that can, subtracts and divides itself and writes the JVP where the caller wants it: the call is
        // the step-size reduction + ONE f! launch (a caller's f_in is the subtrahend the reference uses: plain path then)
        const bool quotient = (p->lazy_caps & FD_LAZY_JVP_CAP_QUOTIENT) && p->lazy_diff && !fin && ((((uintptr_t)out) & kPairMask) == 0);
        if (quotient) { lp.quotient_out = out; lp.base_out = nullptr; }
        const int lrc = p->lazy_fn(fctx, p->d_FX, &lp, p->ldf, (void *)s);
        FD_REQUIRE(lrc == 0 || lrc == FD_LAZY_DECLINED, FD_ERR_CALLBACK, "lazy JVP f! launcher returned %d", lrc);
        if (lrc == 0 && quotient) return FD_OK;
        if (lrc == 0) {
            const real_t *la = central ? p->d_FX + p->ldf : p->d_FX;
            const real_t *lb = central ? p->d_FX : (fin ? fin : p->d_fx);
            if (((((uintptr_t)la) | ((uintptr_t)lb) | ((uintptr_t)out)) & kPairMask) == 0)
                hipLaunchKernelGGL(k_jvp_diff<true>, dim3((unsigned)((p->M + 2 * kBlock - 1) / (2 * kBlock))), dim3(kBlock), 0, s,
                                   la, lb, p->d_eps, central, p->M, out);
            else hipLaunchKernelGGL(k_jvp_diff<false>, dim3(gm), dim3(kBlock), 0, s, la, lb, p->d_eps, central, p->M, out);
            FD_HIP_CHECK(hipGetLastError());
            return FD_OK;
        }
        // declined: the step is already on the device, only the points remain to be written
        if (vx) hipLaunchKernelGGL(k_jvp_points<true>, dim3((unsigned)((p->N + 2 * kBlock - 1) / (2 * kBlock))), dim3(kBlock), 0, s,
                                   xd, vd, p->d_eps, central, p->N, p->d_X, p->ldx);
        else hipLaunchKernelGGL(k_jvp_points<false>, dim3(g), dim3(kBlock), 0, s, xd, vd, p->d_eps, central, p->N, p->d_X, p->ldx);
    } else if (small) {
        hipLaunchKernelGGL(k_jvp_small, dim3(1), dim3(kJvpSmallBlock), 0, s, xd, vd, p->N, relstep, absstep, dir, central,
                           base_in_batch ? 1 : -1, p->d_eps, p->d_X, p->ldx);
    } else {
        if (vx) hipLaunchKernelGGL(k_dot_partial<true>, dim3(p->nparts), dim3(kBlock), 0, s, xd, vd, p->N, p->d_partial);
        else hipLaunchKernelGGL(k_dot_partial<false>, dim3(p->nparts), dim3(kBlock), 0, s, xd, vd, p->N, p->d_partial);
        hipLaunchKernelGGL(k_jvp_eps, dim3(1), dim3(kBlock), 0, s, p->d_partial, p->nparts, relstep, absstep, dir,
                           central ? 0 : 1, p->d_eps);
        if (vx) hipLaunchKernelGGL(k_jvp_points<true>, dim3((unsigned)((p->N + 2 * kBlock - 1) / (2 * kBlock))), dim3(kBlock), 0, s,
                                   xd, vd, p->d_eps, central, p->N, p->d_X, p->ldx);
        else hipLaunchKernelGGL(k_jvp_points<false>, dim3(g), dim3(kBlock), 0, s, xd, vd, p->d_eps, central, p->N, p->d_X, p->ldx);
    }
    FD_HIP_CHECK(hipGetLastError());
    const real_t *a, *b;
    int rc;
    if (central) {
        rc = f(fctx, p->d_FX, p->d_X, 2, p->ldx, p->ldf, 0, p->M, 0, (void *)s);
        FD_REQUIRE(rc == 0, FD_ERR_CALLBACK, "f! launcher returned %d", rc);
        a = p->d_FX + p->ldf;  // f(x + eps v)
        b = p->d_FX;           // f(x - eps v)
    } else if (base_in_batch) {
        rc = f(fctx, p->d_FX, p->d_X, 2, p->ldx, p->ldf, 0, p->M, 0, (void *)s);   // points: x + eps v, x
        FD_REQUIRE(rc == 0, FD_ERR_CALLBACK, "f! launcher returned %d", rc);
        a = p->d_FX;
        b = p->d_FX + p->ldf;
    } else {
        if (fin) {
            b = fin;
        } else {
            rc = f(fctx, p->d_fx, xd, 1, p->N, p->ldf, 0, p->M, 0, (void *)s);
            FD_REQUIRE(rc == 0, FD_ERR_CALLBACK, "f! launcher returned %d", rc);
            b = p->d_fx;
        }
        rc = f(fctx, p->d_FX, p->d_X, 1, p->ldx, p->ldf, 0, p->M, 0, (void *)s);
        FD_REQUIRE(rc == 0, FD_ERR_CALLBACK, "f! launcher returned %d", rc);
        a = p->d_FX;
    }
    if (((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & kPairMask) == 0)
        hipLaunchKernelGGL(k_jvp_diff<true>, dim3((unsigned)((p->M + 2 * kBlock - 1) / (2 * kBlock))), dim3(kBlock), 0, s, a, b,
                           p->d_eps, central, p->M, out);
    else hipLaunchKernelGGL(k_jvp_diff<false>, dim3(gm), dim3(kBlock), 0, s, a, b, p->d_eps, central, p->M, out);
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

int fd_jvp(fd_jvp_plan *p, fd_f_launch f, void *fctx, const void *x, const void *v, int xv_kind, const void *f_in,
           int f_in_kind, double relstep, double absstep, double dir, void *jvp_out, int out_kind)
{
    FD_REQUIRE(p && f && x && v && jvp_out, FD_ERR_ARG, "NULL argument");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const real_t *xd = (const real_t *)x, *vd = (const real_t *)v, *fin = (const real_t *)f_in;
    if (xv_kind == FD_HOST) {
        FD_HIP_CHECK(hipMemcpyAsync(p->d_xs, x, sizeof(real_t) * (size_t)p->N, hipMemcpyHostToDevice, s));
        FD_HIP_CHECK(hipMemcpyAsync(p->d_vs, v, sizeof(real_t) * (size_t)p->N, hipMemcpyHostToDevice, s));
        xd = p->d_xs; vd = p->d_vs;
    }
    if (f_in && f_in_kind == FD_HOST) {
        FD_HIP_CHECK(hipMemcpyAsync(p->d_fin, f_in, sizeof(real_t) * (size_t)p->M, hipMemcpyHostToDevice, s));
        fin = p->d_fin;
    }
    real_t *out = out_kind == FD_DEVICE ? (real_t *)jvp_out : p->d_out;
    const int rc = jvp_enqueue(p, f, fctx, xd, vd, p->fdtype == FD_FORWARD ? fin : nullptr, relstep, absstep, dir, out);
    if (rc) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    if (out_kind == FD_HOST)
        FD_HIP_CHECK(hipMemcpyAsync(jvp_out, out, sizeof(real_t) * (size_t)p->M, hipMemcpyDeviceToHost, s));
    FD_HIP_CHECK(hipStreamSynchronize(s));
    return FD_OK;
}

int fd_jvp_async(fd_jvp_plan *p, fd_f_launch f, void *fctx, const void *x, const void *v, const void *f_in,
                 double relstep, double absstep, double dir, void *jvp_out)
{
    FD_REQUIRE(p && f && x && v && jvp_out, FD_ERR_ARG, "NULL argument");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    return jvp_enqueue(p, f, fctx, (const real_t *)x, (const real_t *)v,
                       p->fdtype == FD_FORWARD ? (const real_t *)f_in : nullptr, relstep, absstep, dir, (real_t *)jvp_out);
}

int fd_jvp_plan_set_lazy_f(fd_jvp_plan *p, fd_f_launch_lazy_jvp lazy)
{
    FD_REQUIRE(p != nullptr, FD_ERR_ARG, "plan is NULL");
    p->lazy_fn = lazy;
    p->lazy_caps = 0;
    return FD_OK;
}

int fd_jvp_plan_set_lazy_caps(fd_jvp_plan *p, int caps)
{
    FD_REQUIRE(p != nullptr, FD_ERR_ARG, "plan is NULL");
    FD_REQUIRE(p->lazy_fn != nullptr || caps == 0, FD_ERR_ARG, "no lazy launcher installed");
    p->lazy_caps = caps;
    return FD_OK;
}

int fd_jvp_get_epsilon(fd_jvp_plan *p, double *eps_out)
{
    FD_REQUIRE(p && eps_out, FD_ERR_ARG, "NULL argument");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    FD_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    real_t e = 0;
    FD_HIP_CHECK(hipMemcpy(&e, p->d_eps, sizeof(real_t), hipMemcpyDeviceToHost));
    *eps_out = (double)e;
    return FD_OK;
}

}  // extern "C"
