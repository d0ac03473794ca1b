// A row function as a caller WITHOUT the library's C++ header would write it (tests/test_gpu_jit.py compiles this file to LLVM bitcode
// with `hipcc -fgpu-rdc -emit-llvm --offload-device-only -c -DREAL=double|float` and hands the bitcode to fd_f_link_rows_bitcode):
// the tridiag_nl fixture, (x[i-1] - 2 x[i]) + x[i+1] + (x[i] x[i]) x[i+1], on the real and on the complex point.
// Only the C interface of include/fdjac.h ("a row function given as LLVM BITCODE") is used.
#include <hip/hip_runtime.h>
#ifndef REAL
#define REAL double
#endif
typedef REAL real_t;
struct fd_cpoint { int kind; const void *obj; };
extern "C" __device__ real_t fdjac_point_get(const fd_cpoint *X, long long j);
extern "C" __device__ void fdjac_point_get_c(const fd_cpoint *X, long long j, real_t *re_im);

extern "C" __device__ real_t fdjac_user_row(const void *params, long long i, const fd_cpoint *X)
{
    const long long n = *(const long long *)params;
    const real_t xi = fdjac_point_get(X, i), xm = fdjac_point_get(X, i > 0 ? i - 1 : i), xp = fdjac_point_get(X, i + 1 < n ? i + 1 : i);
    const real_t a = i > 0 ? xm : (real_t)0, b = i + 1 < n ? xp : (real_t)0;
    real_t v = (a - (real_t)2 * xi) + b;
    v = v + (xi * xi) * b;
    return v;
}

struct cx { real_t re, im; };
static __device__ cx cadd(cx a, cx b) { return {a.re + b.re, a.im + b.im}; }
static __device__ cx csub(cx a, cx b) { return {a.re - b.re, a.im - b.im}; }
static __device__ cx cmul(cx a, cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
static __device__ cx cscale(real_t s, cx a) { return {s * a.re, s * a.im}; }
extern "C" __device__ void fdjac_user_row_c(const void *params, long long i, const fd_cpoint *X, real_t *out)
{
    const long long n = *(const long long *)params;
    real_t t[2];
    fdjac_point_get_c(X, i, t);
    const cx xi = {t[0], t[1]};
    fdjac_point_get_c(X, i > 0 ? i - 1 : i, t);
    const cx xm = {t[0], t[1]};
    fdjac_point_get_c(X, i + 1 < n ? i + 1 : i, t);
    const cx xp = {t[0], t[1]};
    const cx z = {0, 0}, a = i > 0 ? xm : z, b = i + 1 < n ? xp : z;
    cx v = cadd(csub(a, cscale((real_t)2, xi)), b);
    v = cadd(v, cmul(cmul(xi, xi), b));
    out[0] = v.re;
    out[1] = v.im;
}
