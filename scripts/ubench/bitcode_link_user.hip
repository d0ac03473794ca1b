#include <hip/hip_runtime.h>
struct fd_cpoint { int kind; const void *obj; };
extern "C" __device__ double fdjac_point_get(const fd_cpoint *X, long long j);
extern "C" __device__ double fdjac_user_row(const void *params, long long i, const fd_cpoint *X)
{
    const long long n = *(const long long *)params;
    const double xi = fdjac_point_get(X, i), xm = fdjac_point_get(X, i > 0 ? i - 1 : i), xp = fdjac_point_get(X, i + 1 < n ? i + 1 : i);
    const double a = i > 0 ? xm : 0.0, b = i + 1 < n ? xp : 0.0;
    double v = (a - 2.0 * xi) + b;
    v = v + (xi * xi) * b;
    return v;
}
