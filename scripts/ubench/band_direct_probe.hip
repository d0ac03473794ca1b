// Is a register-only, computed-index kernel faster than the LDS-staged row-window kernel for the most regular case
// (tridiagonal CSC, cyclic colours, differences handed over by the lazy launcher: ONE array per colour)?
//   out[p] = D[c(p)][r(p)] / eps[c(p)],   p = 3j - 1 + k  ->  j = (p+1)/3, k = (p+1) - 3j, r = j - 1 + k, c = j % 3
// Every D value is read exactly once (a permutation): 240 MB in, 240 MB out at N = 1e7.
//   direct<U>: each thread produces U independent pairs of consecutive entries (8-B gathers, 16-B stores)
//   linear   : out[i] = D[i] / eps[0] over the same byte volume (16-B loads / stores) -- the speed of light of the mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int BS = 256;
typedef double __attribute__((ext_vector_type(2))) d2;

template <int U>
__global__ void __launch_bounds__(BS) k_direct(const double* __restrict__ D, int64_t ld, const double* __restrict__ eps,
                                                double* __restrict__ out, int64_t nnz) {
    const double e0 = eps[0], e1 = eps[1], e2 = eps[2];
    const int64_t base = (int64_t)blockIdx.x * (BS * 2 * U) + threadIdx.x * 2;
    double a[U][2]; int c[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t p = base + (int64_t)u * (BS * 2) + h;
            const uint32_t P = (uint32_t)(p + 1);
            const uint32_t j = P / 3u, k = P - 3u * j;
            const uint32_t jj = j / 3u;
            c[u][h] = (int)(j - 3u * jj);
            const int64_t r = (int64_t)j - 1 + k;
            a[u][h] = p < nnz ? D[(int64_t)c[u][h] * ld + r] : 0.0;
        }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t p = base + (int64_t)u * (BS * 2);
        d2 q;
        q.x = a[u][0] / (c[u][0] == 0 ? e0 : c[u][0] == 1 ? e1 : e2);
        q.y = a[u][1] / (c[u][1] == 0 ? e0 : c[u][1] == 1 ? e1 : e2);
        if (p + 1 < nnz) *reinterpret_cast<d2*>(out + p) = q;
        else if (p < nnz) out[p] = q.x;
    }
}

// the library's form: runtime w / C through 2^40 magic multipliers, step sizes by ds_bpermute, optional XCD-chunked and
// reversed tile order
__device__ inline int64_t xcd_tile(int64_t block, int64_t ntiles) { return (block & 7) * ((ntiles + 7) / 8) + (block >> 3); }
template <bool XCD, bool REV, bool SHFL>
__global__ void __launch_bounds__(BS) k_direct_rt(const double* __restrict__ D, int64_t ld, const double* __restrict__ eps,
                                                   double* __restrict__ out, int64_t nnz, int64_t off, int w, int u, int C, int shift,
                                                   uint64_t mw, uint64_t mc, int64_t M) {
    const int64_t ntiles = (nnz + 2 * BS - 1) / (2 * BS);
    int64_t xt = XCD ? xcd_tile(blockIdx.x, ntiles) : blockIdx.x;
    if (xt >= ntiles) return;
    const int64_t tile = REV ? ntiles - 1 - xt : xt;
    const int lane = threadIdx.x & 63;
    const double my_eps = lane < C ? eps[lane] : 1.0;
    const double e0 = eps[0], e1 = eps[1], e2 = eps[2];
    const int64_t p = tile * (2 * BS) + 2 * (int64_t)threadIdx.x;
    double q[2]; bool wr[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t Q = (uint32_t)(p + h + off);
        const uint32_t j = (uint32_t)(((uint64_t)Q * mw) >> 40), k = Q - j * (uint32_t)w;
        const uint32_t cj = j + (uint32_t)shift;
        const int c = (int)(cj - (uint32_t)(((uint64_t)cj * mc) >> 40) * (uint32_t)C);
        const int64_t r = (int64_t)j - u + k;
        const bool live = p + h < nnz;
        const bool inside = live & (r >= 0) & (r < M);
        const int64_t at = inside ? (int64_t)c * ld + r : 0;
        const double a = D[at];
        const double e = SHFL ? __shfl(my_eps, c, 64) : (c == 0 ? e0 : c == 1 ? e1 : e2);
        q[h] = inside ? a / e : 0.0;
        wr[h] = live;
    }
    if (wr[0] & wr[1]) *reinterpret_cast<d2*>(out + p) = d2{q[0], q[1]};
    else if (wr[0]) out[p] = q[0];
}

__global__ void __launch_bounds__(BS) k_linear(const double* __restrict__ D, const double* __restrict__ eps, double* __restrict__ out, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * BS + threadIdx.x) * 2;
    if (i + 1 >= n) return;
    const double e = eps[0];
    const d2 v = *reinterpret_cast<const d2*>(D + i);
    *reinterpret_cast<d2*>(out + i) = d2{v.x / e, v.y / e};
}

template <class F> static float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    const int64_t N = 10000000, ld = (N + 31) / 32 * 32, nnz = 3 * N - 2;
    double *D, *eps, *out;
    hipMalloc(&D, 3 * ld * 8 + 64); hipMalloc(&eps, 64); hipMalloc(&out, (nnz + 2) * 8);
    hipMemset(D, 0, 3 * ld * 8 + 64);
    const double he[3] = {1e-7, 2e-7, 3e-7};
    hipMemcpy(eps, he, sizeof he, hipMemcpyHostToDevice);
    const int reps = 30;
#define RUN(U) { const unsigned g = (unsigned)((nnz + BS * 2 * U - 1) / (BS * 2 * U)); \
    float us = timeit([&] { hipLaunchKernelGGL(k_direct<U>, dim3(g), dim3(BS), 0, 0, D, ld, eps, out, nnz); }, reps); \
    printf("direct U=%d  %7.1f us  %6.0f GB/s (480 MB)\n", U, us, 2.0 * nnz * 8 / us * 1e-3); }
    RUN(1) RUN(2) RUN(4) RUN(8)
    {
        const unsigned g = (unsigned)((nnz / 2 + BS - 1) / BS);
        float us = timeit([&] { hipLaunchKernelGGL(k_linear, dim3(g), dim3(BS), 0, 0, D, eps, out, nnz); }, reps);
        printf("linear       %7.1f us  %6.0f GB/s (480 MB)\n", us, 2.0 * nnz * 8 / us * 1e-3);
    }

#define RUNRT(X, R, S) { const int64_t nt = (nnz + 2 * BS - 1) / (2 * BS); const unsigned g = (unsigned)(X ? 8 * ((nt + 7) / 8) : nt); \
    const uint64_t mw = (((uint64_t)1 << 40) + 2) / 3, mc = mw; \
    float us = timeit([&] { hipLaunchKernelGGL((k_direct_rt<X, R, S>), dim3(g), dim3(BS), 0, 0, D, ld, eps, out, nnz, (int64_t)1, 3, 1, 3, 0, mw, mc, N); }, reps); \
    printf("direct rt xcd=%d rev=%d shfl=%d  %7.1f us  %6.0f GB/s\n", (int)X, (int)R, (int)S, us, 2.0 * nnz * 8 / us * 1e-3); }
    RUNRT(false, false, false) RUNRT(false, false, true) RUNRT(true, false, false) RUNRT(false, true, false) RUNRT(true, true, true)
    // correctness spot check of the index map
    double* hD = (double*)malloc(3 * ld * 8);
    for (int c = 0; c < 3; ++c) for (int64_t r = 0; r < N; ++r) hD[c * ld + r] = (c + 1) * 1e-7 * (double)(r + 1);
    hipMemcpy(D, hD, 3 * ld * 8, hipMemcpyHostToDevice);
    { const unsigned g = (unsigned)((nnz + BS * 2 * 2 - 1) / (BS * 2 * 2)); hipLaunchKernelGGL(k_direct<2>, dim3(g), dim3(BS), 0, 0, D, ld, eps, out, nnz); }
    double* ho = (double*)malloc(nnz * 8);
    hipMemcpy(ho, out, nnz * 8, hipMemcpyDeviceToHost);
    int64_t bad = 0;
    for (int64_t j = 0; j < N; ++j) for (int k = 0; k < 3; ++k) {
        const int64_t r = j - 1 + k; if (r < 0 || r >= N) continue;
        const int64_t p = 3 * j - 1 + k; const int c = (int)(j % 3);
        if (ho[p] != hD[c * ld + r] / he[c]) ++bad;
    }
    printf("index map mismatches: %lld\n", (long long)bad);
    return 0;
}
