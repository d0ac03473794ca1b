// What does a 2:1 read:write streaming kernel reach on MI355X?  (the fused difference + decompression kernel
// moves 470 MB of reads and 240 MB of writes per launch at N = 1e7; the 1:1 stream copy reaches 6.2-6.3 TB/s)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double __attribute__((ext_vector_type(2))) d2;

// out = a + b : 2 reads, 1 write, one 16-B element per thread
__global__ void __launch_bounds__(256) add2(const d2* __restrict__ a, const d2* __restrict__ b, d2* __restrict__ o, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}
// tile variant: U elements per thread, block-contiguous
template <int U> __global__ void __launch_bounds__(256) add2_tile(const d2* __restrict__ a, const d2* __restrict__ b, d2* __restrict__ o, int64_t n) {
    int64_t base = (int64_t)blockIdx.x * 256 * U;
    d2 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { int64_t i = base + u * 256 + threadIdx.x; if (i < n) { va[u] = a[i]; vb[u] = b[i]; } }
#pragma unroll
    for (int u = 0; u < U; ++u) { int64_t i = base + u * 256 + threadIdx.x; if (i < n) o[i] = va[u] + vb[u]; }
}
// 4 reads (one shared by 3 outputs) -> 3 writes: the traffic shape of the tridiagonal decompression without indices
// o[3j+k] = (F_k[j] - f[j]) / e   (not the real row mapping; same bytes: 32 B read + 24 B written per j)
__global__ void __launch_bounds__(256) dec_shape(const d2* __restrict__ f, const d2* __restrict__ F0, const d2* __restrict__ F1,
                                                 const d2* __restrict__ F2, d2* __restrict__ o, int64_t n2, double e) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // pair of columns
    if (i >= n2) return;
    d2 b = f[i], a0 = F0[i], a1 = F1[i], a2 = F2[i];
    d2 q0 = (a0 - b) / e, q1 = (a1 - b) / e, q2 = (a2 - b) / e;
    // 6 outputs = 3 d2, contiguous 48 B per thread
    o[3 * i] = d2{q0.x, q1.x}; o[3 * i + 1] = d2{q2.x, q0.y}; o[3 * i + 2] = d2{q1.y, q2.y};
}
// same + the index/colour streams of the CSC kernel (12 B + 3 B per column... read as int4 + 2 bytes per pair of entries)
__global__ void __launch_bounds__(256) dec_shape_idx(const d2* __restrict__ f, const d2* __restrict__ F0, const d2* __restrict__ F1,
                                                     const d2* __restrict__ F2, const int2* __restrict__ rv, const uint16_t* __restrict__ cv,
                                                     d2* __restrict__ o, int64_t n2, double e) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    d2 b = f[i], a0 = F0[i], a1 = F1[i], a2 = F2[i];
    int2 r0 = rv[3 * i], r1 = rv[3 * i + 1], r2 = rv[3 * i + 2];
    unsigned c0 = cv[3 * i], c1 = cv[3 * i + 1], c2 = cv[3 * i + 2];
    double s = (double)((r0.x ^ r1.x ^ r2.x ^ r0.y ^ r1.y ^ r2.y) & 1) + (double)((c0 ^ c1 ^ c2) & 1);   // 0 for our fill
    d2 q0 = (a0 - b) / e, q1 = (a1 - b) / e, q2 = (a2 - b) / e;
    o[3 * i] = d2{q0.x + s, q1.x}; o[3 * i + 1] = d2{q2.x, q0.y}; o[3 * i + 2] = d2{q1.y, q2.y};
}
__global__ void __launch_bounds__(256) rd_only(const d2* __restrict__ a, double* __restrict__ sink, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    d2 v = i < n ? a[i] : d2{0, 0};
    if (v.x == 1.2345e300) sink[0] = v.y;
}
__global__ void __launch_bounds__(256) rd_gs(const d2* __restrict__ a, double* __restrict__ sink, int64_t n) {
    d2 acc = {0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += a[i];
    if (acc.x == 1.2345e300) sink[0] = acc.y;
}
__global__ void __launch_bounds__(256) wr_only(d2* __restrict__ o, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = d2{1.0, 2.0};
}

template <typename F> static float timeit(F launch, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main() {
    const int64_t N = 10000000, n2 = N / 2;           // columns, pairs of columns
    const int64_t nout2 = 3 * n2;                     // d2 elements of the output (240 MB)
    d2 *f, *F0, *F1, *F2, *o, *big; int2* rv; uint16_t* cv; double* sink;
    CK(hipMalloc(&f, N * 8)); CK(hipMalloc(&F0, N * 8)); CK(hipMalloc(&F1, N * 8)); CK(hipMalloc(&F2, N * 8));
    CK(hipMalloc(&o, nout2 * 16)); CK(hipMalloc(&big, nout2 * 16 * 2)); CK(hipMalloc(&rv, nout2 * 8)); CK(hipMalloc(&cv, nout2 * 2));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(f, 0, N * 8)); CK(hipMemset(F0, 0, N * 8)); CK(hipMemset(F1, 0, N * 8)); CK(hipMemset(F2, 0, N * 8));
    CK(hipMemset(big, 0, nout2 * 32)); CK(hipMemset(rv, 0, nout2 * 8)); CK(hipMemset(cv, 0, nout2 * 2));
    const int it = 20;
    float ms;
    ms = timeit([&] { add2<<<(nout2 + 255) / 256, 256>>>(big, big + nout2, o, nout2); }, it);
    printf("add2 (480 MB rd + 240 MB wr)         %8.2f us  %7.1f GB/s\n", ms * 1e3, 3.0 * nout2 * 16 / ms / 1e6);
    ms = timeit([&] { add2_tile<2><<<(nout2 + 511) / 512, 256>>>(big, big + nout2, o, nout2); }, it);
    printf("add2_tile<2>                         %8.2f us  %7.1f GB/s\n", ms * 1e3, 3.0 * nout2 * 16 / ms / 1e6);
    ms = timeit([&] { add2_tile<4><<<(nout2 + 1023) / 1024, 256>>>(big, big + nout2, o, nout2); }, it);
    printf("add2_tile<4>                         %8.2f us  %7.1f GB/s\n", ms * 1e3, 3.0 * nout2 * 16 / ms / 1e6);
    ms = timeit([&] { dec_shape<<<(n2 + 255) / 256, 256>>>(f, F0, F1, F2, o, n2, 3e-7); }, it);
    printf("dec_shape (320 MB rd + 240 MB wr)    %8.2f us  %7.1f GB/s (56 B/col)\n", ms * 1e3, 56.0 * N / ms / 1e6);
    ms = timeit([&] { dec_shape_idx<<<(n2 + 255) / 256, 256>>>(f, F0, F1, F2, rv, cv, o, n2, 3e-7); }, it);
    printf("dec_shape_idx (470 MB rd + 240 MB wr)%8.2f us  %7.1f GB/s (71 B/col)\n", ms * 1e3, 71.0 * N / ms / 1e6);
    ms = timeit([&] { rd_only<<<(nout2 * 2 + 255) / 256, 256>>>(big, sink, nout2 * 2); }, it);
    printf("read only 480 MB                     %8.2f us  %7.1f GB/s\n", ms * 1e3, 2.0 * nout2 * 16 / ms / 1e6);
    ms = timeit([&] { wr_only<<<(nout2 + 255) / 256, 256>>>(o, nout2); }, it);
    printf("write only 240 MB                    %8.2f us  %7.1f GB/s\n", ms * 1e3, 1.0 * nout2 * 16 / ms / 1e6);
    for (int64_t mb : {20, 40, 80, 160, 320}) {
        const int64_t n16 = mb * 1000000 / 16;
        ms = timeit([&] { rd_only<<<(n16 + 255) / 256, 256>>>(big, sink, n16); }, 50);
        printf("read only %4lld MB (1 elt/thread)        %8.2f us  %7.1f GB/s\n", (long long)mb, ms * 1e3, n16 * 16.0 / ms / 1e6);
        ms = timeit([&] { rd_gs<<<2048, 256>>>(big, sink, n16); }, 50);
        printf("read only %4lld MB (grid-stride 2048)    %8.2f us  %7.1f GB/s\n", (long long)mb, ms * 1e3, n16 * 16.0 / ms / 1e6);
    }
    ms = timeit([&] { add2<<<(nout2 + 255) / 256, 256>>>(big, big + nout2, o, nout2); }, it);
    printf("add2 again                           %8.2f us  %7.1f GB/s\n", ms * 1e3, 3.0 * nout2 * 16 / ms / 1e6);
    return 0;
}
