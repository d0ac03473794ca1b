// Does a producer->consumer pair of streaming kernels run faster when the hand-off buffer fits the 256 MiB
// Infinity Cache?  chain: k1: B = A ; k2: C = B.  Reports per-kernel time for several buffer sizes, and the
// effect of traversing k2 in reverse order (most-recently-written lines first).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double __attribute__((ext_vector_type(2))) d2;
template <bool REV> __global__ void __launch_bounds__(256) cp(const d2* __restrict__ s, d2* __restrict__ d, int64_t n) {
    int64_t blk = REV ? (int64_t)gridDim.x - 1 - blockIdx.x : blockIdx.x;
    int64_t i = blk * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
int main() {
    const int64_t maxb = 1ll << 30;
    d2 *A, *B, *C, *junk;
    hipMalloc(&A, maxb); hipMalloc(&B, maxb); hipMalloc(&C, maxb); hipMalloc(&junk, maxb);
    hipMemset(A, 1, maxb); hipMemset(junk, 1, maxb);
    hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
    for (int64_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024}) {
        int64_t n = mb * (1ll << 20) / 16; int g = (int)((n + 255) / 256);
        for (int rev = 0; rev < 2; ++rev) {
            float t1 = 0, t2 = 0; int reps = 10;
            for (int r = 0; r < reps; ++r) {
                // flush the cache with an unrelated 1 GiB read/write
                hipLaunchKernelGGL(cp<false>, dim3((int)((maxb / 16 + 255) / 256)), dim3(256), 0, 0, junk, C, maxb / 16);
                hipEventRecord(e[0]);
                hipLaunchKernelGGL(cp<false>, dim3(g), dim3(256), 0, 0, A, B, n);
                hipEventRecord(e[1]);
                if (rev) hipLaunchKernelGGL(cp<true>, dim3(g), dim3(256), 0, 0, B, C, n);
                else hipLaunchKernelGGL(cp<false>, dim3(g), dim3(256), 0, 0, B, C, n);
                hipEventRecord(e[2]);
                hipEventSynchronize(e[2]);
                float a, b; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]); t1 += a; t2 += b;
            }
            double by = 2.0 * n * 16;
            printf("%5lld MB %s  producer %7.1f us %6.0f GB/s | consumer %7.1f us %6.0f GB/s\n", (long long)mb, rev ? "rev" : "fwd",
                   t1 / reps * 1e3, by / (t1 / reps * 1e-3) / 1e9, t2 / reps * 1e3, by / (t2 / reps * 1e-3) / 1e9);
        }
    }
    return 0;
}
