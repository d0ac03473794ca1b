// Why does the masked sum-of-squares pass run at ~3.5 TB/s?  Variants of the reduction kernel on N=1e7.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int BS = 256;
__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
template <int NC, int U, bool COLORS, bool REDUCE>
__global__ void __launch_bounds__(BS) k(const double* __restrict__ x, const uint8_t* __restrict__ color, int64_t n, double* __restrict__ partial) {
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    const int64_t tile = (int64_t)U * BS * 2;
    for (int64_t base = (int64_t)blockIdx.x * tile; base < n; base += (int64_t)gridDim.x * tile) {
        double2 v[U]; int c0[U], c1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + (int64_t)u * BS * 2 + threadIdx.x * 2;
            if (i + 1 < n) {
                v[u] = *reinterpret_cast<const double2*>(x + i);
                if (COLORS) { unsigned cc = *reinterpret_cast<const uint16_t*>(color + i); c0[u] = cc & 0xFF; c1[u] = cc >> 8; }
                else { c0[u] = (int)(i % 3); c1[u] = (int)((i + 1) % 3); }
            } else { v[u] = make_double2(0, 0); c0[u] = c1[u] = -2; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double s0 = v[u].x * v[u].x, s1 = v[u].y * v[u].y;
#pragma unroll
            for (int c = 0; c < NC; ++c) { acc[c] += (c0[u] == c) ? s0 : 0.0; acc[c] += (c1[u] == c) ? s1 : 0.0; }
        }
    }
    if (REDUCE) {
        __shared__ double red[BS / 64][NC];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < NC; ++c) { const double s = wave_sum(acc[c]); if (lane == 0) red[wave][c] = s; }
        __syncthreads();
        if (threadIdx.x < NC) { double s = 0; for (int w = 0; w < BS / 64; ++w) s += red[w][threadIdx.x]; partial[(int64_t)blockIdx.x * 8 + threadIdx.x] = s; }
    } else {
        double s = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) s += acc[c];
        if (s == 1.2345) partial[0] = s;
    }
}
// colours packed 8 per 8-byte word: thread handles 8 consecutive elements
template <int NC>
__global__ void __launch_bounds__(BS) k8(const double* __restrict__ x, const uint8_t* __restrict__ color, int64_t n, double* __restrict__ partial) {
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    for (int64_t i = ((int64_t)blockIdx.x * BS + threadIdx.x) * 8; i + 7 < n; i += (int64_t)gridDim.x * BS * 8) {
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const double2*>(x + i + 2 * u);
        const uint64_t cc = *reinterpret_cast<const uint64_t*>(color + i);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c0 = (int)((cc >> (16 * u)) & 0xFF), c1 = (int)((cc >> (16 * u + 8)) & 0xFF);
            const double s0 = v[u].x * v[u].x, s1 = v[u].y * v[u].y;
#pragma unroll
            for (int c = 0; c < NC; ++c) { acc[c] += (c0 == c) ? s0 : 0.0; acc[c] += (c1 == c) ? s1 : 0.0; }
        }
    }
    __shared__ double red[BS / 64][NC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NC; ++c) { const double s = wave_sum(acc[c]); if (lane == 0) red[wave][c] = s; }
    __syncthreads();
    if (threadIdx.x < NC) { double s = 0; for (int w = 0; w < BS / 64; ++w) s += red[w][threadIdx.x]; partial[(int64_t)blockIdx.x * 8 + threadIdx.x] = s; }
}
int main() {
    const int64_t n = 10000000;
    double *x, *partial, *junk; uint8_t* col;
    hipMalloc(&x, n * 8); hipMalloc(&col, n); hipMalloc(&partial, 8 * 8 * 65536); hipMalloc(&junk, 1ll << 30);
    std::vector<double> hx(n); std::vector<uint8_t> hc(n);
    for (int64_t i = 0; i < n; ++i) { hx[i] = (i % 1000) * 1e-3; hc[i] = i % 3; }
    hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(col, hc.data(), n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        float tot = 0, best = 1e9;
        for (int r = 0; r < 12; ++r) {
            hipMemsetAsync(junk, r, 1ll << 30, 0);   // evict x from the caches
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) { tot += ms; best = ms < best ? ms : best; }
        }
        printf("%-44s avg %.1f us best %.1f us  (%.0f GB/s)\n", name, tot / 10 * 1e3, best * 1e3, 9.0 * n / (best * 1e-3) / 1e9);
    };
    for (int g : {512, 1024, 1628, 2048, 2442, 4883}) {
        char nm[80]; snprintf(nm, 80, "NC4 U4 colours reduce grid=%d", g);
        run(nm, [&] { hipLaunchKernelGGL((k<4, 4, true, true>), dim3(g), dim3(BS), 0, 0, x, col, n, partial); });
    }
    run("NC4 U4 no-colour-loads reduce grid=1628", [&] { hipLaunchKernelGGL((k<4, 4, false, true>), dim3(1628), dim3(BS), 0, 0, x, col, n, partial); });
    run("NC4 U4 colours no-reduce grid=1628", [&] { hipLaunchKernelGGL((k<4, 4, true, false>), dim3(1628), dim3(BS), 0, 0, x, col, n, partial); });
    run("NC1 U4 colours reduce grid=1628", [&] { hipLaunchKernelGGL((k<1, 4, true, true>), dim3(1628), dim3(BS), 0, 0, x, col, n, partial); });
    run("NC4 U1 colours reduce grid=19532", [&] { hipLaunchKernelGGL((k<4, 1, true, true>), dim3(19532), dim3(BS), 0, 0, x, col, n, partial); });
    run("NC4 U2 colours reduce grid=9766", [&] { hipLaunchKernelGGL((k<4, 2, true, true>), dim3(9766), dim3(BS), 0, 0, x, col, n, partial); });
    run("NC4 U2 colours reduce grid=2048", [&] { hipLaunchKernelGGL((k<4, 2, true, true>), dim3(2048), dim3(BS), 0, 0, x, col, n, partial); });
    run("NC4 U8 colours reduce grid=1221", [&] { hipLaunchKernelGGL((k<4, 8, true, true>), dim3(1221), dim3(BS), 0, 0, x, col, n, partial); });
    run("k8 (8 consecutive/thread) grid=2048", [&] { hipLaunchKernelGGL((k8<4>), dim3(2048), dim3(BS), 0, 0, x, col, n, partial); });
    run("k8 (8 consecutive/thread) grid=4883", [&] { hipLaunchKernelGGL((k8<4>), dim3(4883), dim3(BS), 0, 0, x, col, n, partial); });
    return 0;
}
