// Stream-copy ceiling exploration on MI355X: which launch geometry / load-store flavour gets closest to HBM peak.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double __attribute__((ext_vector_type(2))) d2;

template <int BS> __global__ void __launch_bounds__(BS) copy_gs(const d2* __restrict__ s, d2* __restrict__ d, int64_t n) {
    int64_t st = (int64_t)gridDim.x * BS;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += st) d[i] = s[i];
}
template <int BS, int U> __global__ void __launch_bounds__(BS) copy_gs_unroll(const d2* __restrict__ s, d2* __restrict__ d, int64_t n) {
    int64_t st = (int64_t)gridDim.x * BS;
    int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    for (; i + (U - 1) * st < n; i += U * st) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s[i + u * st];
#pragma unroll
        for (int u = 0; u < U; ++u) d[i + u * st] = v[u];
    }
    for (; i < n; i += st) d[i] = s[i];
}
template <int BS, int U> __global__ void __launch_bounds__(BS) copy_gs_unroll_nt(const d2* __restrict__ s, d2* __restrict__ d, int64_t n) {
    int64_t st = (int64_t)gridDim.x * BS;
    int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    for (; i + (U - 1) * st < n; i += U * st) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&s[i + u * st]);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], &d[i + u * st]);
    }
    for (; i < n; i += st) d[i] = s[i];
}
// block-contiguous tiles: each block copies TILE consecutive elements, U loads in flight
template <int BS, int U> __global__ void __launch_bounds__(BS) copy_tile(const d2* __restrict__ s, d2* __restrict__ d, int64_t n) {
    int64_t base = (int64_t)blockIdx.x * BS * U;
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { int64_t i = base + u * BS + threadIdx.x; if (i < n) v[u] = s[i]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { int64_t i = base + u * BS + threadIdx.x; if (i < n) d[i] = v[u]; }
}
template <int BS, int U> __global__ void __launch_bounds__(BS) copy_tile_nt(const d2* __restrict__ s, d2* __restrict__ d, int64_t n) {
    int64_t base = (int64_t)blockIdx.x * BS * U;
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { int64_t i = base + u * BS + threadIdx.x; if (i < n) v[u] = __builtin_nontemporal_load(&s[i]); }
#pragma unroll
    for (int u = 0; u < U; ++u) { int64_t i = base + u * BS + threadIdx.x; if (i < n) __builtin_nontemporal_store(v[u], &d[i]); }
}
// read-only and write-only ceilings
template <int BS, int U> __global__ void __launch_bounds__(BS) read_only(const d2* __restrict__ s, double* __restrict__ sink, int64_t n) {
    int64_t st = (int64_t)gridDim.x * BS;
    d2 acc = {0, 0};
    int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    for (; i + (U - 1) * st < n; i += U * st) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += s[i + u * st];
    }
    if (acc.x + acc.y == 1.2345) sink[0] = acc.x;
}
template <int BS> __global__ void __launch_bounds__(BS) write_only(d2* __restrict__ d, int64_t n) {
    int64_t st = (int64_t)gridDim.x * BS;
    d2 v = {1.0, 2.0};
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += st) d[i] = v;
}

int main(int argc, char** argv) {
    int64_t bytes = (argc > 1 ? atoll(argv[1]) : 1024) * (1ll << 20);
    int64_t n = bytes / 16;
    d2 *a, *b; double* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, double bytes_moved) {
        launch(); hipDeviceSynchronize();
        float best = 1e9, tot = 0;
        for (int r = 0; r < 10; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; tot += ms; }
        printf("%-34s best %.1f us  %.0f GB/s   avg %.0f GB/s\n", name, best * 1e3, bytes_moved / (best * 1e-3) / 1e9, bytes_moved * 10 / (tot * 1e-3) / 1e9);
    };
    double bm = 2.0 * bytes;
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64]; snprintf(nm, 64, "gs256 grid=%d", g);
        run(nm, [&] { hipLaunchKernelGGL(copy_gs<256>, dim3(g), dim3(256), 0, 0, a, b, n); }, bm);
    }
    run("gs256 unroll4 grid=2048", [&] { hipLaunchKernelGGL((copy_gs_unroll<256, 4>), dim3(2048), dim3(256), 0, 0, a, b, n); }, bm);
    run("gs256 unroll8 grid=2048", [&] { hipLaunchKernelGGL((copy_gs_unroll<256, 8>), dim3(2048), dim3(256), 0, 0, a, b, n); }, bm);
    run("gs256 unroll4 grid=4096", [&] { hipLaunchKernelGGL((copy_gs_unroll<256, 4>), dim3(4096), dim3(256), 0, 0, a, b, n); }, bm);
    run("gs256 unroll4 nt grid=2048", [&] { hipLaunchKernelGGL((copy_gs_unroll_nt<256, 4>), dim3(2048), dim3(256), 0, 0, a, b, n); }, bm);
    run("gs256 unroll8 nt grid=2048", [&] { hipLaunchKernelGGL((copy_gs_unroll_nt<256, 8>), dim3(2048), dim3(256), 0, 0, a, b, n); }, bm);
    run("gs512 unroll4 grid=1024", [&] { hipLaunchKernelGGL((copy_gs_unroll<512, 4>), dim3(1024), dim3(512), 0, 0, a, b, n); }, bm);
    run("gs1024 unroll4 grid=512", [&] { hipLaunchKernelGGL((copy_gs_unroll<1024, 4>), dim3(512), dim3(1024), 0, 0, a, b, n); }, bm);
    run("tile256x1 (1 elem/thread)", [&] { hipLaunchKernelGGL((copy_tile<256, 1>), dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n); }, bm);
    run("tile256x4", [&] { hipLaunchKernelGGL((copy_tile<256, 4>), dim3((n + 1023) / 1024), dim3(256), 0, 0, a, b, n); }, bm);
    run("tile256x8", [&] { hipLaunchKernelGGL((copy_tile<256, 8>), dim3((n + 2047) / 2048), dim3(256), 0, 0, a, b, n); }, bm);
    run("tile256x4 nt", [&] { hipLaunchKernelGGL((copy_tile_nt<256, 4>), dim3((n + 1023) / 1024), dim3(256), 0, 0, a, b, n); }, bm);
    run("tile256x8 nt", [&] { hipLaunchKernelGGL((copy_tile_nt<256, 8>), dim3((n + 2047) / 2048), dim3(256), 0, 0, a, b, n); }, bm);
    run("tile512x4 nt", [&] { hipLaunchKernelGGL((copy_tile_nt<512, 4>), dim3((n + 2047) / 2048), dim3(512), 0, 0, a, b, n); }, bm);
    run("read-only unroll4 grid=2048", [&] { hipLaunchKernelGGL((read_only<256, 4>), dim3(2048), dim3(256), 0, 0, a, sink, n); }, (double)bytes);
    run("read-only unroll8 grid=4096", [&] { hipLaunchKernelGGL((read_only<256, 8>), dim3(4096), dim3(256), 0, 0, a, sink, n); }, (double)bytes);
    run("write-only grid=2048", [&] { hipLaunchKernelGGL(write_only<256>, dim3(2048), dim3(256), 0, 0, b, n); }, (double)bytes);
    run("write-only grid=8192", [&] { hipLaunchKernelGGL(write_only<256>, dim3(8192), dim3(256), 0, 0, b, n); }, (double)bytes);
    run("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, bm);
    return 0;
}
