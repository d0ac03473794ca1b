// What bounds the masked sum-of-squares pass (90 MB read in ~26 us at N = 1e7)?  Hypothesis: in the steady-state
// loop the pass follows 240 MB of nzval stores; those lines sit dirty in the 256 MiB Infinity Cache and every line the
// pass reads evicts one -> the pass moves 90 MB of reads PLUS ~90 MB of write-backs.  This probe times the same
// reduction kernel after (a) a dirty flush (1 GiB memset-like store kernel), (b) a clean flush (1 GiB read-only
// kernel), (c) a 240 MB store kernel that itself follows a clean flush, (d) nothing (x resident), with normal and
// non-temporal loads, with and without colour loads, and for several grids.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int BS = 256;
typedef double __attribute__((ext_vector_type(2))) d2;
__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// MODE 0: colours loaded (u8 pairs); 1: colours computed (cyclic, C = 3); NT: non-temporal loads of x
template <int NC, int U, int MODE, bool NT>
__global__ void __launch_bounds__(BS) k_eps(const double* __restrict__ x, const uint8_t* __restrict__ color, int64_t n,
                                            double* __restrict__ partial, int contiguous) {
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    const int64_t tile = (int64_t)U * BS * 2;
    const int64_t ntiles = (n + tile - 1) / tile;
    // contiguous: block b owns tiles [b*per, (b+1)*per); else grid-stride
    const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    int64_t t = contiguous ? (int64_t)blockIdx.x * per : blockIdx.x;
    const int64_t tend = contiguous ? (t + per < ntiles ? t + per : ntiles) : ntiles;
    const int64_t tstep = contiguous ? 1 : gridDim.x;
    for (; t < tend; t += tstep) {
        const int64_t base = t * tile;
        d2 v[U]; int c0[U], c1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + (int64_t)u * BS * 2 + threadIdx.x * 2;
            if (i + 1 < n) {
                if (NT) v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(x + i));
                else v[u] = *reinterpret_cast<const d2*>(x + i);
                if (MODE == 0) { unsigned cc = *reinterpret_cast<const uint16_t*>(color + i); c0[u] = cc & 0xFF; c1[u] = cc >> 8; }
                else { const int r = (int)(i % 3); c0[u] = r; c1[u] = r == 2 ? 0 : r + 1; }
            } else { v[u] = d2{0, 0}; c0[u] = c1[u] = -2; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double s0 = v[u].x * v[u].x, s1 = v[u].y * v[u].y;
#pragma unroll
            for (int c = 0; c < NC; ++c) { acc[c] += (c0[u] == c) ? s0 : 0.0; acc[c] += (c1[u] == c) ? s1 : 0.0; }
        }
    }
    __shared__ double red[BS / 64][NC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NC; ++c) { const double s = wave_sum(acc[c]); if (lane == 0) red[wave][c] = s; }
    __syncthreads();
    if (threadIdx.x < NC) { double s = 0; for (int w = 0; w < BS / 64; ++w) s += red[w][threadIdx.x]; partial[(int64_t)blockIdx.x * 8 + threadIdx.x] = s; }
}
template <bool NT> __global__ void __launch_bounds__(BS) k_store(d2* __restrict__ d, int64_t n, double v) {
    const int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    if (i < n) { if (NT) __builtin_nontemporal_store(d2{v, v}, d + i); else d[i] = d2{v, v}; }
}
__global__ void __launch_bounds__(BS) k_read(const d2* __restrict__ s, int64_t n, double* __restrict__ sink) {
    const int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    double a = 0;
    if (i < n) { const d2 v = s[i]; a = v.x + v.y; }
    if (a == 1.2345e300) sink[0] = a;
}
int main() {
    const int64_t n = 10000000;
    const int64_t GB = 1ll << 30;
    double *x, *partial, *sink; d2 *junk, *nz; uint8_t* col;
    hipMalloc(&x, n * 8); hipMalloc(&col, n); hipMalloc(&partial, 8 * 8 * 65536); hipMalloc(&junk, GB); hipMalloc(&nz, 240000000);
    hipMalloc(&sink, 64);
    std::vector<double> hx(n); std::vector<uint8_t> hc(n);
    for (int64_t i = 0; i < n; ++i) { hx[i] = (i % 1000) * 1e-3; hc[i] = i % 3; }
    hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(col, hc.data(), n, hipMemcpyHostToDevice);
    hipMemset(junk, 1, GB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto dirty_flush = [&] { hipLaunchKernelGGL(k_store<false>, dim3((unsigned)(GB / 16 / BS)), dim3(BS), 0, 0, junk, GB / 16, 1.0); };
    auto clean_flush = [&] { hipLaunchKernelGGL(k_read, dim3((unsigned)(GB / 16 / BS)), dim3(BS), 0, 0, junk, GB / 16, sink); };
    auto nz_store = [&](bool nt) {
        const int64_t m = 240000000 / 16;
        if (nt) hipLaunchKernelGGL(k_store<true>, dim3((unsigned)((m + BS - 1) / BS)), dim3(BS), 0, 0, nz, m, 2.0);
        else hipLaunchKernelGGL(k_store<false>, dim3((unsigned)((m + BS - 1) / BS)), dim3(BS), 0, 0, nz, m, 2.0);
    };
    auto run = [&](const char* name, int pre, auto launch) {
        float tot = 0, best = 1e9;
        for (int r = 0; r < 12; ++r) {
            if (pre == 0) dirty_flush();
            else if (pre == 1) { dirty_flush(); clean_flush(); clean_flush(); }
            else if (pre == 2) { dirty_flush(); clean_flush(); clean_flush(); nz_store(false); }
            else if (pre == 3) { dirty_flush(); clean_flush(); clean_flush(); nz_store(true); }
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) { tot += ms; best = ms < best ? ms : best; }
        }
        static const char* pn[] = {"after 1GiB stores (dirty)", "after 1GiB reads (clean)", "after 240MB stores", "after 240MB NT stores", "warm (x resident)"};
        printf("%-40s | %-26s avg %6.1f us best %6.1f us  (%5.0f GB/s of 90 MB)\n", name, pn[pre], tot / 10 * 1e3, best * 1e3, 9.0 * n / (best * 1e-3) / 1e9);
    };
    for (int pre = 0; pre < 5; ++pre) {
        run("U4 colours grid=977 stride", pre, [&] { hipLaunchKernelGGL((k_eps<4, 4, 0, false>), dim3(977), dim3(BS), 0, 0, x, col, n, partial, 0); });
        run("U4 colours grid=977 stride NT", pre, [&] { hipLaunchKernelGGL((k_eps<4, 4, 0, true>), dim3(977), dim3(BS), 0, 0, x, col, n, partial, 0); });
    }
    for (int pre : {0, 1, 2}) {
        for (int g : {977, 1628, 2442, 4883}) {
            char nm[96];
            snprintf(nm, 96, "U4 colours grid=%d contiguous", g);
            run(nm, pre, [&] { hipLaunchKernelGGL((k_eps<4, 4, 0, false>), dim3(g), dim3(BS), 0, 0, x, col, n, partial, 1); });
            snprintf(nm, 96, "U4 cyclic(no colour load) grid=%d stride", g);
            run(nm, pre, [&] { hipLaunchKernelGGL((k_eps<4, 4, 1, false>), dim3(g), dim3(BS), 0, 0, x, col, n, partial, 0); });
        }
        run("U8 cyclic grid=2442 stride", pre, [&] { hipLaunchKernelGGL((k_eps<4, 8, 1, false>), dim3(2442), dim3(BS), 0, 0, x, col, n, partial, 0); });
        run("U2 cyclic grid=9766 stride", pre, [&] { hipLaunchKernelGGL((k_eps<4, 2, 1, false>), dim3(9766), dim3(BS), 0, 0, x, col, n, partial, 0); });
        run("U1 cyclic grid=19532 one-shot", pre, [&] { hipLaunchKernelGGL((k_eps<4, 1, 1, false>), dim3(19532), dim3(BS), 0, 0, x, col, n, partial, 0); });
    }
    // the nzval-like store kernel itself, after a clean flush: what does a 240 MB store cost, NT or not?
    for (int nt = 0; nt < 2; ++nt) {
        float best = 1e9;
        for (int r = 0; r < 6; ++r) {
            dirty_flush(); clean_flush(); clean_flush();
            hipEventRecord(e0); nz_store(nt != 0); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("240 MB %s store after clean flush: best %.1f us (%.0f GB/s)\n", nt ? "NT" : "plain", best * 1e3, 240e6 / (best * 1e-3) / 1e9);
    }
    return 0;
}
