// What does timing one kernel per step cost?  (a) nothing, (b) hipEventRecord before + after (two marker packets),
// (c) hipExtLaunchKernelGGL with start / stop events (timestamps of the dispatch itself).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_axpy(double *y, const double *x, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 2.0 * x[i] + 1.0;
}
int main()
{
    const long n = 1 << 20;
    double *x, *y;
    hipMalloc(&x, n * 8); hipMalloc(&y, n * 8);
    hipMemset(x, 0, n * 8);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int reps = 2000;
    std::vector<hipEvent_t> ev(2 * reps);
    for (auto &e : ev) hipEventCreate(&e);
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int mode = 0; mode < 3; ++mode) {
        for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(k_axpy, dim3(n / 256), dim3(256), 0, s, y, x, n);
        hipStreamSynchronize(s);
        const double t0 = now();
        for (int r = 0; r < reps; ++r) {
            if (mode == 1) hipEventRecord(ev[2 * r], s);
            if (mode == 2) hipExtLaunchKernelGGL(k_axpy, dim3(n / 256), dim3(256), 0, s, ev[2 * r], ev[2 * r + 1], 0, y, x, n);
            else hipLaunchKernelGGL(k_axpy, dim3(n / 256), dim3(256), 0, s, y, x, n);
            if (mode == 1) hipEventRecord(ev[2 * r + 1], s);
        }
        const double t1 = now();
        hipStreamSynchronize(s);
        const double t2 = now();
        double sum = 0; int cnt = 0;
        if (mode) for (int r = 0; r < reps; ++r) { float ms = 0; if (hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]) == hipSuccess) { sum += ms; ++cnt; } }
        printf("mode %d (%s): host %.2f us/launch, wall %.2f us/launch, event-measured kernel %.2f us (%d samples)\n", mode,
               mode == 0 ? "plain" : mode == 1 ? "hipEventRecord x2" : "hipExtLaunchKernelGGL events", (t1 - t0) / reps, (t2 - t0) / reps,
               cnt ? sum / cnt * 1e3 : 0.0, cnt);
    }
    return 0;
}
