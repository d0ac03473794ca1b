// Can the IEEE division by a per-colour constant (df / eps, one per stored entry) be replaced by a multiply and two
// FMA correction steps WITHOUT changing a single bit?   y = RN(1/b) (true division, once per colour),
//   q0 = RN(a y);  r0 = RN(a - b q0) [fma];  q1 = RN(q0 + r0 y) [fma];  r1 = RN(a - b q1) [fma];  q2 = RN(q1 + r1 y) [fma]
// q1 is a faithful rounding of a/b, so by Markstein's theorem q2 = RN(a/b) when nothing over/underflows; zero, tiny, huge
// and non-finite numerators or quotients take the true division.  This probe counts bitwise mismatches against a / b over random and
// adversarial operands (quotients next to rounding ties) and times both forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ double fast_div(double a, double b, double y) {
    const double q0 = a * y;
    const double m = fabs(q0), ma = fabs(a);
    // (the remainders are ~2^-52 of a and of q: both operands well inside the normal range, else the true division)
    if (!(m >= 0x1p-900 && m <= 0x1p900 && ma >= 0x1p-900 && ma <= 0x1p900)) return a / b;
    const double r0 = __builtin_fma(-b, q0, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-b, q1, a);
    return __builtin_fma(r1, y, q1);
}
__device__ __forceinline__ uint64_t splitmix(uint64_t &s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__device__ __forceinline__ double from_bits(uint64_t u) { return __longlong_as_double((long long)u); }
__global__ void k_check(uint64_t seed, int iters, int mode, unsigned long long *bad, double *sample) {
    uint64_t s = seed + 0x1234567ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x);
    unsigned long long nb = 0;
    for (int it = 0; it < iters; ++it) {
        // divisor: an eps-like value (1e-12 .. 1e-2, either sign), or any normal double with a moderate exponent
        uint64_t rb = splitmix(s);
        double b = from_bits((rb & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 40 + (rb >> 52) % 34) << 52));
        if (mode == 1) b = from_bits((rb & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 300 + (rb >> 52) % 600) << 52));
        const double y = 1.0 / b;
        uint64_t ra = splitmix(s);
        double a;
        if (mode == 2) {
            // adversarial: a chosen so that a / b lies next to a rounding tie: a = RN(b * (q + half ulp)) +- few ulps
            const double q = from_bits((ra & 0x000FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 20 + (ra >> 52) % 40) << 52));
            const double qh = from_bits((uint64_t)__double_as_longlong(q));   // q
            const double mid = __builtin_fma(qh, b, 0.5 * b * (from_bits(((uint64_t)__double_as_longlong(q) & 0x7FF0000000000000ull)) * 0x1p-52));
            a = from_bits((uint64_t)__double_as_longlong(mid) + ((ra >> 60) & 7) - 3);
        } else {
            a = from_bits((ra & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 60 + (ra >> 52) % 120) << 52));
            if (mode == 1) a = from_bits((ra & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1 + (ra >> 52) % 2045) << 52));
            if ((ra & 0xFFFF) == 0) a = 0.0;
            if ((ra & 0xFFFF) == 1) a = -0.0;
        }
        const double t = a / b, f = fast_div(a, b, y);
        if (__double_as_longlong(t) != __double_as_longlong(f) && !(t != t && f != f)) { if (!nb) { sample[0] = a; sample[1] = b; sample[2] = t; sample[3] = f; } ++nb; }
    }
    if (nb) atomicAdd(bad, nb);
}
template <bool FAST> __global__ void k_time(const double *a, const double *eps, double *out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double b = eps[i & 3], y = 1.0 / b;
    double acc = a[i];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = FAST ? fast_div(acc, b, y) + 1.0 : acc / b + 1.0;
    out[i] = acc;
}
int main() {
    unsigned long long *bad; double *sample;
    hipMalloc(&bad, 8); hipMalloc(&sample, 32);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(bad, 0, 8);
        const int blocks = 4096, threads = 256, iters = 4000;
        for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(k_check, dim3(blocks), dim3(threads), 0, 0, 777ull + 1000003ull * rep + mode, iters, mode, bad, sample);
        unsigned long long hb = 0; double hs[4];
        hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hs, sample, 32, hipMemcpyDeviceToHost);
        printf("mode %d (%s): %.3g pairs, mismatches %llu", mode, mode == 0 ? "eps-like divisors" : mode == 1 ? "wide exponent ranges" : "quotients next to ties",
               4.0 * blocks * threads * iters, hb);
        if (hb) printf("  e.g. a=%a b=%a true=%a fast=%a", hs[0], hs[1], hs[2], hs[3]);
        printf("\n");
    }
    const int64_t n = 1 << 24;
    double *a, *e, *o; hipMalloc(&a, n * 8); hipMalloc(&e, 32); hipMalloc(&o, n * 8);
    hipMemset(a, 0x3f, n * 8); const double he[4] = {1e-7, 2e-7, 3e-7, 5e-7}; hipMemcpy(e, he, 32, hipMemcpyHostToDevice);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    for (int fast = 0; fast < 2; ++fast) {
        for (int w = 0; w < 2; ++w) { if (fast) hipLaunchKernelGGL(k_time<true>, dim3(n / 256), dim3(256), 0, 0, a, e, o, n); else hipLaunchKernelGGL(k_time<false>, dim3(n / 256), dim3(256), 0, 0, a, e, o, n); }
        hipEventRecord(t0);
        for (int w = 0; w < 10; ++w) { if (fast) hipLaunchKernelGGL(k_time<true>, dim3(n / 256), dim3(256), 0, 0, a, e, o, n); else hipLaunchKernelGGL(k_time<false>, dim3(n / 256), dim3(256), 0, 0, a, e, o, n); }
        hipEventRecord(t1); hipEventSynchronize(t1);
        float ms; hipEventElapsedTime(&ms, t0, t1);
        printf("%s: %.1f us per launch (16 dependent divisions per element, %lld elements)\n", fast ? "multiply + 2 fma corrections" : "IEEE division", ms * 100, (long long)n);
    }
    return 0;
}
