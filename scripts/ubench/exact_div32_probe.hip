// Float32: can the IEEE division by a per-column constant (df / eps) go through Float64 without changing a bit?
//   yd = RN64(1 / (double)b)  (once per divisor);   q = RN32(RN64((double)a * yd))
// The exact quotient of two 24-bit significands is either representable in 25 bits or at least 2^-49 (relative) away from every 25-bit
// midpoint (a - m b is a non-zero integer multiple of the operands' common unit), while (double)a * yd is within 2^-52 of it: the
// Float64 product lies on the same side of every Float32 rounding boundary as the exact quotient, so rounding it gives RN32(a / b).
// A tiny / huge / zero / non-finite DIVISOR and a tiny non-zero or NaN numerator take the true division.  This probe counts bitwise mismatches against a / b over random
// and adversarial operands (quotients next to Float32 rounding ties) and times both forms.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/ubench/exact_div32_probe.hip -o /tmp/exact_div32_probe && /tmp/exact_div32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float fast_div(float a, float b, double yd)
{
    // (the divisor's range is tested once per column in a kernel; of the numerator only "not tiny": a DENORMAL quotient has fewer bits,
    //  its rounding ties are not 2^-49 away any more -- measured: 3 mismatches in 1.7e10 without this test.  Zero, huge and infinite
    //  numerators are fine: the Float64 product neither overflows nor loses bits that matter before the final rounding)
    const float mb = fabsf(b), ma = fabsf(a);
    if (!(mb >= 0x1p-60f && mb <= 0x1p60f && (ma >= 0x1p-60f || ma == 0.0f))) return a / b;
    return (float)((double)a * yd);
}
__device__ __forceinline__ uint64_t splitmix(uint64_t &s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__global__ void k_check(uint64_t seed, int iters, int mode, unsigned long long *bad, float *sample)
{
    uint64_t s = seed + 0x1234567ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x);
    unsigned long long nb = 0;
    for (int it = 0; it < iters; ++it) {
        const uint64_t rb = splitmix(s), ra = splitmix(s);
        // divisor: eps-like (2^-30 .. 2^-3), or any exponent in the guarded range and beyond (mode 1)
        unsigned eb = mode == 1 ? 127 - 80 + (unsigned)((rb >> 40) % 160) : 127 - 30 + (unsigned)((rb >> 40) % 28);
        const float b = __uint_as_float(((unsigned)rb & 0x807FFFFFu) | (eb << 23));
        const double yd = 1.0 / (double)b;
        float a;
        if (mode == 2) {
            // adversarial: a / b next to a Float32 rounding tie: q a 24-bit value, m = q + half an ulp, a = RN32(m b) +- a few ulps
            const float q = __uint_as_float(((unsigned)ra & 0x007FFFFFu) | ((127 - 10 + (unsigned)((ra >> 40) % 20)) << 23));
            const double m = (double)q + 0.5 * (double)__uint_as_float(__float_as_uint(q) & 0x7F800000u) * 0x1p-23;
            const float a0 = (float)(m * (double)b);
            a = __uint_as_float(__float_as_uint(a0) + (unsigned)((ra >> 60) & 7) - 3u);
        } else {
            unsigned ea = mode == 1 ? (unsigned)((ra >> 40) % 256) : 127 - 40 + (unsigned)((ra >> 40) % 80);      // (mode 1: denormals, infinities and NaNs among them)
            a = __uint_as_float(((unsigned)ra & 0x807FFFFFu) | (ea << 23));
            if ((ra >> 32 & 0xFFFF) == 0) a = 0.0f;
            if ((ra >> 32 & 0xFFFF) == 1) a = -0.0f;
            if ((ra >> 32 & 0xFFFF) == 2) a = __uint_as_float((unsigned)ra & 0x807FFFFFu);      // a denormal
        }
        const float t = a / b, f = fast_div(a, b, yd);
        if (__float_as_uint(t) != __float_as_uint(f) && !(t != t && f != f)) { if (!nb) { sample[0] = a; sample[1] = b; sample[2] = t; sample[3] = f; } ++nb; }
    }
    if (nb) atomicAdd(bad, nb);
}
template <bool FAST> __global__ void k_time(const float *a, const float *eps, float *out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float b = eps[i & 3];
    const double yd = 1.0 / (double)b;
    float acc = a[i];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = FAST ? fast_div(acc, b, yd) + 1.0f : acc / b + 1.0f;
    out[i] = acc;
}
int main()
{
    unsigned long long *bad; float *sample;
    hipMalloc(&bad, 8); hipMalloc(&sample, 16);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(bad, 0, 8);
        const int blocks = 4096, threads = 256, iters = 4000;
        for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(k_check, dim3(blocks), dim3(threads), 0, 0, 777ull + 1000003ull * rep + mode, iters, mode, bad, sample);
        unsigned long long hb = 0; float hs[4];
        hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hs, sample, 16, hipMemcpyDeviceToHost);
        printf("mode %d (%s): %.3g pairs, mismatches %llu", mode, mode == 0 ? "eps-like divisors" : mode == 1 ? "wide exponent ranges, zeros, denormals" : "quotients next to ties",
               4.0 * blocks * threads * iters, hb);
        if (hb) printf("  e.g. a=%a b=%a true=%a fast=%a", hs[0], hs[1], hs[2], hs[3]);
        printf("\n");
    }
    const int64_t n = 1 << 24;
    float *a, *e, *o; hipMalloc(&a, n * 4); hipMalloc(&e, 16); hipMalloc(&o, n * 4);
    hipMemset(a, 0x3f, n * 4); const float he[4] = {1e-3f, 2e-3f, 3e-3f, 5e-3f}; hipMemcpy(e, he, 16, hipMemcpyHostToDevice);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    for (int fast = 0; fast < 2; ++fast) {
        for (int w = 0; w < 2; ++w) { if (fast) hipLaunchKernelGGL(k_time<true>, dim3(n / 256), dim3(256), 0, 0, a, e, o, n); else hipLaunchKernelGGL(k_time<false>, dim3(n / 256), dim3(256), 0, 0, a, e, o, n); }
        hipEventRecord(t0);
        for (int w = 0; w < 10; ++w) { if (fast) hipLaunchKernelGGL(k_time<true>, dim3(n / 256), dim3(256), 0, 0, a, e, o, n); else hipLaunchKernelGGL(k_time<false>, dim3(n / 256), dim3(256), 0, 0, a, e, o, n); }
        hipEventRecord(t1); hipEventSynchronize(t1);
        float ms = 0; hipEventElapsedTime(&ms, t0, t1);
        printf("%s: %.1f us per launch (16 dependent divisions per thread, %lld threads)\n", fast ? "through Float64" : "a / b", ms * 100.0, (long long)n);
    }
    return 0;
}
