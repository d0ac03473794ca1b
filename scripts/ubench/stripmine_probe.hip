// Can the f! -> decompression hand-off (320 MB written, 320 MB read back per Jacobian at N = 1e7) live in the 256 MiB
// Infinity Cache if the call is strip-mined into row chunks that REUSE one small scratch?
//   producer (f!-like):   S[b][r] = x[r] * w_b            b = 0..3     reads 8 B/row, writes 32 B/row
//   consumer (decompress-like): out[3r..3r+2] = S[1..3][r] - S[0][r]   reads 32 B/row, writes 24 B/row
// Steady-state loop over "steps"; a step = K chunk pairs (producer, consumer) over N rows, the scratch holds N/K rows of
// 4 arrays and is reused by every chunk.  K = 1 is today's pipeline (scratch = 320 MB).  Streaming accesses (x loads,
// out stores) optionally non-temporal so that they do not displace the scratch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int BS = 256;
typedef double __attribute__((ext_vector_type(2))) d2;
template <bool NTL>
__global__ void __launch_bounds__(BS) k_prod(const double* __restrict__ x, double* __restrict__ S, int64_t ld, int64_t r0, int64_t r1) {
    const int64_t r = r0 + ((int64_t)blockIdx.x * BS + threadIdx.x) * 2;
    if (r >= r1) return;
    d2 v;
    if (NTL) v = __builtin_nontemporal_load(reinterpret_cast<const d2*>(x + r)); else v = *reinterpret_cast<const d2*>(x + r);
    const int64_t o = r - r0;
#pragma unroll
    for (int b = 0; b < 4; ++b) *reinterpret_cast<d2*>(S + b * ld + o) = d2{v.x * (1.0 + b), v.y * (1.0 + b)};
}
template <bool NTS>
__global__ void __launch_bounds__(BS) k_cons(const double* __restrict__ S, int64_t ld, double* __restrict__ out, int64_t r0, int64_t r1) {
    const int64_t r = r0 + ((int64_t)blockIdx.x * BS + threadIdx.x) * 2;
    if (r >= r1) return;
    const int64_t o = r - r0;
    d2 s[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) s[b] = *reinterpret_cast<const d2*>(S + b * ld + o);
    // 6 outputs for 2 rows: 3 x 16-B stores
    d2 q0 = {s[1].x - s[0].x, s[2].x - s[0].x}, q1 = {s[3].x - s[0].x, s[1].y - s[0].y}, q2 = {s[2].y - s[0].y, s[3].y - s[0].y};
    d2* op = reinterpret_cast<d2*>(out + 3 * r);
    if (NTS) { __builtin_nontemporal_store(q0, op); __builtin_nontemporal_store(q1, op + 1); __builtin_nontemporal_store(q2, op + 2); }
    else { op[0] = q0; op[1] = q1; op[2] = q2; }
}
int main() {
    const int64_t n = 10000000;
    double *x, *S, *out;
    hipMalloc(&x, n * 8); hipMalloc(&S, 4 * n * 8 + 4096); hipMalloc(&out, 3 * n * 8 + 64);
    hipMemset(x, 0, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nt = 0; nt < 4; ++nt)
        for (int K : {1, 2, 3, 4, 6, 8, 12, 16, 32}) {
            const int64_t chunk = ((n + K - 1) / K + 511) & ~(int64_t)511;
            const int64_t ld = chunk;
            auto step = [&] {
                for (int64_t r0 = 0; r0 < n; r0 += chunk) {
                    const int64_t r1 = r0 + chunk < n ? r0 + chunk : n;
                    const unsigned g = (unsigned)(((r1 - r0) / 2 + BS - 1) / BS);
                    if (nt & 1) hipLaunchKernelGGL(k_prod<true>, dim3(g), dim3(BS), 0, 0, x, S, ld, r0, r1);
                    else hipLaunchKernelGGL(k_prod<false>, dim3(g), dim3(BS), 0, 0, x, S, ld, r0, r1);
                    if (nt & 2) hipLaunchKernelGGL(k_cons<true>, dim3(g), dim3(BS), 0, 0, S, ld, out, r0, r1);
                    else hipLaunchKernelGGL(k_cons<false>, dim3(g), dim3(BS), 0, 0, S, ld, out, r0, r1);
                }
            };
            for (int w = 0; w < 3; ++w) step();
            hipEventRecord(e0);
            const int reps = 20;
            for (int r = 0; r < reps; ++r) step();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("x loads %-3s out stores %-3s K=%2d scratch %6.1f MB : %7.1f us per step (2 x %d launches)\n", (nt & 1) ? "NT" : "-", (nt & 2) ? "NT" : "-",
                   K, 4.0 * chunk * 8 / 1e6, ms / reps * 1e3, K);
        }
    return 0;
}
