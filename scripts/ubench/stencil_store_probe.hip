// Round 3: the fused "f! + quotient + store" launch for the 5-point stencil on an nx x ny grid (config 3: 4000 x 2500, central
// differences, 5 colours (i + 2j) mod 5), CSC nzval in storage order.  COLUMN-centric like the tridiagonal wave kernel: lane t of
// a wavefront owns the grid columns (i, j), (i + 1, j) with i = i0 + 2t, loads the 6 x 5 window of x around them (11 aligned
// 16-B loads), evaluates the five rows each column touches at x +- eps e_k, and the wavefront's 128 x 5 quotients go through a
// wave-private LDS window into dense aligned non-temporal 16-B stores (the first / last column of a grid row hold 4 entries:
// the slots shift by one).  Checked bit for bit against the row-centric, colour-batched evaluation on the host (the reference's
// loop order: perturb colour c everywhere, f!, difference, decompress).
//   out bytes: 8 * (5N - 2nx - 2ny) = 400 MB, in: 80 MB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef double __attribute__((ext_vector_type(2))) d2;
constexpr int BS = 256;

template <bool NL> __device__ __host__ inline double frow(double c, double w, double e, double s, double n) {
    double v = (((w + e) + s) + n) - 4.0 * c;
    if (NL) v = v + (c * c) * e;
    return v;
}
__device__ __host__ inline long long colptr5(long long k, long long nx, long long ny) {
    const long long j = k / nx, i = k - j * nx;
    const long long north = k - (ny - 1) * nx;
    return 5 * k - (k < nx ? k : nx) - (j + (i > 0 ? 1 : 0)) - j - (north > 0 ? north : 0);
}
__device__ inline d2 ldp(const double* __restrict__ x, long long k, int i, int nx, bool rowok) {
    // the pair (i, i+1) of a grid row starting at x + k - i; zero outside the grid
    if (rowok && i >= 0 && i + 1 < nx) return *reinterpret_cast<const d2*>(x + k);
    d2 v = {0.0, 0.0};
    if (rowok && i >= 0 && i < nx) v.x = x[k];
    if (rowok && i + 1 >= 0 && i + 1 < nx) v.y = x[k + 1];
    return v;
}
__device__ inline long long xcd_tile(long long block, long long ntiles) { return (block & 7) * ((ntiles + 7) / 8) + (block >> 3); }

template <bool NL, bool NT, bool XCD>
__global__ void __launch_bounds__(BS) k_stencil_store(const double* __restrict__ x, const double* __restrict__ eps, double* __restrict__ out,
                                                      int nx, int ny, long long nnz) {
    __shared__ __attribute__((aligned(16))) double s_all[BS / 64][648];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* win = s_all[wave];
    const int T = (nx + 127) / 128;
    const long long ngroups = ((long long)ny * T + 3) / 4;
    const long long grp = XCD ? xcd_tile(blockIdx.x, ngroups) : blockIdx.x;
    if (grp >= ngroups) return;
    const long long wt = grp * 4 + wave;
    if (wt >= (long long)ny * T) return;
    const int j = (int)(wt / T), i0 = (int)(wt - (long long)j * T) * 128;
    const int i = i0 + 2 * lane;
    const long long k = (long long)j * nx + i;
    const bool act = i < nx;
    // window rows j-2 .. j+2, columns i-2 .. i+3
    double W[5][6];
#pragma unroll
    for (int dj = -2; dj <= 2; ++dj) {
        const bool rowok = act && j + dj >= 0 && j + dj < ny;
        const long long kr = k + (long long)dj * nx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if ((dj == -2 || dj == 2) && c != 1) { W[dj + 2][2 * c] = 0.0; W[dj + 2][2 * c + 1] = 0.0; continue; }
            const d2 v = ldp(x, kr + 2 * (c - 1), i + 2 * (c - 1), nx, rowok);
            W[dj + 2][2 * c] = v.x; W[dj + 2][2 * c + 1] = v.y;
        }
    }
    // x + 0.0 of every unperturbed coordinate (what the coloured point holds there); x - 0.0 == x
    double P0[5][6];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) P0[a][b] = W[a][b] + 0.0;
    double q[10];
    bool ex[10];
    const int c0 = (i + 2 * j) % 5;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int ii = i + o;
        const int cc = c0 + o >= 5 ? c0 + o - 5 : c0 + o;
        const double e = eps[cc];
        const double xc = W[2][2 + o], pc = xc + e, mc = xc - e, e2 = 2 * e;
        const bool hw = ii > 0, he = ii < nx - 1, hs = j > 0, hn = j < ny - 1;
#define PV(dj, di) P0[(dj) + 2][(di) + 2 + o]
#define MV(dj, di) W[(dj) + 2][(di) + 2 + o]
        // S row (ii, j-1): n = centre
        {
            const bool rhw = hw, rhe = he, rhs = j - 1 > 0;
            const double pl = frow<NL>(PV(-1, 0), rhw ? PV(-1, -1) : 0.0, rhe ? PV(-1, 1) : 0.0, rhs ? PV(-2, 0) : 0.0, pc);
            const double mi = frow<NL>(MV(-1, 0), rhw ? MV(-1, -1) : 0.0, rhe ? MV(-1, 1) : 0.0, rhs ? MV(-2, 0) : 0.0, mc);
            q[5 * o + 0] = (pl - mi) / e2; ex[5 * o + 0] = hs;
        }
        // W row (ii-1, j): e = centre
        {
            const bool rhw = ii - 1 > 0;
            const double pl = frow<NL>(PV(0, -1), rhw ? PV(0, -2) : 0.0, pc, hs ? PV(-1, -1) : 0.0, hn ? PV(1, -1) : 0.0);
            const double mi = frow<NL>(MV(0, -1), rhw ? MV(0, -2) : 0.0, mc, hs ? MV(-1, -1) : 0.0, hn ? MV(1, -1) : 0.0);
            q[5 * o + 1] = (pl - mi) / e2; ex[5 * o + 1] = hw;
        }
        // C row
        {
            const double pl = frow<NL>(pc, hw ? PV(0, -1) : 0.0, he ? PV(0, 1) : 0.0, hs ? PV(-1, 0) : 0.0, hn ? PV(1, 0) : 0.0);
            const double mi = frow<NL>(mc, hw ? MV(0, -1) : 0.0, he ? MV(0, 1) : 0.0, hs ? MV(-1, 0) : 0.0, hn ? MV(1, 0) : 0.0);
            q[5 * o + 2] = (pl - mi) / e2; ex[5 * o + 2] = true;
        }
        // E row (ii+1, j): w = centre
        {
            const bool rhe = ii + 1 < nx - 1;
            const double pl = frow<NL>(PV(0, 1), pc, rhe ? PV(0, 2) : 0.0, hs ? PV(-1, 1) : 0.0, hn ? PV(1, 1) : 0.0);
            const double mi = frow<NL>(MV(0, 1), mc, rhe ? MV(0, 2) : 0.0, hs ? MV(-1, 1) : 0.0, hn ? MV(1, 1) : 0.0);
            q[5 * o + 3] = (pl - mi) / e2; ex[5 * o + 3] = he;
        }
        // N row (ii, j+1): s = centre
        {
            const bool rhn = j + 1 < ny - 1;
            const double pl = frow<NL>(PV(1, 0), hw ? PV(1, -1) : 0.0, he ? PV(1, 1) : 0.0, pc, rhn ? PV(2, 0) : 0.0);
            const double mi = frow<NL>(MV(1, 0), hw ? MV(1, -1) : 0.0, he ? MV(1, 1) : 0.0, mc, rhn ? MV(2, 0) : 0.0);
            q[5 * o + 4] = (pl - mi) / e2; ex[5 * o + 4] = hn;
        }
#undef PV
#undef MV
    }
    const bool fast = j >= 1 && j <= ny - 2;
    if (!fast) {
        if (!act) return;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            long long p = colptr5(k + o, nx, ny);
#pragma unroll
            for (int m = 0; m < 5; ++m) if (ex[5 * o + m]) out[p++] = q[5 * o + m];
        }
        return;
    }
    // interior grid row: slots in storage order; the first column of the grid row has no W entry, the last no E entry
    const int nc = nx - i0 < 128 ? nx - i0 : 128;
    const long long Pt = colptr5((long long)j * nx + i0, nx, ny);
    const int off = (int)(Pt & 1);
    const int cnt = 5 * nc - (i0 == 0 ? 1 : 0) - (i0 + nc == nx ? 1 : 0);
    if (act) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int ii = i + o;
            const int base = off + 5 * (ii - i0) - (i0 == 0 && ii > 0 ? 1 : 0);
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                if (!ex[5 * o + m]) continue;
                int sl = base + m;
                if (ii == 0 && m > 1) sl -= 1;
                if (ii == nx - 1 && m == 4) sl -= 1;
                win[sl] = q[5 * o + m];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double* obase = out + (Pt - off);
    const int lo = off, hi = off + cnt;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const int sl = 2 * (64 * a + lane);
        if (sl >= hi) break;
        const bool l0 = sl >= lo, l1 = sl + 1 < hi;
        if (l0 && l1) {
            const d2 v = *reinterpret_cast<const d2*>(win + sl);
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<d2*>(obase + sl));
            else *reinterpret_cast<d2*>(obase + sl) = v;
        } else {
            if (l0) obase[sl] = win[sl];
            if (l1 && sl + 1 >= lo) obase[sl + 1] = win[sl + 1];
        }
    }
}

__global__ void __launch_bounds__(BS) k_stream15(const double* __restrict__ x, double* __restrict__ out, long long npairs) {
    const long long t = (long long)blockIdx.x * BS + threadIdx.x;
    if (t >= npairs) return;
    const d2 v = *reinterpret_cast<const d2*>(x + 2 * t);
    d2* o = reinterpret_cast<d2*>(out + (long long)blockIdx.x * (BS * 10));
#pragma unroll
    for (int a = 0; a < 5; ++a) __builtin_nontemporal_store(d2{v.x + a, v.y}, o + a * BS + threadIdx.x);
}

template <class F> static float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms * 1e3f / reps;
}

template <bool NL> static void host_reference(const std::vector<double>& x, const double* he, int nx, int ny, std::vector<double>& ref) {
    // the reference's order: per colour c, the points x +- eps_c * mask_c (x + 0.0 / x - 0.0 elsewhere), f! row by row,
    // (f(+) - f(-)) / (2 eps), decompression: nzval[p] of every column of colour c = the quotient of its row
    const long long N = (long long)nx * ny;
    std::vector<double> xp(N), xm(N), d(N);
    for (int c = 0; c < 5; ++c) {
        const double e = he[c];
        for (long long k = 0; k < N; ++k) { const int j = (int)(k / nx), i = (int)(k - (long long)j * nx); const double dd = ((i + 2 * j) % 5 == c) ? e : 0.0; xp[k] = x[k] + dd; xm[k] = x[k] - dd; }
        for (long long k = 0; k < N; ++k) {
            const int j = (int)(k / nx), i = (int)(k - (long long)j * nx);
            const bool hw = i > 0, hee = i < nx - 1, hs = j > 0, hn = j < ny - 1;
            const double pl = frow<NL>(xp[k], hw ? xp[k - 1] : 0.0, hee ? xp[k + 1] : 0.0, hs ? xp[k - nx] : 0.0, hn ? xp[k + nx] : 0.0);
            const double mi = frow<NL>(xm[k], hw ? xm[k - 1] : 0.0, hee ? xm[k + 1] : 0.0, hs ? xm[k - nx] : 0.0, hn ? xm[k + nx] : 0.0);
            d[k] = (pl - mi) / (2 * e);
        }
        for (long long k = 0; k < N; ++k) {
            const int j = (int)(k / nx), i = (int)(k - (long long)j * nx);
            if ((i + 2 * j) % 5 != c) continue;
            long long p = colptr5(k, nx, ny);
            if (j > 0) ref[p++] = d[k - nx];
            if (i > 0) ref[p++] = d[k - 1];
            ref[p++] = d[k];
            if (i < nx - 1) ref[p++] = d[k + 1];
            if (j < ny - 1) ref[p++] = d[k + nx];
        }
    }
}

int main(int argc, char** argv) {
    const int nx = argc > 1 ? atoi(argv[1]) : 4000, ny = argc > 2 ? atoi(argv[2]) : 2500;
    const long long N = (long long)nx * ny, nnz = 5 * N - 2 * nx - 2 * ny;
    std::vector<double> hx(N);
    uint64_t st = 88172645463325252ull;
    for (long long i = 0; i < N; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; hx[i] = (double)(st >> 11) * (1.0 / 9007199254740992.0); }
    const double he[5] = {6.1e-6, 6.3e-6, 6.2e-6, 6.05e-6, 6.4e-6};
    double *x, *eps, *out;
    hipMalloc(&x, (size_t)N * 8 + 64); hipMalloc(&eps, 64); hipMalloc(&out, ((size_t)5 * N + 16) * 8);
    hipMemcpy(x, hx.data(), (size_t)N * 8, hipMemcpyHostToDevice);
    hipMemcpy(eps, he, sizeof he, hipMemcpyHostToDevice);
    if (colptr5(N, nx, ny) != nnz) { printf("colptr closed form is wrong: %lld vs %lld\n", colptr5(N, nx, ny), nnz); return 1; }
    std::vector<double> ref(nnz), got(nnz);
    const int T = (nx + 127) / 128;
    const long long ngroups = ((long long)ny * T + 3) / 4;
    const int reps = 30;
    const double MB = ((double)N * 8 + (double)nnz * 8) * 1e-6;
    printf("%d x %d grid, %0.1f MB per launch (x in, nzval out)\n", nx, ny, MB);
#define RUN(NL, NT, XCD) { const unsigned g = (unsigned)(XCD ? 8 * ((ngroups + 7) / 8) : ngroups); \
        auto f = [&] { hipLaunchKernelGGL((k_stencil_store<NL, NT, XCD>), dim3(g), dim3(BS), 0, 0, x, eps, out, nx, ny, nnz); }; \
        hipMemset(out, 0xFF, (size_t)nnz * 8); f(); \
        hipMemcpy(got.data(), out, (size_t)nnz * 8, hipMemcpyDeviceToHost); \
        host_reference<NL>(hx, he, nx, ny, ref); \
        long long bad = 0, first = -1; for (long long q = 0; q < nnz; ++q) if (memcmp(&got[q], &ref[q], 8) != 0) { if (first < 0) first = q; ++bad; } \
        float us = timeit(f, reps); \
        printf("stencil store nl=%d nt=%d xcd=%d  %7.1f us  %6.0f GB/s   mismatches vs colour-batched host reference: %lld (first %lld)\n", (int)NL, (int)NT, (int)XCD, us, MB / us * 1e3, bad, first); }
    RUN(false, true, true) RUN(true, true, true) RUN(false, false, true) RUN(false, true, false)
    {
        const long long npairs = N / 2; const unsigned g = (unsigned)((npairs + BS - 1) / BS);
        auto f = [&] { hipLaunchKernelGGL(k_stream15, dim3(g), dim3(BS), 0, 0, x, out, npairs); };
        float us = timeit(f, reps);
        printf("stream 1:5 (nt)                     %7.1f us  %6.0f GB/s (480 MB)\n", us, 480.0 / us * 1e3);
    }
    return 0;
}
