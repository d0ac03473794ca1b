// fd_plan_matches: is this plan still the plan of THESE pattern / colour arrays?
//
// The reference keeps `colorvec` and `sparsity` BY REFERENCE and re-reads them on every call (src/jacobians.jl:512-513, the
// O(nnz) pattern comparison of ext/FiniteDiffSparseArraysExt.jl:51-52): an in-place edit takes effect on the next call.  A plan
// is compiled from a snapshot.  A host shim therefore finds its plan in O(1) by the IDENTITY of the arrays (pointer + length)
// and offers an explicit invalidate; callers that want the reference's re-read semantics ask the library to compare CONTENT --
// with kernels when the arrays live on the device (nothing crosses PCIe, nothing is allocated), with host threads otherwise --
// never by copying the arrays or by a sampled hash.
//
// A plan created with FD_PLAN_FINGERPRINT records one 64-bit fingerprint per array it was compiled from:
//     F(a) = sum_k mix(value_2k + (value_2k+1 << 32) + k * K)   (mod 2^64; value_i = a[i0 + i] - index base, widened to 64 bits;
//                                                                a missing last partner counts as 0x7fffffff)
//     mix(z) = (z ^ (z >> 29)) * C, then ^ (>> 32)
// -- commutative, so blocks / threads add their partial sums in any order and the result is deterministic; position-dependent,
// so permuted or shifted contents differ; a function of the VALUES, not of the index width.  (Round 4 mixed every element with
// three 64-bit multiplications: the comparison of a 200 MB pattern was bound by the integer multiplier, 100 us; one multiplication
// per PAIR leaves it to the memory system.)  fd_plan_matches recomputes the fingerprints of the caller's current arrays over
// the same ranges (the plan's column window for colptr / rowval) and compares.
#include "fdjac_internal.h"

#include <algorithm>
#include <atomic>
#include <thread>

#ifndef FDJAC_F32   /* shared by both instantiations: defined once, by the Float64 build */

namespace {

constexpr uint64_t kFpK = 0xD6E8FEB86659FD93ull;      // odd: k -> k * K is a bijection mod 2^64
__host__ __device__ inline uint64_t fp_pair(int64_t v0, int64_t v1, uint64_t kK)      // kK = k * K (advanced incrementally)
{
    uint64_t z = (uint64_t)v0 + ((uint64_t)v1 << 32) + kK;
    z = (z ^ (z >> 29)) * 0xBF58476D1CE4E5B9ull;
    return z ^ (z >> 32);
}
// the pairs [k0, k0 + stride, ...) below npairs = ceil(n / 2) of the range a[i0 .. i0 + n), values minus `base`.  VEC: the range starts
// at an address aligned to two elements -- a pair is ONE load (8 / 16 bytes per lane, lane-consecutive) instead of two half-used ones.
template <typename IT, bool VEC> __device__ inline uint64_t fp_pairs(const IT *__restrict__ a, long long i0, long long n, long long base, long long k0, long long stride)
{
    typedef IT pair_t __attribute__((ext_vector_type(2)));
    const long long npairs = (n + 1) / 2;
    uint64_t s = 0, kK = (uint64_t)k0 * kFpK;
    const uint64_t dK = (uint64_t)stride * kFpK;
    long long k = k0;
    constexpr int U = 8;                                      // pairs (loads) in flight per thread
    for (; k + (U - 1) * stride < npairs; k += U * stride) {
        int64_t v0[U], v1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = 2 * (k + u * stride);
            if (VEC && i + 1 < n) {
                const pair_t p = __builtin_nontemporal_load(reinterpret_cast<const pair_t *>(a + i0 + i));
                v0[u] = (int64_t)p.x; v1[u] = (int64_t)p.y;
            } else {
                v0[u] = (int64_t)a[i0 + i];
                v1[u] = i + 1 < n ? (int64_t)a[i0 + i + 1] : (int64_t)0x7fffffff + base;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { s += fp_pair(v0[u] - base, v1[u] - base, kK); kK += dK; }
    }
    for (; k < npairs; k += stride) {
        const long long i = 2 * k;
        const int64_t v0 = (int64_t)a[i0 + i], v1 = i + 1 < n ? (int64_t)a[i0 + i + 1] : (int64_t)0x7fffffff + base;
        s += fp_pair(v0 - base, v1 - base, kK);
        kK += dK;
    }
    return s;
}
// Int32 arrays whose range starts 16-byte aligned: FOUR elements = two pairs per load (16 bytes per lane -- 8-byte accesses reach about
// 0.6 of the 16-byte rate on this part); quad q holds the pairs 2q and 2q + 1.  The sum is commutative: which lane adds which pair
// does not matter.
__device__ inline uint64_t fp_quads32(const int32_t *__restrict__ a, long long i0, long long n, long long base, long long q0, long long stride)
{
    const long long nquads = n / 4;               // whole quads; the tail (n mod 4 elements = up to 2 pairs) is added by the caller
    uint64_t s = 0, kK = (uint64_t)(2 * q0) * kFpK;
    const uint64_t dK = (uint64_t)(2 * stride) * kFpK;
    constexpr int U = 8;
    typedef int fp_i4 __attribute__((ext_vector_type(4)));
    long long q = q0;
    for (; q + (U - 1) * stride < nquads; q += U * stride) {
        fp_i4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const fp_i4 *>(a + i0 + 4 * (q + u * stride)));      // (read once: keep x / nzval in the caches)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s += fp_pair((int64_t)v[u].x - base, (int64_t)v[u].y - base, kK) + fp_pair((int64_t)v[u].z - base, (int64_t)v[u].w - base, kK + kFpK);
            kK += dK;
        }
    }
    for (; q < nquads; q += stride) {
        const int4 v = *reinterpret_cast<const int4 *>(a + i0 + 4 * q);
        s += fp_pair((int64_t)v.x - base, (int64_t)v.y - base, kK) + fp_pair((int64_t)v.z - base, (int64_t)v.w - base, kK + kFpK);
        kK += dK;
    }
    return s;
}
template <typename IT> __device__ inline uint64_t fp_pairs_any(const IT *a, long long i0, long long n, long long base, long long k0, long long stride)
{
    if constexpr (sizeof(IT) == 4) {
        if ((((unsigned long long)(a + i0)) & 15) == 0 && n >= 4) {
            uint64_t s = fp_quads32((const int32_t *)a, i0, n, base, k0, stride);
            // the pairs of the last n mod 4 elements: by the first thread(s) of the grid
            const long long kt = 2 * (n / 4), npairs = (n + 1) / 2;
            if (k0 < npairs - kt) {
                const long long k = kt + k0, i = 2 * k;
                const int64_t v0 = (int64_t)a[i0 + i], v1 = i + 1 < n ? (int64_t)a[i0 + i + 1] : (int64_t)0x7fffffff + base;
                s += fp_pair(v0 - base, v1 - base, (uint64_t)k * kFpK);
            }
            return s;
        }
    }
    const bool vec = (((unsigned long long)(a + i0)) & (2 * sizeof(IT) - 1)) == 0;
    return vec ? fp_pairs<IT, true>(a, i0, n, base, k0, stride) : fp_pairs<IT, false>(a, i0, n, base, k0, stride);
}

// one wave reads 64 consecutive elements per step; grid-stride over the array; one atomic per workgroup
template <typename IT> __global__ void __launch_bounds__(256) k_fingerprint(const IT *__restrict__ a, int64_t i0, int64_t n, int64_t base,
                                                                              unsigned long long *__restrict__ out)
{
    uint64_t s = fp_pairs_any<IT>(a, i0, n, base, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down((unsigned long long)s, o, 64);
    __shared__ uint64_t s_w[4];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)(s_w[0] + s_w[1] + s_w[2] + s_w[3]));
}

// fd_plan_matches_async: the three arrays in ONE launch -- workgroups [0, g0) fingerprint array 0, [g0, g0 + g1) array 1, the rest
// array 2 -- and the verdict on the device.  Every workgroup stores its sum in its own slot and takes a ticket of its GROUP (32
// groups, a line each: thousands of arrivals on ONE address were what the first version of this kernel spent its time on); the last
// arrival of a group takes the final ticket, the last of those adds the slots per array, compares with the plan's words fpx[3..5]
// and raises the two sticky stale words (pinned host memory).  No copy back, no synchronisation.  Layout of fpx (8-byte words):
// [3..5] the plan's fingerprints, [6] final ticket, [kFpGroup0 + 16 g] ticket of group g, [kFpPart0 + b] sum of workgroup b.
constexpr int kFpGroups = 32, kFpGroup0 = 16, kFpPart0 = kFpGroup0 + 16 * kFpGroups;
struct Fp3 {
    const void *a[3];
    int bytes[3];
    long long i0[3], n[3], base[3];
    int g[3];
};
__global__ void __launch_bounds__(256) k_fingerprint3_check(Fp3 f, unsigned long long *__restrict__ fpx, int *__restrict__ stale_plan, int *__restrict__ stale_ctx)
{
    int k = 0, b = (int)blockIdx.x;
    if (b >= f.g[0]) { b -= f.g[0]; k = 1; if (b >= f.g[1]) { b -= f.g[1]; k = 2; } }
    const long long n = f.n[k], stride = (long long)f.g[k] * 256, k0 = (long long)b * 256 + threadIdx.x;
    uint64_t s = f.bytes[k] == 8 ? fp_pairs_any<int64_t>((const int64_t *)f.a[k], f.i0[k], n, f.base[k], k0, stride)
                                 : fp_pairs_any<int32_t>((const int32_t *)f.a[k], f.i0[k], n, f.base[k], k0, stride);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down((unsigned long long)s, o, 64);
    __shared__ uint64_t s_w[4][3];
    __shared__ int s_last;
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6][0] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(fpx + kFpPart0 + blockIdx.x, (unsigned long long)(s_w[0][0] + s_w[1][0] + s_w[2][0] + s_w[3][0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the slot is out before the ticket is taken
        const int G = (int)gridDim.x < kFpGroups ? (int)gridDim.x : kFpGroups, grp = (int)blockIdx.x % G;
        const unsigned long long in_group = ((unsigned long long)gridDim.x - grp + G - 1) / G;
        int last = 0;
        if (__hip_atomic_fetch_add(fpx + kFpGroup0 + 16 * grp, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_group - 1) {
            __hip_atomic_store(fpx + kFpGroup0 + 16 * grp, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(fpx + 6, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)G - 1) {
                __hip_atomic_store(fpx + 6, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    uint64_t t[3] = {0, 0, 0};
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) {
        const unsigned long long v = __hip_atomic_load(fpx + kFpPart0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t[i < f.g[0] ? 0 : i < f.g[0] + f.g[1] ? 1 : 2] += v;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t[q] += __shfl_down((unsigned long long)t[q], o, 64);
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6][q] = t[q];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    bool same = true;
    for (int q = 0; q < 3; ++q)
        if (f.g[q] > 0 && s_w[0][q] + s_w[1][q] + s_w[2][q] + s_w[3][q] != fpx[3 + q]) same = false;
    if (!same) {
        __hip_atomic_store(stale_plan, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(stale_ctx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the pairs [klo, khi) of the range (host threads split the PAIRS)
template <typename IT> uint64_t host_range(const IT *a, int64_t i0, int64_t n, int64_t klo, int64_t khi, int64_t base)
{
    uint64_t s = 0, kK = (uint64_t)klo * kFpK;
    for (int64_t k = klo; k < khi; ++k) {
        const int64_t i = 2 * k;
        const int64_t v0 = (int64_t)a[i0 + i], v1 = i + 1 < n ? (int64_t)a[i0 + i + 1] : (int64_t)0x7fffffff + base;
        s += fp_pair(v0 - base, v1 - base, kK);
        kK += kFpK;
    }
    return s;
}

uint64_t host_fingerprint(const void *a, int bytes, int64_t i0, int64_t n, int64_t base)
{
    const int64_t np = (n + 1) / 2;      // pairs
    auto range = [&](int64_t lo, int64_t hi) {
        return bytes == 8 ? host_range((const int64_t *)a, i0, n, lo, hi, base) : host_range((const int32_t *)a, i0, n, lo, hi, base);
    };
    unsigned hw = std::thread::hardware_concurrency();
    const int64_t nthr = std::min<int64_t>({(int64_t)std::max(1u, hw), (int64_t)32, np >> 17});   // >= 2^18 elements per thread
    if (nthr <= 1) return range(0, np);
    std::vector<uint64_t> part((size_t)nthr, 0);
    std::vector<std::thread> th;
    const int64_t per = (np + nthr - 1) / nthr;
    int64_t done_to = 0;
    try {
        for (int64_t k = 0; k < nthr; ++k) {
            const int64_t lo = k * per, hi = std::min(np, lo + per);
            if (lo < hi) th.emplace_back([&part, &range, k, lo, hi] { part[(size_t)k] = range(lo, hi); });
            done_to = hi;
        }
    } catch (...) {
    }
    uint64_t s = done_to < np ? range(done_to, np) : 0;      // (a thread could not be started: the rest runs here)
    for (auto &t : th) t.join();
    for (uint64_t v : part) s += v;
    return s;
}

}  // namespace

// Fingerprints of up to three index arrays in one go.  Device arrays: one kernel per array accumulating into acc_dev[k] (3
// device words owned by the caller), ONE read-back.  Host arrays: host threads.  n[k] == 0 or a[k] == NULL skips array k (0).
extern "C" int fdjac_fingerprint3(const fd_ctx *ctx, const void *const *a, const int *bytes, const int64_t *i0, const int64_t *n,
                                  const int64_t *base, int memkind, unsigned long long *acc_dev, uint64_t *out)
{
    out[0] = out[1] = out[2] = 0;
    if (memkind == FD_HOST) {
        for (int k = 0; k < 3; ++k)
            if (a[k] && n[k] > 0) out[k] = host_fingerprint(a[k], bytes[k], i0[k], n[k], base[k]);
        return FD_OK;
    }
    hipStream_t s = ctx->stream;
    FD_HIP_CHECK(hipMemsetAsync(acc_dev, 0, 3 * sizeof(unsigned long long), s));
    for (int k = 0; k < 3; ++k) {
        if (!(a[k] && n[k] > 0)) continue;
        const int64_t blocks = std::min<int64_t>((n[k] + 255) / 256, (int64_t)ctx->num_cus * 2);      // (see fdjac_fingerprint3_check)
        if (bytes[k] == 8)
            hipLaunchKernelGGL((k_fingerprint<int64_t>), dim3((unsigned)blocks), dim3(256), 0, s, (const int64_t *)a[k], i0[k], n[k], base[k], acc_dev + k);
        else
            hipLaunchKernelGGL((k_fingerprint<int32_t>), dim3((unsigned)blocks), dim3(256), 0, s, (const int32_t *)a[k], i0[k], n[k], base[k], acc_dev + k);
    }
    FD_HIP_CHECK(hipGetLastError());
    unsigned long long h[3] = {0, 0, 0};
    FD_HIP_CHECK(hipMemcpyAsync(h, acc_dev, sizeof h, hipMemcpyDeviceToHost, s));
    FD_HIP_CHECK(hipStreamSynchronize(s));
    for (int k = 0; k < 3; ++k) out[k] = h[k];
    return FD_OK;
}

// the fused check (see k_fingerprint3_check): expected words already in fpx[3..5]
static int fp3_blocks_per_cu() { return 4; }      // (1 .. 16 measured in round 5: see fdjac_fingerprint3_check)
extern "C" size_t fdjac_fingerprint3_check_words(const fd_ctx *ctx) { return (size_t)kFpPart0 + (size_t)ctx->num_cus * 16 + 8; }
extern "C" int fdjac_fingerprint3_check(const fd_ctx *ctx, const void *const *a, const int *bytes, const int64_t *i0, const int64_t *n, const int64_t *base,
                                        unsigned long long *fpx, int *stale_plan, int *stale_ctx)
{
    Fp3 f;
    int total = 0;
    // a few workgroups per CU in ALL, shared out between the arrays by their bytes.  (With ONE ticket and three accumulators on one
    // line for all workgroups, N = 10^7 -- 200 MB of Int32 pattern -- took 48 / 44 / 47 / 53 / 70 / 112 us at 1 / 2 / 3 / 4 / 8 / 16
    // workgroups per CU: the arrivals queued up; hence the slots and the group tickets.)
    double all_bytes = 0;
    for (int k = 0; k < 3; ++k) all_bytes += (a[k] && n[k] > 0) ? (double)n[k] * bytes[k] : 0.0;
    const int per_cu = fp3_blocks_per_cu();
    for (int k = 0; k < 3; ++k) {
        f.a[k] = a[k]; f.bytes[k] = bytes[k]; f.i0[k] = i0[k]; f.n[k] = (a[k] && n[k] > 0) ? n[k] : 0; f.base[k] = base[k];
        const int64_t share = f.n[k] > 0 ? (int64_t)((double)ctx->num_cus * per_cu * ((double)f.n[k] * bytes[k] / all_bytes)) + 1 : 0;
        f.g[k] = f.n[k] > 0 ? (int)std::max<int64_t>(1, std::min<int64_t>((f.n[k] + 2047) / 2048, share)) : 0;
        total += f.g[k];
    }
    if (total == 0) return FD_OK;
    // beside the Jacobian, not ahead of it: the check's own stream, started behind whatever the main stream has enqueued so far (the
    // producers of the caller's arrays); checks of one context follow each other on that stream (they share tickets per plan)
    fd_ctx *c = const_cast<fd_ctx *>(ctx);
    if (!c->check_stream) {
        FD_HIP_CHECK(hipStreamCreateWithFlags(&c->check_stream, hipStreamNonBlocking));
        FD_HIP_CHECK(hipEventCreateWithFlags(&c->check_event, hipEventDisableTiming));
    }
    FD_HIP_CHECK(hipEventRecord(c->check_event, c->stream));
    FD_HIP_CHECK(hipStreamWaitEvent(c->check_stream, c->check_event, 0));
    hipLaunchKernelGGL(k_fingerprint3_check, dim3((unsigned)total), dim3(256), 0, c->check_stream, f, fpx, stale_plan, stale_ctx);
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

#else
extern "C" int fdjac_fingerprint3(const fd_ctx *ctx, const void *const *a, const int *bytes, const int64_t *i0, const int64_t *n,
                                  const int64_t *base, int memkind, unsigned long long *acc_dev, uint64_t *out);
extern "C" int fdjac_fingerprint3_check(const fd_ctx *ctx, const void *const *a, const int *bytes, const int64_t *i0, const int64_t *n, const int64_t *base,
                                        unsigned long long *fpx, int *stale_plan, int *stale_ctx);
extern "C" size_t fdjac_fingerprint3_check_words(const fd_ctx *ctx);
#endif

namespace fdjac {

static int read_index(const void *a, int bytes, int64_t i, int memkind, int64_t *v)
{
    if (memkind == FD_HOST) {
        *v = bytes == 8 ? ((const int64_t *)a)[i] : (int64_t)((const int32_t *)a)[i];
        return FD_OK;
    }
    int64_t v64 = 0;
    int32_t v32 = 0;
    if (bytes == 8) FD_HIP_CHECK(hipMemcpy(&v64, (const char *)a + 8 * (size_t)i, 8, hipMemcpyDeviceToHost));
    else FD_HIP_CHECK(hipMemcpy(&v32, (const char *)a + 4 * (size_t)i, 4, hipMemcpyDeviceToHost));
    *v = bytes == 8 ? v64 : (int64_t)v32;
    return FD_OK;
}

// The ranges a plan's fingerprints cover, in the CALLER's units (complex elements for FD_PLAN_COMPLEX_X plans).
//   idx_kind 1 (CSC): a = colptr[col0 .. col1] (col1 - col0 + 1 values), b = rowval[e0 .. e1) with e = colptr - base
//   idx_kind 2 (index lists): a = rows_index[0 .. nnz), b = cols_index[0 .. nnz)
//   idx_kind 0: the pattern is structural (Tridiagonal, BandedMatrix, BlockBandedMatrix, the dense arm): colours only
static int fingerprint_now(fd_plan *p, const fd_pattern_arrays *now, int64_t a0, int64_t an, uint64_t h[3], int64_t *b0_out, int64_t *bn_out)
{
    const fd_fingerprint &fp = p->fp;
    const void *arr[3] = {fp.idx_kind ? now->idx_a : nullptr, fp.idx_kind ? now->idx_b : nullptr, now->colorvec};
    int bytes[3] = {now->idx_bytes, now->idx_bytes, now->color_bytes};
    int64_t i0[3] = {a0, 0, 0}, n[3] = {arr[0] ? an : 0, 0, arr[2] ? now->len_color : 0}, base[3] = {now->idx_base, now->idx_base, 0};
    int64_t b0 = 0, bn = 0;
    if (fp.idx_kind == 1 && fp.valid) {
        // comparing with a recorded plan: its own rowval range.  (colptr's fingerprint is position-dependent and covers both end
        // points: if it matches, the caller's range IS this one; if not, the answer is "no" whatever rowval holds -- so the two
        // blocking reads of colptr's end points are only needed when the range is first recorded)
        b0 = fp.b0;
        bn = fp.bn;
        if (b0 + bn > now->len_b) { b0 = 0; bn = -1; }
    } else if (fp.idx_kind == 1 && arr[0]) {
        int64_t e0 = 0, e1 = 0;
        int rc = read_index(arr[0], bytes[0], a0, now->memkind, &e0);
        if (!rc) rc = read_index(arr[0], bytes[0], a0 + an - 1, now->memkind, &e1);
        if (rc) return rc;
        b0 = e0 - now->idx_base;
        bn = e1 - e0;
        if (b0 < 0 || bn < 0 || b0 + bn > now->len_b) { b0 = 0; bn = -1; }   // cannot be the plan's pattern (reported as a mismatch)
    } else if (fp.idx_kind == 2) {
        bn = now->len_b;
    }
    i0[1] = b0;
    n[1] = (arr[1] && bn > 0) ? bn : 0;
    *b0_out = b0;
    *bn_out = bn;
    if (now->memkind == FD_DEVICE && !p->d_fp) FD_HIP_CHECK(hipMalloc((void **)&p->d_fp, 3 * sizeof(unsigned long long)));
    return fdjac_fingerprint3(p->ctx, arr, bytes, i0, n, base, now->memkind, p->d_fp, h);
}

static int check_arrays(const fd_pattern_arrays *now)
{
    FD_REQUIRE(now != nullptr, FD_ERR_ARG, "the array description is NULL");
    FD_REQUIRE(now->memkind == FD_HOST || now->memkind == FD_DEVICE, FD_ERR_ARG, "memkind must be FD_HOST or FD_DEVICE");
    FD_REQUIRE((!now->idx_a && !now->idx_b) || now->idx_bytes == 4 || now->idx_bytes == 8, FD_ERR_ARG, "idx_bytes must be 4 or 8");
    FD_REQUIRE(now->idx_base == 0 || now->idx_base == 1, FD_ERR_ARG, "idx_base must be 0 or 1");
    FD_REQUIRE(!now->colorvec || now->color_bytes == 4 || now->color_bytes == 8, FD_ERR_ARG, "color_bytes must be 4 or 8");
    FD_REQUIRE(now->len_a >= 0 && now->len_b >= 0 && now->len_color >= 0, FD_ERR_ARG, "negative array length");
    return FD_OK;
}

// Called by the public fd_plan_create_* functions after a plan was built from `src` with FD_PLAN_FINGERPRINT set.
int plan_record_fingerprint(fd_plan *p, int idx_kind, const fd_pattern_arrays *src, int64_t col0, int64_t col1)
{
    int rc = check_arrays(src);
    if (rc) return rc;
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    fd_fingerprint &fp = p->fp;
    fp = fd_fingerprint();
    fp.idx_kind = idx_kind;
    fp.len_a = src->len_a;
    fp.len_b = src->len_b;
    fp.len_color = src->len_color;
    fp.a0 = idx_kind == 1 ? col0 : 0;
    fp.an = idx_kind == 1 ? col1 - col0 + 1 : (idx_kind == 2 ? src->len_a : 0);
    uint64_t h[3];
    rc = fingerprint_now(p, src, fp.a0, fp.an, h, &fp.b0, &fp.bn);
    if (rc) return rc;
    FD_REQUIRE(fp.bn >= 0, FD_ERR_SHAPE, "colptr does not describe a range of rowval");
    fp.h_a = h[0];
    fp.h_b = h[1];
    fp.h_color = h[2];
    fp.valid = true;
    return FD_OK;
}

}  // namespace fdjac

extern "C" int fd_plan_matches(fd_plan *p, const fd_pattern_arrays *now, int *matches_out)
{
    using namespace fdjac;
    FD_REQUIRE(p && matches_out, FD_ERR_ARG, "NULL argument");
    *matches_out = 0;
    int rc = check_arrays(now);
    if (rc) return rc;
    const fd_fingerprint &fp = p->fp;
    FD_REQUIRE(fp.valid, FD_ERR_UNSUPPORTED, "the plan was created without FD_PLAN_FINGERPRINT");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    // lengths first: a resized array cannot be the plan's (and must not be read with the plan's ranges)
    if (now->colorvec && now->len_color != fp.len_color) return FD_OK;
    if (fp.idx_kind && now->idx_a && now->len_a != fp.len_a) return FD_OK;
    if (fp.idx_kind == 2 && now->idx_b && now->len_b != fp.len_b) return FD_OK;   // (a CSC rowval may be longer than nnz: bounds only)
    uint64_t h[3];
    int64_t b0 = 0, bn = 0;
    rc = fingerprint_now(p, now, fp.a0, fp.an, h, &b0, &bn);
    if (rc) return rc;
    bool same = true;
    if (now->colorvec) same = same && h[2] == fp.h_color;
    if (fp.idx_kind && now->idx_a) same = same && h[0] == fp.h_a && (fp.idx_kind != 1 || (b0 == fp.b0 && bn == fp.bn));
    if (fp.idx_kind && now->idx_b) same = same && h[1] == fp.h_b;      // (a CSC plan knows its rowval range: compared with or without colptr)
    *matches_out = same ? 1 : 0;
    return FD_OK;
}

// the deferred form: one fused launch, the verdict raised on the device (see k_fingerprint3_check)
extern "C" int fd_plan_matches_async(fd_plan *p, const fd_pattern_arrays *now)
{
    using namespace fdjac;
    FD_REQUIRE(p != nullptr, FD_ERR_ARG, "plan is NULL");
    int rc = check_arrays(now);
    if (rc) return rc;
    FD_REQUIRE(now->memkind == FD_DEVICE, FD_ERR_ARG, "fd_plan_matches_async compares device arrays (host arrays: fd_plan_matches)");
    const fd_fingerprint &fp = p->fp;
    FD_REQUIRE(fp.valid, FD_ERR_UNSUPPORTED, "the plan was created without FD_PLAN_FINGERPRINT");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    fd_ctx *ctx = p->ctx;
    if (!ctx->h_stale) {
        FD_HIP_CHECK(hipHostMalloc((void **)&ctx->h_stale, sizeof(int), hipHostMallocMapped));
        *ctx->h_stale = 0;
        FD_HIP_CHECK(hipHostGetDevicePointer((void **)&ctx->d_stale, ctx->h_stale, 0));
    }
    if (!p->d_fpx) {
        FD_HIP_CHECK(hipHostMalloc((void **)&p->h_pstale, sizeof(int), hipHostMallocMapped));
        *p->h_pstale = 0;
        FD_HIP_CHECK(hipHostGetDevicePointer((void **)&p->d_pstale, p->h_pstale, 0));
        const size_t words = fdjac_fingerprint3_check_words(ctx);
        FD_HIP_CHECK(hipMalloc((void **)&p->d_fpx, words * sizeof(unsigned long long)));
        FD_HIP_CHECK(hipMemsetAsync(p->d_fpx, 0, words * sizeof(unsigned long long), ctx->stream));      // (tickets start at 0 and reset themselves)
        const unsigned long long init[8] = {0, 0, 0, fp.h_a, fp.h_b, fp.h_color, 0, 0};
        FD_HIP_CHECK(hipMemcpyAsync(p->d_fpx, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
        FD_HIP_CHECK(hipStreamSynchronize(ctx->stream));     // (once per plan: `init` lives on this stack)
    }
    // what the host can see at once: a resized array cannot be the plan's (and must not be read with the plan's ranges)
    bool host_mismatch = (now->colorvec && now->len_color != fp.len_color) || (fp.idx_kind && now->idx_a && now->len_a != fp.len_a) ||
                         (fp.idx_kind == 2 && now->idx_b && now->len_b != fp.len_b) ||
                         (fp.idx_kind == 1 && now->idx_b && fp.b0 + fp.bn > now->len_b);
    if (host_mismatch) {
        *p->h_pstale = 1;
        set_error("the arrays have other lengths than the ones this plan was compiled from");
        return FD_ERR_STALE;
    }
    const bool use_a = fp.idx_kind && now->idx_a, use_b = fp.idx_kind && now->idx_b;
    const void *arr[3] = {use_a ? now->idx_a : nullptr, use_b ? now->idx_b : nullptr, now->colorvec};
    const int bytes[3] = {now->idx_bytes, now->idx_bytes, now->color_bytes};
    const int64_t i0[3] = {fp.a0, fp.idx_kind == 1 ? fp.b0 : 0, 0};
    const int64_t n[3] = {use_a ? fp.an : 0, use_b ? (fp.idx_kind == 1 ? fp.bn : now->len_b) : 0, now->colorvec ? now->len_color : 0};
    const int64_t base[3] = {now->idx_base, now->idx_base, 0};
    return fdjac_fingerprint3_check(ctx, arr, bytes, i0, n, base, p->d_fpx, p->d_pstale, ctx->d_stale);
}

extern "C" int fd_plan_stale(fd_plan *p, int *stale_out)
{
    FD_REQUIRE(p && stale_out, FD_ERR_ARG, "NULL argument");
    *stale_out = p->h_pstale ? *(volatile int *)p->h_pstale : 0;
    return FD_OK;
}
