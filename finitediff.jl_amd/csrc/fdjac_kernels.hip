// Hand-written gfx950 (CDNA4, wave64) kernels of the coloured-Jacobian hot path.
//
// Every stage is HBM-bandwidth bound (no dense contraction => no MFMA).  Layout rules:
//   * all streams are read / written with 16 B per lane where alignment allows (r2_t),
//   * a workgroup (256 threads = 4 waves) owns a contiguous tile of its stream so that the
//     lines it gathers from the batched f! outputs are not shared with other XCDs' L2s,
//   * the step sizes eps[] of the current colour chunk are staged in LDS.
//
// Reference loops each kernel replaces are cited per kernel (paths relative to /root/reference).
#include <cstdlib>
#include <type_traits>

#include "fdjac_internal.h"
#include "fdjac_eps_dev.h"

namespace fdjac {

typedef r2_t d2_t;   // a pair of elements (16 B for Float64)

template <typename CT> struct ColorTraits;
// "none": the column has no colour (stored entries are written as 0); "pad": not a stored entry at all
// (index lists are padded to a whole number of tiles so the hot loads need no bounds checks).
template <> struct ColorTraits<uint8_t> { static constexpr int none = 0xFF; static constexpr int pad = 0xFE; };
template <> struct ColorTraits<int32_t> { static constexpr int none = -1; static constexpr int pad = -2; };

// ---------------------------------------------------------------------------------------------
// K1a  masked sums of squares for ALL colours in one pass over x (C <= kRegColors).
//   Replaces, for every colour at once:  @. x2 = x1*(_color==color_i); tmp = norm(x2)
//   (src/jacobians.jl:559-560, 600-601).  Deterministic: fixed per-thread order, fixed
//   shuffle tree, per-block partials reduced by k_eps_finalize in fixed order.
//   partial layout: partial[block * ldp + c].
// ---------------------------------------------------------------------------------------------
// CYC: the colours are cyclic, color[j] == (j + shift) mod C for every column (found at plan time: tridiagonal /
//   banded patterns coloured with mod1(j, C)) -- the kernel computes them and reads x alone (8 B per column instead of 9).
// NT:  non-temporal loads of x.  In a steady-state loop the pass follows the previous call's nzval stores; plain
//   loads allocate in the Infinity Cache and pay for the write-back of as many dirty lines as they bring in
//   (scripts/ubench/eps_mall_probe.hip: 26.6 us after 240 MB of stores, 21.2 us after reads only, 21.4 us with
//   non-temporal loads after stores).  Same values, same summation order => same bits in all four variants.
// The reduction is DEFINED as a rank-aligned two-level sum (round 5), a function of N alone:
//   level 0  x is cut into kEpsGroups = 64 contiguous groups of tpg tiles (tile = 2048 elements); a group into bpg blocks of tpb
//            consecutive tiles; a block sums its tiles per thread, then the fixed shuffle tree, then its 4 waves in order;
//   level 1  group sum[g][c]  = the group's bpg block sums added in block order;
//   level 2  S[c] = the 64 group sums added in group order;  eps[c] from S[c] (eps_rule).
// A rank of a W-rank job owns whole groups (ceil(64 / W) each): it exchanges 64 / W x C doubles instead of every block's
// partial sums, and any split gives the bits of the unsharded call.  The levels run inside THIS launch: the last block of a
// group to arrive (agent-scope ticket) adds the group's block sums, the last group to finish adds the group sums and writes
// eps -- no finalize launch (ONE ticket for the whole launch was measured too: 1024 arrivals on one address cost 6 us more than
// 64 x 16 + 64).  Hand-off between workgroups: 8-byte agent-scope atomic stores and loads on both sides with
// the stores drained before the ticket (MI355X_MICROARCH.md, inter-workgroup visibility); which block does the adding never
// changes what is added in which order.  final_groups = 0: stop after level 1 (a shard: its group sums are exchanged).
template <typename CT, int NC, bool CYC, bool NT>
__global__ void __launch_bounds__(kBlock)
k_eps_partial_reg(const real_t *__restrict__ x, const CT *__restrict__ color, int64_t n,
                  double *__restrict__ partial, double *__restrict__ gsum, unsigned *__restrict__ tick, int ldp, int cyc_C, int cyc_shift,
                  int block_off, EpsGrid eg, int pair, real_t *__restrict__ eps, real_t *__restrict__ eps2)
{
    const int gblock = (int)blockIdx.x + block_off;
    const int grp = gblock / eg.bpg;
    __shared__ double red[kBlock / 64][NC];
    double s;
    if (!eps_block_sum<CT, NC, CYC, NT>(x, color, n, cyc_C, cyc_shift, gblock, eg, pair, red, s)) return;
    const int lane = threadIdx.x & 63;
    // ---- level 0 result of this block, then the hand-off (wave 0 only; lanes < NC carry one colour each) ----
    if (lane < NC) __hip_atomic_store(partial + (int64_t)gblock * ldp + lane, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the block sums have left this CU before the ticket is drawn
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(tick + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    if (t != (unsigned)(eg.bpg - 1)) return;
    // ---- level 1: this block arrived last in its group -- every block sum of the group is in memory ----
    // (the loads of all addends are issued together, one or a few per lane, and parked in LDS; lane c then adds its colour's in the
    //  defined order -- a chain of dependent agent-scope loads would cost one memory round trip per addend)
    __shared__ double stage[kEpsGroups * kRegColors];
    {
        const double *pg = partial + (int64_t)grp * eg.bpg * ldp;
        const int cnt = eg.bpg * ldp;                          // <= 16 x 8 doubles
        constexpr int kLv = kEpsBlocksPerGroup * kRegColors / 64;      // 8 loads per lane, all in flight together
        double v[kLv];
#pragma unroll
        for (int u = 0; u < kLv; ++u) v[u] = (lane + 64 * u < cnt) ? __hip_atomic_load(pg + lane + 64 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
        for (int u = 0; u < kLv; ++u) stage[lane + 64 * u] = v[u];
    }
    __builtin_amdgcn_wave_barrier();     // (one wave: its LDS instructions execute in order)
    if (lane < NC) {
        double gs = 0.0;
        for (int k = 0; k < eg.bpg; ++k) gs += stage[k * ldp + lane];
        __hip_atomic_store(gsum + (int64_t)grp * ldp + lane, gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) __hip_atomic_store(tick + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (launches on the stream are ordered)
    if (eg.final_groups <= 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned t2 = 0;
    if (lane == 0) t2 = __hip_atomic_fetch_add(tick + kEpsGroups, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t2 = (unsigned)__builtin_amdgcn_readfirstlane((int)t2);
    if (t2 != (unsigned)(eg.final_groups - 1)) return;
    // ---- level 2: the last group -- every group sum is in memory ----
    __builtin_amdgcn_wave_barrier();
    {
        double v[kRegColors];
#pragma unroll
        for (int u = 0; u < kRegColors; ++u) v[u] = __hip_atomic_load(gsum + u * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < kRegColors; ++u) stage[u * 64 + lane] = v[u];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < NC) {
        double tot = 0.0;
        for (int g = 0; g < kEpsGroups; ++g) tot += stage[g * ldp + lane];
        if (lane < eg.C) {
            const real_t e = eps_rule<real_t>(tot, eg.relstep, eg.absstep, eg.dir, eg.is_forward);
            eps[lane] = e;
            if (eps2) eps2[lane] = (real_t)2 * e;           // (central differences handed over as f(+) - f(-), see launch_scale)
        }
    }
    if (lane == 0) __hip_atomic_store(tick + kEpsGroups, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same reduction with the hand-offs of the fused step (fdjac_eps_dev.h) instead of tickets: blockIdx c < C is the FINISHER of
// colour c, the rest are the reduction's workgroups; a block sum is published by ONE atomic store into a slot that held a sentinel,
// the finisher polls the colour's slots with dense loads, adds levels 1 and 2 in the defined order and writes eps[c].  Two trips
// through memory from the last block sum to the step size instead of the ticket chain's six (N = 10^6: 12.3 -> us, N = 10^7: 21.2 -> us).
// Finishers have the lowest block indices and wait only for workgroups that never wait: no deadlock whatever is resident.
template <typename CT, int NC, bool CYC, bool NT>
__global__ void __launch_bounds__(kBlock)
k_eps_flags(const real_t *__restrict__ x, const CT *__restrict__ color, int64_t n, FusedEps fz)
{
    __shared__ __attribute__((aligned(16))) double s_lds[kFzMaxBlocks + 2 * kEpsGroups + 2];
    const int b = (int)blockIdx.x;
    if (b < fz.eg.C) { fused_finisher(fz, b, s_lds); return; }
    fused_eps_block<NC, true, CT, CYC, NT>(x, n, fz, b - fz.eg.C, reinterpret_cast<double (*)[NC]>(s_lds), color);
}

// level 2 on its own: a sharded reduction's group sums have been exchanged (fd_plan_eps_finalize; with RCCL after the all-gather)
__global__ void __launch_bounds__(64)
k_eps_final(const double *__restrict__ gsum, int ldp, int C, double relstep, double absstep, double dir, int is_forward,
            real_t *__restrict__ eps, real_t *__restrict__ eps2)
{
    const int c = threadIdx.x;
    if (c >= C) return;
    double tot = 0.0;
    for (int g = 0; g < kEpsGroups; ++g) tot += gsum[(int64_t)g * ldp + c];
    const real_t e = eps_rule<real_t>(tot, relstep, absstep, dir, is_forward);
    eps[c] = e;
    if (eps2) eps2[c] = (real_t)2 * e;
}

// K1b  same reduction for many colours: block (chunk k, colour c) walks colour c's column list
//   (perm sorted by colour, built once at plan time) -- deterministic for any C.
__global__ void __launch_bounds__(kBlock)
k_eps_partial_seg(const real_t *__restrict__ x, const int32_t *__restrict__ perm,
                  const int64_t *__restrict__ cptr, int64_t C, int nchunks,
                  double *__restrict__ partial, int pair)
{
    const int64_t b = blockIdx.x;
    const int64_t c = b / nchunks;
    const int k = (int)(b - c * nchunks);
    const int64_t lo = cptr[c], hi = cptr[c + 1];
    const int64_t len = hi - lo;
    const int64_t per = (len + nchunks - 1) / nchunks;
    const int64_t s = lo + per * k;
    const int64_t e = (s + per < hi) ? s + per : hi;
    double acc = 0.0;
    for (int64_t i = s + threadIdx.x; i < e; i += kBlock) {
        const double v = (double)x[perm[i]];
        acc += v * v;
        if (pair) { const double w = (double)x[perm[i] + 1]; acc += w * w; }   // complex-valued x: + im^2
    }
    __shared__ double red[kBlock / 64];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) t += red[w];
        partial[(int64_t)k * C + c] = t;
    }
}

// K1c  eps[c] from the partial sums.  One block per colour.
//   forward: max(relstep*abs(sqrt(norm)), absstep)*dir   (src/epsilons.jl:26-29; the sqrt of the
//            2-norm is src/jacobians.jl:561)
//   central: max(relstep*abs(sqrt(norm)), absstep)       (src/epsilons.jl:50-53; jacobians.jl:602)
__global__ void __launch_bounds__(kBlock)
k_eps_finalize(const double *__restrict__ partial, int nparts, int ldp, double relstep,
               double absstep, double dir, int is_forward, real_t *__restrict__ eps, real_t *__restrict__ eps2)
{
    const int c = blockIdx.x;
    double acc = 0.0;
    for (int k = threadIdx.x; k < nparts; k += kBlock) acc += partial[(int64_t)k * ldp + c];
    __shared__ double red[kBlock / 64];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; ++w) t += red[w];
        const real_t e = eps_rule<real_t>(t, relstep, absstep, dir, is_forward);
        eps[c] = e;
        if (eps2) eps2[c] = (real_t)2 * e;                  // (central differences handed over as f(+) - f(-), see launch_scale)
    }
}

// K1d  small problems (N <= kSmallN, C <= kRegColors, one colour chunk): ONE single-workgroup launch does the masked
//   sums of squares, the step sizes, and -- unless a lazy launcher perturbs on its own -- the perturbed points of
//   every colour plus (forward differences without f_in) a copy of x as one more batch member, so that f(x) rides in
//   the same f! launch.  A Jacobian is then three launches (this, f!, decompression) instead of five or six; below
//   ~10^5 unknowns the call is launch-latency bound (scripts/latency_probe.py).  Same per-element arithmetic as
//   k_eps_partial_reg / k_eps_finalize / k_perturb; the summation tree is this kernel's own (deterministic).
//   PMODE: -1 step sizes only, 0 forward, 1 central, 2 complex-step points.
constexpr int kSmallBlock = 1024;
template <typename CT, int NC, int PMODE>
__global__ void __launch_bounds__(kSmallBlock)
k_eps_perturb_small(const real_t *__restrict__ x, const CT *__restrict__ color, int64_t n, double relstep,
                    double absstep, double dir, int is_forward, int C, real_t *__restrict__ eps,
                    real_t *__restrict__ X, int64_t ldx, int base_row)
{
    __shared__ double red[kSmallBlock / 64][NC];
    __shared__ real_t s_eps[NC];
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    for (int64_t i = (int64_t)threadIdx.x * 2; i < n; i += (int64_t)kSmallBlock * 2) {
        r2_t v;
        int c0, c1 = -2;
        if (i + 1 < n) {
            v = *reinterpret_cast<const r2_t *>(x + i);
            load_color_pair<CT>(color + i, c0, c1);
        } else {
            v = r2_t{x[i], 0};
            c0 = color[i];
        }
        const double s0 = (double)v.x * (double)v.x, s1 = (double)v.y * (double)v.y;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            acc[c] += (c0 == c) ? s0 : 0.0;
            acc[c] += (c1 == c) ? s1 : 0.0;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double t = wave_sum(acc[c]);
        if (lane == 0) red[wave][c] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < NC) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kSmallBlock / 64; ++w) t += red[w][threadIdx.x];
        const real_t nrm = (real_t)sqrt(t);
        const real_t xs = fabs(sqrt(nrm));
        const real_t a = (real_t)relstep * xs;
        real_t e = (a > (real_t)absstep) ? a : (real_t)absstep;
        if (is_forward) e = e * (real_t)dir;
        s_eps[threadIdx.x] = e;
        if ((int)threadIdx.x < C) eps[threadIdx.x] = e;
    }
    if (PMODE < 0) return;
    __syncthreads();
    const int B = C;
    for (int64_t j = (int64_t)threadIdx.x * 2; j < n; j += (int64_t)kSmallBlock * 2) {
        const bool pair = j + 1 < n;
        real_t v0, v1 = 0;
        int c0, c1 = -1;
        if (pair) {
            const r2_t v = *reinterpret_cast<const r2_t *>(x + j);
            v0 = v.x; v1 = v.y;
            load_color_pair<CT>(color + j, c0, c1);
        } else {
            v0 = x[j];
            c0 = (int)color[j];
        }
        for (int b = 0; b < B; ++b) {
            const real_t e = s_eps[b];
            const real_t e0 = (c0 == b) ? e : (real_t)0, e1 = (c1 == b) ? e : (real_t)0;
            if (PMODE == 2) {
                real_t *dst = X + ((int64_t)b * ldx + j) * 2;
                *reinterpret_cast<r2_t *>(dst) = r2_t{v0, e0};
                if (pair) *reinterpret_cast<r2_t *>(dst + 2) = r2_t{v1, e1};
            } else {
                real_t *dp = X + (int64_t)b * ldx + j;
                if (pair) *reinterpret_cast<r2_t *>(dp) = r2_t{v0 + e0, v1 + e1};
                else dp[0] = v0 + e0;
                if (PMODE == 1) {
                    real_t *dm = X + (int64_t)(B + b) * ldx + j;
                    if (pair) *reinterpret_cast<r2_t *>(dm) = r2_t{v0 - e0, v1 - e1};
                    else dm[0] = v0 - e0;
                }
            }
        }
        if (base_row >= 0) {   // x itself as one more point of the batch: f(x) comes out of the same f! launch
            real_t *dp = X + (int64_t)base_row * ldx + j;
            if (pair) *reinterpret_cast<r2_t *>(dp) = r2_t{v0, v1};
            else dp[0] = v0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2  perturbed points for a whole chunk of colours, written from the pristine x:
//   forward  X[b][j]   = x[j] + eps_b*(color[j]==b)                 (src/jacobians.jl:562)
//   central  X[b][j]   = x[j] + eps_b*mask ; X[B+b][j] = x[j] - eps_b*mask   (:603-604)
//   complex  X[b][j]   = (x[j], eps*mask)                                     (:633)
//   One pass over x and colour; no un-perturb pass is needed (:584,:619-620,:646) because x is
//   never modified.  MODE 0/1/2 as fd_fdtype.
// ---------------------------------------------------------------------------------------------
template <typename CT, int MODE>
__global__ void __launch_bounds__(kBlock)
k_perturb(const real_t *__restrict__ x, const CT *__restrict__ color,
          const real_t *__restrict__ eps, int c_lo, int B, int64_t j0, int64_t j1,
          real_t *__restrict__ X, int64_t ldx)
{
    // j0 is even by construction (host rounds the window down), so pairs are 16-B aligned.
    const int64_t stride = (int64_t)gridDim.x * kBlock * 2;
    for (int64_t j = j0 + ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 2; j < j1; j += stride) {
        const bool pair = (j + 1 < j1);
        real_t v0, v1 = 0.0;
        if (pair) {
            const r2_t v = *reinterpret_cast<const r2_t *>(x + j);
            v0 = v.x; v1 = v.y;
        } else {
            v0 = x[j];
        }
        // colours outside the chunk (and "none") never equal a b in [0,B)
        int c0, c1 = -1;
        if (pair) {
            load_color_pair<CT>(color + j, c0, c1);
            c1 -= c_lo;
        } else {
            c0 = (int)color[j];
        }
        c0 -= c_lo;
        for (int b = 0; b < B; ++b) {
            const real_t e = eps[c_lo + b];
            const real_t e0 = (c0 == b) ? e : 0.0, e1 = (c1 == b) ? e : 0.0;
            if (MODE == 2) {
                real_t *dst = X + ((int64_t)b * ldx + j) * 2;
                *reinterpret_cast<r2_t *>(dst) = r2_t{v0, e0};
                if (pair) *reinterpret_cast<r2_t *>(dst + 2) = r2_t{v1, e1};
            } else {
                real_t *dp = X + (int64_t)b * ldx + j;
                if (pair) *reinterpret_cast<r2_t *>(dp) = r2_t{v0 + e0, v1 + e1};
                else dp[0] = v0 + e0;
                if (MODE == 1) {
                    real_t *dm = X + (int64_t)(B + b) * ldx + j;
                    if (pair) *reinterpret_cast<r2_t *>(dm) = r2_t{v0 - e0, v1 - e1};
                    else dm[0] = v0 - e0;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// value of one stored entry: the fused difference (src/jacobians.jl:565 / 607 / 635) evaluated
// only at the rows the pattern needs, for the colour that owns the entry's column.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ real_t entry_value(const real_t *__restrict__ FXa,
                                              const real_t *__restrict__ FXb, int64_t ld, int cb,
                                              int64_t r, real_t e)
{
    if (MODE == 0) {
        const real_t a = FXa[(int64_t)cb * ld + r];
        const real_t b = FXb[r];
        return (a - b) / e;
    } else if (MODE == 1) {
        const real_t a = FXa[(int64_t)cb * ld + r];
        const real_t b = FXb[(int64_t)cb * ld + r];
        return (a - b) / (2 * e);
    } else {
        const real_t a = FXa[((int64_t)cb * ld + r) * 2 + 1];
        return a / e;
    }
}

// K3  fused difference + decompression in STORAGE ORDER for index-list patterns (CSC nzval, or
//   dense J through dest[]).  For every stored entry p:
//       nzval[p] = (fx1_c[rowval[p]] - fx[rowval[p]]) / eps_c ,  c = colour of p's column
//   which is `_colorediteration!` (ext/FiniteDiffSparseArraysExt.jl:38-47; :20-28 with dest;
//   src/iteration_utils.jl:25-32 for COO lists) for all colours of the chunk at once: rowval,
//   the per-entry colour and nzval are each streamed exactly once, fully coalesced, and the
//   batched f! outputs are gathered near-sequentially (banded patterns) out of L2.
//   LDS stages the chunk's eps[] slice.
template <typename CT, int MODE, bool HAS_DEST, int U, bool LDS_EPS>
__global__ void __launch_bounds__(kBlock)
k_decompress_list(const int32_t *__restrict__ rowval, const CT *__restrict__ nzcolor,
                  const int64_t *__restrict__ dest, const real_t *__restrict__ FXa,
                  const real_t *__restrict__ FXb, int64_t ld, const real_t *__restrict__ eps,
                  int c_lo, int c_hi, real_t *__restrict__ out, int64_t n, int vec_ok)
{
    extern __shared__ real_t s_eps[];
    const int nB = c_hi - c_lo;
    if (LDS_EPS) {
        for (int c = threadIdx.x; c < nB; c += kBlock) s_eps[c] = eps[c_lo + c];
        __syncthreads();
    }
    const int none = ColorTraits<CT>::none;
    // block tile = U*512 stored entries; pair u of thread t sits at tile + u*512 + 2t, so every
    // wave instruction (index load, colour load, value store) touches one dense 512-B / 1-KiB run.
    const int64_t ntiles = (n + (U * kBlock * 2) - 1) / (U * kBlock * 2);
    const int64_t xt = xcd_tile(blockIdx.x, ntiles);   // XCD x walks its own contiguous range of tiles
    if (xt >= ntiles) return;
    const int64_t tile_id = (vec_ok & 4) ? ntiles - 1 - xt : xt;   // reversed tile order, see tile_order_reversed()
    const int64_t t0 = tile_id * (U * kBlock * 2);

    // phase 1: indices and colours (independent coalesced loads; the lists are padded to whole tiles)
    int r[2 * U], c[2 * U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t p = t0 + (int64_t)u * kBlock * 2 + threadIdx.x * 2;
        const int2 rr = *reinterpret_cast<const int2 *>(rowval + p);
        r[2 * u] = rr.x; r[2 * u + 1] = rr.y;
        load_color_pair<CT>(nzcolor + p, c[2 * u], c[2 * u + 1]);
    }
    // phase 2: branch-free gathers.  Entries that will not be written (another chunk's colour, a
    // column without colour, past the end) gather from slot 0 / their own valid row and are discarded.
    real_t a[2 * U], b[2 * U], e[2 * U];
    bool valid[2 * U];
#pragma unroll
    for (int k = 0; k < 2 * U; ++k) {
        const int cb = c[k] - c_lo;
        valid[k] = (c[k] != none) & (c[k] != ColorTraits<CT>::pad) & ((unsigned)cb < (unsigned)nB);
        const int cs = valid[k] ? cb : 0;
        const int64_t at = (int64_t)cs * ld + r[k];
        e[k] = LDS_EPS ? s_eps[cs] : eps[c_lo + cs];
        if (MODE == 0) { a[k] = FXa[at]; b[k] = FXb[r[k]]; }
        else if (MODE == 1) { a[k] = FXa[at]; b[k] = FXb[at]; }
        else { a[k] = FXa[at * 2 + 1]; b[k] = 0.0; }
    }
    // phase 3: the difference (src/jacobians.jl:565 / 607 / 635), IEEE division
    real_t v[2 * U];
#pragma unroll
    for (int k = 0; k < 2 * U; ++k) {
        real_t q;
        if (MODE == 0) q = (a[k] - b[k]) / e[k];
        else if (MODE == 1) q = (a[k] - b[k]) / (2 * e[k]);
        else q = a[k] / e[k];
        v[k] = valid[k] ? q : 0.0;
    }
    // phase 4: stores.  valid -> value; column without colour -> 0 (first chunk only); else untouched.
    // The whole wave takes the 16-B store path except at the array end / uncoloured columns / chunked runs.
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t p = t0 + (int64_t)u * kBlock * 2 + threadIdx.x * 2;
        const bool w0 = valid[2 * u] | ((c[2 * u] == none) & (c_lo == 0));
        const bool w1 = valid[2 * u + 1] | ((c[2 * u + 1] == none) & (c_lo == 0));
        real_t q0 = v[2 * u], q1 = v[2 * u + 1];
        asm volatile("" : "+v"(q0), "+v"(q1));  // keep the divisions above the branch (no sinking / duplication)
        if (HAS_DEST) {
            if (w0) out[dest[p]] = q0;
            if (w1) out[dest[p + 1]] = q1;
        } else {
            const bool both = w0 & w1 & ((vec_ok & 1) != 0);
            if (__builtin_amdgcn_ballot_w64(both) == __builtin_amdgcn_ballot_w64(true)) {
                d2_t pk = {q0, q1};
                *reinterpret_cast<d2_t *>(out + p) = pk;
            } else {
                if (w0) out[p] = q0;
                if (w1) out[p + 1] = q1;
            }
        }
    }
}

// K3b  the same decompression for SCATTERED patterns (e.g. 5-point stencils, where one wave of
//   storage-ordered entries gathers from ~40 different 128-B lines per instruction and the kernel
//   becomes texture-addresser bound).  At plan time every tile of kSortTile entries is reordered by
//   colour, so a wave's gathers walk one batched f! output array with near-consecutive rows; the
//   values are then scattered to their storage position *inside LDS* and written to nzval with
//   dense 16-B stores.  This is the north-star's "LDS staging + sparse scatter", done per tile.
//   spos[q] = position (0..kSortTile-1) of sorted entry q inside its tile.
//   ALLW: every entry of every tile is written (single colour chunk, no uncoloured column) -> no flags.
//   FXL (forward differences): f(x) is not gathered.  The rows a tile touches lie in a few runs (plan time: at most kFxWin
//   windows, kFxRows rows in total -- a band tile has one, a 3-D stencil tile five); the workgroup loads those runs of f(x)
//   densely into LDS (the area the values are transposed in afterwards) and every entry looks its row up there: one global
//   gather per entry instead of two.  A tile whose rows do not fit (fxwin[0] < 0) gathers as before.
template <typename CT, int MODE, bool LDS_EPS, bool ALLW, bool SORTED, bool FXL = false>
__global__ void __launch_bounds__(kBlock)
k_decompress_sorted(const int32_t *__restrict__ srow, const CT *__restrict__ scol,
                    const uint16_t *__restrict__ spos, const real_t *__restrict__ FXa,
                    const real_t *__restrict__ FXb, int64_t ld, const real_t *__restrict__ eps, int c_lo,
                    int c_hi, real_t *__restrict__ out, int64_t n, int vec_ok, const int32_t *__restrict__ tile_order,
                    const int32_t *__restrict__ fxwin)
{
    constexpr int E = kSortTile / kBlock;   // entries per thread (8); entry e of thread t is tile + e*256 + t,
                                            // so one wave-level gather covers 64 CONSECUTIVE sorted entries
    extern __shared__ real_t s_mem[];
    real_t *s_val = s_mem;                                             // kSortTile values
    uint8_t *s_flag = reinterpret_cast<uint8_t *>(s_mem + kSortTile);  // kSortTile flags (ALLW: unused)
    const int nB = c_hi - c_lo;
    real_t *s_eps = s_mem + kSortTile + (ALLW ? 0 : kSortTile / sizeof(real_t));   // flags: one byte per entry
    if (LDS_EPS)
        for (int c = threadIdx.x; c < nB; c += kBlock) s_eps[c] = eps[c_lo + c];
    const int none = ColorTraits<CT>::none;
    const int64_t ntiles = (n + kSortTile - 1) / kSortTile;
    const int64_t xt = xcd_tile(blockIdx.x, ntiles);
    if (xt >= ntiles) return;
    const int64_t rank = (vec_ok & 4) ? ntiles - 1 - xt : xt;      // reversed tile order, see tile_order_reversed()
    const int64_t tile_id = tile_order ? (int64_t)tile_order[rank] : rank;   // (far-band patterns: region by region across the planes)
    const int64_t t0 = tile_id * kSortTile;
    if (LDS_EPS) __syncthreads();

    int r[E], c[E], pos[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int lq = k * kBlock + threadIdx.x;
        r[k] = srow[t0 + lq];
        c[k] = (int)scol[t0 + lq];
        pos[k] = SORTED ? (int)spos[t0 + lq] : lq;
    }
    real_t a[E], b[E], e[E];
    bool valid[E];
    bool fxl = false;
    if constexpr (FXL && MODE == 0) {
        // the tile's runs of f(x) into LDS (s_val's area: it is written only after every lane has taken its f(x) values out)
        const int32_t *fw = fxwin + tile_id * (2 * kFxWin);
        int ws[kFxWin], wl[kFxWin];
#pragma unroll
        for (int w = 0; w < kFxWin; ++w) { ws[w] = fw[2 * w]; wl[w] = fw[2 * w + 1]; }
        fxl = ws[0] >= 0;
        if (fxl) {
            int wo = 0;
#pragma unroll
            for (int w = 0; w < kFxWin; ++w) {
                for (int i = threadIdx.x; i < wl[w]; i += kBlock) s_val[wo + i] = FXb[(int64_t)ws[w] + i];
                wo += wl[w];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < E; ++k) {
                int slot = 0, wo2 = 0;
#pragma unroll
                for (int w = 0; w < kFxWin; ++w) {
                    const unsigned d = (unsigned)(r[k] - ws[w]);
                    slot = d < (unsigned)wl[w] ? wo2 + (int)d : slot;
                    wo2 += wl[w];
                }
                b[k] = s_val[slot];
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int cb = c[k] - c_lo;
        valid[k] = (c[k] != none) & (c[k] != ColorTraits<CT>::pad) & ((unsigned)cb < (unsigned)nB);
        const int cs = valid[k] ? cb : 0;
        const int64_t at = (int64_t)cs * ld + r[k];
        e[k] = LDS_EPS ? s_eps[cs] : eps[c_lo + cs];
        if (MODE == 0) { a[k] = FXa[at]; if (!(FXL && fxl)) b[k] = FXb[r[k]]; }
        else if (MODE == 1) { a[k] = FXa[at]; b[k] = FXb[at]; }
        else { a[k] = FXa[at * 2 + 1]; b[k] = 0.0; }
    }
#pragma unroll
    for (int k = 0; k < E; ++k) {
        real_t q;
        if (MODE == 0) q = (a[k] - b[k]) / e[k];
        else if (MODE == 1) q = (a[k] - b[k]) / (2 * e[k]);
        else q = a[k] / e[k];
        s_val[pos[k]] = valid[k] ? q : 0.0;
        if (!ALLW) s_flag[pos[k]] = (uint8_t)(valid[k] | ((c[k] == none) & (c_lo == 0)));
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < E / 2; ++u) {
        const int lp = u * kBlock * 2 + threadIdx.x * 2;
        const int64_t p = t0 + lp;
        const d2_t pk = *reinterpret_cast<const d2_t *>(s_val + lp);
        bool w0 = p < n, w1 = p + 1 < n;
        if (!ALLW) { w0 = w0 & (s_flag[lp] != 0); w1 = w1 & (s_flag[lp + 1] != 0); }
        const bool both = w0 & w1 & ((vec_ok & 1) != 0);
        if (__builtin_amdgcn_ballot_w64(both) == __builtin_amdgcn_ballot_w64(true)) {
            *reinterpret_cast<d2_t *>(out + p) = pk;
        } else {
            if (w0) out[p] = pk.x;
            if (w1) out[p + 1] = pk.y;
        }
    }
}

// K3c  the same decompression for LOCALLY BANDED patterns (tridiagonal / banded CSC, block patterns
//   with few colours per tile): the rows of a tile of kSortTile storage-ordered entries fall into a
//   short window [rmin, rmin + 2*npairs) and use ncol <= NCT consecutive colours (both found at plan
//   time).  The workgroup loads that window of fx and of the ncol batched f! arrays DENSELY (16 B per
//   lane, every 128-B line exactly once, no address divergence), forms the difference quotient
//   (src/jacobians.jl:565 / 607 / 635) once per (colour,row) into LDS, and the stored entries then
//   gather from LDS and leave with dense 16-B stores.  For a band of width C coloured with C colours
//   every (colour,row) quotient is used by exactly one stored entry, so nothing is loaded twice or in
//   vain.  Scattered patterns (a 5-point stencil touches rows k-nx, k, k+nx) get up to four windows per
//   tile; their re-reads are served by the L2 and replace the divergent gathers that make the gather
//   kernels texture-addresser bound.  Because row and colour are tile-relative, the per-entry index
//   shrinks from rowval (4 B) + colour (1 B) to ONE 16-bit code:  bits 0-10 LDS slot of the row (windows
//   concatenated), bits 11-13 colour - cmin, bit 14 "column
//   has no colour" (entry is written as 0), bit 15 "padding" (not a stored entry).
//   Same operations on the same operands as k_decompress_list => bit-identical results.
template <int MODE, int NCT, bool FXB_VEC, int U>
__global__ void __launch_bounds__(kBlock)
k_decompress_window(const uint32_t *__restrict__ wcode2, const int4 *__restrict__ wtiles,
                    const real_t *__restrict__ FXa, const real_t *__restrict__ FXb, int64_t ld, int64_t M,
                    const real_t *__restrict__ eps, int c_lo, int c_hi, real_t *__restrict__ out, int64_t n,
                    int vec_ok, int wp, int per_P, int per_S, int per_magic, int64_t tile0, int64_t ntl,
                    int64_t bd_t0, int64_t bd_t1, int64_t bd_off, int bd_w, int bd_u, int bd_C, uint64_t bd_mw)
{
    // (tile0, ntl): this launch covers the tiles [tile0, tile0 + ntl) -- all of them, or one row strip's)
    constexpr int T = U * kBlock * 2;             // entries per tile; U pairs of entries per thread
    extern __shared__ real_t s_mem_w[];
    uint16_t *s_head = reinterpret_cast<uint16_t *>(s_mem_w);   // first kWinPeriodMax codes of a regular tile (see phase 1)
    real_t *s_eps = reinterpret_cast<real_t *>(reinterpret_cast<char *>(s_mem_w) + kWinHeadBytes);   // step sizes of the tile's colours
    real_t *s_win = s_eps + kWinMaxCol;           // [ncol][wp] differences f(x+eps_c) - f(x) over the tile's row windows
    const int64_t xt = xcd_tile(blockIdx.x, ntl);
    if (xt >= ntl) return;
    const int64_t tile_id = tile0 + ((vec_ok & 4) ? ntl - 1 - xt : xt);   // reversed tile order, see tile_order_reversed()
    const int64_t t0 = tile_id * T;
    int4 th, wa, wb;
    if (tile_id >= bd_t0 && tile_id < bd_t1) {
        // Uniform band (plan-time check: these tiles' stored descriptors ARE what follows, band_tile_desc in fdjac_api.hip):
        // entries Q0 .. Q1 = w*j + k touch the rows j - u + k -- no descriptor load in front of the window loads
        const uint32_t Q0 = (uint32_t)(t0 + bd_off), Q1 = Q0 + (uint32_t)(T - 1);      // (full tiles only)
        const uint32_t j0 = fd_div31(Q0, bd_mw), k0 = Q0 - j0 * (uint32_t)bd_w;
        const uint32_t j1 = fd_div31(Q1, bd_mw), k1 = Q1 - j1 * (uint32_t)bd_w;
        const int wm2 = bd_w - 2;                                  // (signed: -1 for a diagonal band)
        const int rmin = (int)j0 - bd_u + ((int)k0 < 1 ? (int)k0 : 1);
        const int rmax = (int)j1 - bd_u + ((int)k1 > wm2 ? (int)k1 : wm2);
        const int r0 = rmin & ~1, np = (rmax - r0) / 2 + 1;
        th = int4{0, bd_C, np, 1 | 0x100};
        wa = int4{r0, np, 0, np};
        wb = int4{0, np, 0, np};
    } else {
        th = wtiles[3 * tile_id]; wa = wtiles[3 * tile_id + 1]; wb = wtiles[3 * tile_id + 2];
    }
    const int cmin = __builtin_amdgcn_readfirstlane(th.x);
    int cb0 = cmin;
    int cb1 = cb0 + __builtin_amdgcn_readfirstlane(th.y);
    const int npairs = __builtin_amdgcn_readfirstlane(th.z);          // row pairs of all windows together
    // up to 4 row windows, concatenated in LDS: window k holds pairs [end_{k-1}, end_k) and starts at row rw_k
    const int rw0 = __builtin_amdgcn_readfirstlane(wa.x), e0 = __builtin_amdgcn_readfirstlane(wa.y);
    const int rw1 = __builtin_amdgcn_readfirstlane(wa.z), e1 = __builtin_amdgcn_readfirstlane(wa.w);
    const int rw2 = __builtin_amdgcn_readfirstlane(wb.x), e2 = __builtin_amdgcn_readfirstlane(wb.y);
    const int rw3 = __builtin_amdgcn_readfirstlane(wb.z);
    cb0 = cb0 > c_lo ? cb0 : c_lo;                // colours of this tile that belong to the current chunk
    cb1 = cb1 < c_hi ? cb1 : c_hi;
    const int ncol = cb1 > cb0 ? cb1 - cb0 : 0;   // 0: nothing to load, entries only get their zeros

    // phase 1: the packed (row, colour) codes of the stored entries (in flight while the window is loaded).
    // Regular tiles (flag in the descriptor, plan-wide period per_P and slot step per_S: code[q + P] == code[q] + S)
    // stage only their first kWinPeriodMax codes; every entry computes its own in phase 3.
    const bool periodic = per_P > 0 && ((__builtin_amdgcn_readfirstlane(th.w) >> 8) & 1);
    uint32_t code[U];
    if (periodic) {
#pragma unroll
        for (int u = 0; u < U; ++u) code[u] = 0;
        if (threadIdx.x < kWinPeriodMax / 2)
            reinterpret_cast<uint32_t *>(s_head)[threadIdx.x] = wcode2[(t0 >> 1) + threadIdx.x];
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) code[u] = wcode2[(t0 >> 1) + u * kBlock + threadIdx.x];   // two 16-bit codes
    }
    if (threadIdx.x < NCT) s_eps[threadIdx.x] = ((int)threadIdx.x < ncol) ? eps[cb0 + threadIdx.x] : 1.0;

    {
    // phase 2: dense window loads -> differences -> LDS.  (The quotient is formed in phase 3, once per stored
        // entry: windows of scattered patterns hold values no entry of this tile uses, and an IEEE division is ~50
        // cycles per wave.)
    #pragma unroll 1
        for (int i = threadIdx.x; i < npairs; i += kBlock) {
            const int64_t row = i < e0 ? (int64_t)rw0 + 2 * i
                              : i < e1 ? (int64_t)rw1 + 2 * (i - e0)
                              : i < e2 ? (int64_t)rw2 + 2 * (i - e1) : (int64_t)rw3 + 2 * (i - e2);
            d2_t b = {0.0, 0.0};
            if (MODE == 0 && FXb != nullptr) {   // nullptr: the subtrahend is identically zero (imag-only complex step)
                if (FXB_VEC) {
                    b = *reinterpret_cast<const d2_t *>(FXb + row);
                } else {   // caller's f_in: no padding / alignment guarantees
                    if (row < M) b.x = FXb[row];
                    if (row + 1 < M) b.y = FXb[row + 1];
                }
            }
            d2_t a[NCT], bm[NCT];
    #pragma unroll
            for (int cc = 0; cc < NCT; ++cc) {
                const int64_t at = (int64_t)(cb0 - c_lo + cc) * ld + row;
                a[cc] = d2_t{0.0, 0.0};
                bm[cc] = b;
                if (cc < ncol) {
                    if (MODE == 2) {
                        const d2_t p0 = *reinterpret_cast<const d2_t *>(FXa + at * 2);
                        const d2_t p1 = *reinterpret_cast<const d2_t *>(FXa + at * 2 + 2);
                        a[cc] = d2_t{p0.y, p1.y};
                    } else {
                        a[cc] = *reinterpret_cast<const d2_t *>(FXa + at);
                        if (MODE == 1) bm[cc] = *reinterpret_cast<const d2_t *>(FXb + at);
                    }
                }
            }
    #pragma unroll
            for (int cc = 0; cc < NCT; ++cc) {
                if (cc < ncol) {
                    const d2_t df = (MODE == 2) ? a[cc] : d2_t{a[cc].x - bm[cc].x, a[cc].y - bm[cc].y};
                    *reinterpret_cast<d2_t *>(s_win + (size_t)cc * wp + 2 * i) = df;
                }
            }
        }
    }
    __syncthreads();

    // phase 3 + 4: entries pick their difference out of LDS, divide by their colour's step (IEEE division, as
    // src/jacobians.jl:565 / 607 / 635); dense stores as in k_decompress_list
    const int cshift = cmin - cb0;   // tile-relative colour -> chunk-window-relative colour
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t p = t0 + (int64_t)u * kBlock * 2 + threadIdx.x * 2;
        real_t q[2];
        bool w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned cd = (code[u] >> (16 * h)) & 0xFFFFu;
            if (periodic) {
                const int q = u * kBlock * 2 + (int)threadIdx.x * 2 + h;     // entry index inside the tile
                const int k = (q * per_magic) >> 20;                          // q / per_P (checked at plan time)
                cd = (unsigned)s_head[q - k * per_P] + (unsigned)(k * per_S);
            }
            const int cs = (int)((cd >> 11) & 7u) + cshift;
            const bool colored = (cd & 0xC000u) == 0;
            const bool valid = colored & ((unsigned)cs < (unsigned)ncol);
            const int at = valid ? cs * wp + (int)(cd & 0x7FFu) : 0;
            real_t df = s_win[at];
            const real_t e = s_eps[valid ? cs : 0];
            const real_t v = (MODE == 1) ? df / (2 * e) : df / e;
            q[h] = valid ? v : 0.0;
            w[h] = valid | (((cd & 0x4000u) != 0) & (c_lo == 0));
        }
        const bool both = w[0] & w[1] & ((vec_ok & 1) != 0);
        if (__builtin_amdgcn_ballot_w64(both) == __builtin_amdgcn_ballot_w64(true)) {
            d2_t pk = {q[0], q[1]};
            *reinterpret_cast<d2_t *>(out + p) = pk;
        } else {
            if (w[0]) out[p] = q[0];
            if (w[1]) out[p + 1] = q[1];
        }
    }
}

// K3d  row windows over 2-D (strided) tiles.  For a 2-D stencil in natural ordering a storage-ordered tile needs three
//   far-apart row windows and loads every f! value three times (two of them from L2); measured on the 5-point
//   stencil those re-reads cost a third of the kernel (322 us vs 213 us with the far windows switched off).  Here a
//   tile is up to kW2MaxRun COLUMN RUNS one stencil stride apart (R consecutive grid rows x L positions), so the
//   windows of grid rows g-1 .. g+R are loaded once for R runs: (R+2)(L+2)/(RL) = 1.4 loads per value instead of 3.
//   The stored entries of a tile are the concatenation of its runs (each contiguous in nzval); codes are stored in
//   tile order.  Descriptor (kW2Desc ints per tile): [0] first colour [1] colours [2] window pairs [3] windows
//   [4] runs [5] entries (runs padded to even) [6,7] first code (int64); [8+2k, 9+2k] window k: first row, end pair;
//   [32+3r..34+3r] run r: first output position (int64), end entry.  Same arithmetic as k_decompress_window.
template <int MODE, int NCT, bool FXB_VEC>
__global__ void __launch_bounds__(kBlock)
k_decompress_window2d(const uint16_t *__restrict__ wcode, const int *__restrict__ desc,
                      const real_t *__restrict__ FXa, const real_t *__restrict__ FXb, int64_t ld, int64_t M,
                      const real_t *__restrict__ eps, int c_lo, int c_hi, real_t *__restrict__ out, int64_t ntiles,
                      int vec_ok, int wp)
{
    extern __shared__ real_t s_mem_2d[];
    real_t *s_eps = s_mem_2d;                                  // kWinMaxCol step sizes
    int *s_desc = reinterpret_cast<int *>(s_mem_2d + kWinMaxCol);   // kW2Desc ints
    real_t *s_win = s_mem_2d + kWinMaxCol + kW2Desc * 4 / sizeof(real_t);   // [ncol][wp] differences
    const int64_t xt = xcd_tile(blockIdx.x, ntiles);
    if (xt >= ntiles) return;
    const int64_t tile_id = (vec_ok & 4) ? ntiles - 1 - xt : xt;
    if (threadIdx.x < kW2Desc) s_desc[threadIdx.x] = desc[tile_id * kW2Desc + threadIdx.x];
    __syncthreads();
    const int cmin = s_desc[0];
    int cb0 = cmin, cb1 = cmin + s_desc[1];
    const int npairs = s_desc[2], nwin = s_desc[3], nruns = s_desc[4], nent = s_desc[5];
    const int64_t code0 = ((int64_t)(uint32_t)s_desc[6]) | ((int64_t)s_desc[7] << 32);
    cb0 = cb0 > c_lo ? cb0 : c_lo;
    cb1 = cb1 < c_hi ? cb1 : c_hi;
    const int ncol = cb1 > cb0 ? cb1 - cb0 : 0;
    if (threadIdx.x < NCT) s_eps[threadIdx.x] = ((int)threadIdx.x < ncol) ? eps[cb0 + threadIdx.x] : 1.0;
    // the packed entry codes of the whole tile, requested NOW: they travel while the windows are loaded, and the entry
    // phase below is one straight pass instead of kW2Iter dependent (load code -> gather -> store) rounds.  A tile holds at
    // most 2048 + 2*kW2MaxRun entries (try_window2d_plan) = kW2Iter rounds of 512.
    constexpr int kW2Iter = (2048 + 2 * kW2MaxRun + 2 * kBlock - 1) / (2 * kBlock);
    uint32_t tcode[kW2Iter];
#pragma unroll
    for (int it = 0; it < kW2Iter; ++it) {
        const int e = 2 * (int)threadIdx.x + it * 2 * kBlock;
        tcode[it] = e < nent ? *reinterpret_cast<const uint32_t *>(wcode + code0 + e) : 0x80008000u;
    }

    {
#pragma unroll 1
        for (int i = threadIdx.x; i < npairs; i += kBlock) {
            int k = 0;
            while (k + 1 < nwin && i >= s_desc[9 + 2 * k]) ++k;
            const int pbase = k ? s_desc[7 + 2 * k] : 0;
            const int64_t row = (int64_t)s_desc[8 + 2 * k] + 2 * (i - pbase);
            d2_t b = {0.0, 0.0};
            if (MODE == 0 && FXb != nullptr) {   // nullptr: the subtrahend is identically zero (imag-only complex step)
                if (FXB_VEC) {
                    b = *reinterpret_cast<const d2_t *>(FXb + row);
                } else {
                    if (row < M) b.x = FXb[row];
                    if (row + 1 < M) b.y = FXb[row + 1];
                }
            }
            d2_t a[NCT], bm[NCT];
    #pragma unroll
            for (int cc = 0; cc < NCT; ++cc) {
                const int64_t at = (int64_t)(cb0 - c_lo + cc) * ld + row;
                a[cc] = d2_t{0.0, 0.0};
                bm[cc] = b;
                if (cc < ncol) {
                    if (MODE == 2) {
                        const d2_t p0 = *reinterpret_cast<const d2_t *>(FXa + at * 2);
                        const d2_t p1 = *reinterpret_cast<const d2_t *>(FXa + at * 2 + 2);
                        a[cc] = d2_t{p0.y, p1.y};
                    } else {
                        a[cc] = *reinterpret_cast<const d2_t *>(FXa + at);
                        if (MODE == 1) bm[cc] = *reinterpret_cast<const d2_t *>(FXb + at);
                    }
                }
            }
    #pragma unroll
            for (int cc = 0; cc < NCT; ++cc)
                if (cc < ncol) {
                    const d2_t df = (MODE == 2) ? a[cc] : d2_t{a[cc].x - bm[cc].x, a[cc].y - bm[cc].y};
                    *reinterpret_cast<d2_t *>(s_win + (size_t)cc * wp + 2 * i) = df;
                }
        }
    }
    __syncthreads();

    const int cshift = cmin - cb0;
#pragma unroll
    for (int it = 0; it < kW2Iter; ++it) {
        const int e = 2 * (int)threadIdx.x + it * 2 * kBlock;
        if (e >= nent) break;
        const uint32_t code = tcode[it];
        int r = 0;
        while (r + 1 < nruns && e >= s_desc[34 + 3 * r]) ++r;
        const int ebase = r ? s_desc[31 + 3 * r] : 0;
        const int64_t o0 = ((int64_t)(uint32_t)s_desc[32 + 3 * r]) | ((int64_t)s_desc[33 + 3 * r] << 32);
        const int64_t p = o0 + (e - ebase);
        real_t q[2];
        bool w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned cd = (code >> (16 * h)) & 0xFFFFu;
            const int cs = (int)((cd >> 11) & 7u) + cshift;
            const bool colored = (cd & 0xC000u) == 0;
            const bool valid = colored & ((unsigned)cs < (unsigned)ncol);
            const int at = valid ? cs * wp + (int)(cd & 0x7FFu) : 0;
            real_t df = s_win[at];
            const real_t ee = s_eps[valid ? cs : 0];
            const real_t v = (MODE == 1) ? df / (2 * ee) : df / ee;
            q[h] = valid ? v : 0.0;
            w[h] = valid | (((cd & 0x4000u) != 0) & (c_lo == 0));
        }
        if (w[0] & w[1] & ((p & 1) == 0) & ((vec_ok & 1) != 0)) {
            d2_t pk = {q[0], q[1]};
            *reinterpret_cast<d2_t *>(out + p) = pk;
        } else {
            if (w[0]) out[p] = q[0];
            if (w[1]) out[p + 1] = q[1];
        }
    }
}

// K4a  Tridiagonal J: three dense diagonals, no index traffic at all.
//   d[j] = D_c(j)[j] ; dl[j] = D_c(j)[j+1] ; du[j-1] = D_c(j)[j-1],  D_c = (fx1_c - fx)/eps_c
//   (what src/iteration_utils.jl:25-32 stores through Tridiagonal's setindex!).
//   j runs over the local column window [j0,j1); outputs are indexed relative to the window:
//   d[j-j0], dl[j-j0], du[j-1-du0] with du0 = max(j0-1,0).
template <typename CT, int MODE>
__global__ void __launch_bounds__(kBlock)
k_decompress_tridiag(const CT *__restrict__ color, const real_t *__restrict__ FXa,
                     const real_t *__restrict__ FXb, int64_t ld, const real_t *__restrict__ eps,
                     int c_lo, int c_hi, int64_t N, int64_t j0, int64_t j1,
                     real_t *__restrict__ dl, real_t *__restrict__ d, real_t *__restrict__ du)
{
    const int none = ColorTraits<CT>::none;
    const int64_t du0 = j0 > 0 ? j0 - 1 : 0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = j0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < j1; j += stride) {
        const int c = color[j];
        if (c == none) {
            if (c_lo == 0) {
                d[j - j0] = 0.0;
                if (j + 1 < N) dl[j - j0] = 0.0;
                if (j > 0) du[j - 1 - du0] = 0.0;
            }
            continue;
        }
        if (c < c_lo || c >= c_hi) continue;
        const real_t e = eps[c];
        const int cb = c - c_lo;
        d[j - j0] = entry_value<MODE>(FXa, FXb, ld, cb, j, e);
        if (j + 1 < N) dl[j - j0] = entry_value<MODE>(FXa, FXb, ld, cb, j + 1, e);
        if (j > 0) du[j - 1 - du0] = entry_value<MODE>(FXa, FXb, ld, cb, j - 1, e);
    }
}

// K4a'  Tridiagonal J through row windows: a workgroup owns kTriTile consecutive columns [a, a+kTriTile), loads
//   rows [a-1, a+kTriTile+1) of fx and of the (<= NCT) batched f! arrays densely into LDS as differences, and
//   writes its slices of the three diagonals with dense 16-B stores:
//       d[i] = D_c(i)[i],   dl[i] = D_c(i)[i+1],   du[i] = D_c(i+1)[i]      (D_c = (fx1_c - fx)/eps_c)
//   Used when maximum(colorvec) <= NCT (a tridiagonal pattern needs 3 colours): every loaded value is then used.
//   Same arithmetic as k_decompress_tridiag => bit-identical.
constexpr int kTriTile = 1024;
template <typename CT, int MODE, int NCT>
__global__ void __launch_bounds__(kBlock)
k_decompress_tridiag_window(const CT *__restrict__ color, const real_t *__restrict__ FXa,
                            const real_t *__restrict__ FXb, int64_t ld, const real_t *__restrict__ eps, int c_lo,
                            int c_hi, int64_t N, int64_t j0, int64_t j1, real_t *__restrict__ dl,
                            real_t *__restrict__ d, real_t *__restrict__ du, int fxb_vec, int vec_ok)
{
    constexpr int WP = kTriTile + 4;                 // LDS pitch (rows a-2 .. a+kTriTile+1, even start)
    extern __shared__ real_t s_mem_t[];              // kWinMaxCol step sizes, then ncol x WP differences
    real_t *s_eps = s_mem_t;
    real_t *s_win = s_mem_t + kWinMaxCol;
    const int none = ColorTraits<CT>::none;
    const int64_t ntiles = (j1 - j0 + kTriTile - 1) / kTriTile;
    const int64_t xt = xcd_tile(blockIdx.x, ntiles);
    if (xt >= ntiles) return;
    const int64_t tile_id = (vec_ok & 8) ? ntiles - 1 - xt : xt;   // reversed tile order, see tile_order_reversed()
    const int64_t a = j0 + tile_id * kTriTile;       // j0 is even by construction (host), so a is even
    const int64_t b = (a + kTriTile < j1) ? a + kTriTile : j1;
    const int ncol = c_hi - c_lo;                    // <= NCT
    const int64_t rbeg = a >= 2 ? a - 2 : 0;         // even; rows needed: a-1 .. b
    const int64_t rend = (b + 1 < N) ? b + 1 : N;    // exclusive
    const int npairs = (int)((rend - rbeg + 1) / 2);
    if (threadIdx.x < NCT) s_eps[threadIdx.x] = ((int)threadIdx.x < ncol) ? eps[c_lo + threadIdx.x] : 1.0;
#pragma unroll 1
    for (int i = threadIdx.x; i < npairs; i += kBlock) {
        const int64_t row = rbeg + 2 * i;
        d2_t bb = {0.0, 0.0};
        if (MODE == 0 && FXb != nullptr) {   // nullptr: the subtrahend is identically zero (imag-only complex step)
            if (fxb_vec) {
                bb = *reinterpret_cast<const d2_t *>(FXb + row);
            } else {
                if (row < N) bb.x = FXb[row];
                if (row + 1 < N) bb.y = FXb[row + 1];
            }
        }
        d2_t av[NCT], bv[NCT];
#pragma unroll
        for (int cc = 0; cc < NCT; ++cc) {
            const int64_t at = (int64_t)cc * ld + row;
            av[cc] = d2_t{0.0, 0.0};
            bv[cc] = bb;
            if (cc < ncol) {
                if (MODE == 2) {
                    const d2_t p0 = *reinterpret_cast<const d2_t *>(FXa + at * 2);
                    const d2_t p1 = *reinterpret_cast<const d2_t *>(FXa + at * 2 + 2);
                    av[cc] = d2_t{p0.y, p1.y};
                } else {
                    av[cc] = *reinterpret_cast<const d2_t *>(FXa + at);
                    if (MODE == 1) bv[cc] = *reinterpret_cast<const d2_t *>(FXb + at);
                }
            }
        }
#pragma unroll
        for (int cc = 0; cc < NCT; ++cc)
            if (cc < ncol) {
                const d2_t df = (MODE == 2) ? av[cc] : d2_t{av[cc].x - bv[cc].x, av[cc].y - bv[cc].y};
                *reinterpret_cast<d2_t *>(s_win + cc * WP + 2 * i) = df;
            }
    }
    __syncthreads();

    const int64_t du0 = j0 > 0 ? j0 - 1 : 0;
    auto quot = [&](int c, int64_t r) -> real_t {   // D_c[r] for a colour of this chunk
        const real_t df = s_win[(c - c_lo) * WP + (int)(r - rbeg)];
        const real_t e = s_eps[c - c_lo];
        return (MODE == 1) ? df / (2 * e) : df / e;
    };
    if (a == j0 && j0 > 0 && threadIdx.x == 0) {   // du of the plan's first column (its pair partner is another rank's)
        const int c = (int)color[j0];
        if ((unsigned)(c - c_lo) < (unsigned)ncol) du[0] = quot(c, j0 - 1);
        else if ((c == none) & (c_lo == 0)) du[0] = 0.0;
    }
#pragma unroll 1
    for (int64_t i = a + 2 * (int64_t)threadIdx.x; i < b; i += 2 * kBlock) {
        // colours of columns i, i+1, i+2 (i is even)
        int c0, c1, c2 = none, c3;
        if (i + 1 < N) load_color_pair<CT>(color + i, c0, c1); else { c0 = (int)color[i]; c1 = none; }
        if (i + 3 < N) load_color_pair<CT>(color + i + 2, c2, c3); else if (i + 2 < N) c2 = (int)color[i + 2];
        const bool in0 = (unsigned)(c0 - c_lo) < (unsigned)ncol, in1 = (i + 1 < b) & ((unsigned)(c1 - c_lo) < (unsigned)ncol);
        const bool in1x = (i + 1 < N) & ((unsigned)(c1 - c_lo) < (unsigned)ncol);      // column i+1 may belong to the next tile
        const bool in2 = (i + 2 < N) & (i + 2 < j1) & ((unsigned)(c2 - c_lo) < (unsigned)ncol);
        const bool z0 = (c0 == none) & (c_lo == 0), z1 = (i + 1 < b) & (c1 == none) & (c_lo == 0);
        const bool z1x = (i + 1 < N) & (i + 1 < j1) & (c1 == none) & (c_lo == 0), z2 = (i + 2 < N) & (i + 2 < j1) & (c2 == none) & (c_lo == 0);
        // d[i], d[i+1]
        {
            const real_t q0 = in0 ? quot(c0, i) : 0.0, q1 = in1 ? quot(c1, i + 1) : 0.0;
            const bool w0 = in0 | z0, w1 = in1 | z1;
            real_t *o = d + (i - j0);
            if (__builtin_amdgcn_ballot_w64(w0 & w1 & (vec_ok & 1)) == __builtin_amdgcn_ballot_w64(true)) {
                *reinterpret_cast<d2_t *>(o) = d2_t{q0, q1};
            } else { if (w0) o[0] = q0; if (w1) o[1] = q1; }
        }
        // dl[i] (column i, row i+1), dl[i+1] (column i+1, row i+2)
        {
            const bool e0 = i + 1 < N, e1 = (i + 1 < b) & (i + 2 < N);
            const real_t q0 = (in0 & e0) ? quot(c0, i + 1) : 0.0, q1 = (in1 & e1) ? quot(c1, i + 2) : 0.0;
            const bool w0 = (in0 | z0) & e0, w1 = (in1 | z1) & e1;
            real_t *o = dl + (i - j0);
            if (__builtin_amdgcn_ballot_w64(w0 & w1 & ((vec_ok >> 1) & 1)) == __builtin_amdgcn_ballot_w64(true)) {
                *reinterpret_cast<d2_t *>(o) = d2_t{q0, q1};
            } else { if (w0) o[0] = q0; if (w1) o[1] = q1; }
        }
        // du[i] (column i+1, row i), du[i+1] (column i+2, row i+1); columns must lie inside [j0, j1)
        {
            const bool e0 = (i + 1 < N) & (i + 1 < j1), e1 = (i + 1 < b) & (i + 2 < N) & (i + 2 < j1);
            const real_t q0 = (in1x & e0) ? quot(c1, i) : 0.0, q1 = (in2 & e1) ? quot(c2, i + 1) : 0.0;
            const bool w0 = (in1x | z1x) & e0, w1 = (in2 | z2) & e1;
            real_t *o = du + (i - du0);
            if (__builtin_amdgcn_ballot_w64(w0 & w1 & ((vec_ok >> 2) & 1)) == __builtin_amdgcn_ballot_w64(true)) {
                *reinterpret_cast<d2_t *>(o) = d2_t{q0, q1};
            } else { if (w0) o[0] = q0; if (w1) o[1] = q1; }
        }
    }
}

// K4b  BandedMatrix J: data is (l+u+1) x N column-major; slot k of column j holds row j-u+k
//   (ext/FiniteDiffBandedMatricesExt.jl:13-27, storage per its line 22).  One thread per data
//   slot => dense coalesced stores, implicit indices, one colour byte per column (L1-resident).
template <typename CT, int MODE>
__global__ void __launch_bounds__(kBlock)
k_decompress_banded(const CT *__restrict__ color, const real_t *__restrict__ FXa,
                    const real_t *__restrict__ FXb, int64_t ld, const real_t *__restrict__ eps,
                    int c_lo, int c_hi, int64_t M, int64_t l, int64_t u, int64_t j0, int64_t j1,
                    real_t *__restrict__ data)
{
    const int none = ColorTraits<CT>::none;
    const int64_t w = l + u + 1;
    const int64_t total = (j1 - j0) * w;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t jj = e / w;
        const int64_t k = e - jj * w;
        const int64_t j = j0 + jj;
        const int64_t r = j - u + k;
        const int c = color[j];
        if (r < 0 || r >= M || c == none) {
            if (c_lo == 0) data[e] = 0.0;
            continue;
        }
        if (c < c_lo || c >= c_hi) continue;
        data[e] = entry_value<MODE>(FXa, FXb, ld, c - c_lo, r, eps[c]);
    }
}


// K4c  BandedBlockBandedMatrix J (ext/FiniteDiffBlockBandedMatricesExt.jl:16-42): every in-band block (K, J) owns a slab of banded
//   data, (lam + mu + 1) x n_J values with column stride st_J; the reference stores entry (k, j) of the block at
//   start(K, J) + j st + mu + k - j (its lines 29-36).  One thread per slot (column j, block-band d = K - J + bu, sub-band t = mu + k - j):
//   implicit indices -- an int32 block number per column and the per-block tables are all the index data there is (the entry-list
//   plan moved 13 B of index per 8-B value) --, lane-consecutive slots of a column, dense stores in BlockBandedMatrices' own layout.
//   Slots of rows outside their block and columns without colour are written as 0 (the reference's prologue zero-fills J,
//   src/jacobians.jl:530-532).
template <typename CT, int MODE>
__global__ void __launch_bounds__(kBlock)
k_decompress_bbb(const CT *__restrict__ color, const real_t *__restrict__ FXa, const real_t *__restrict__ FXb, int64_t ld,
                 const real_t *__restrict__ eps, int c_lo, int c_hi, int64_t N, int64_t nb, int bl, int bu, int lam, int mu,
                 const int32_t *__restrict__ off, const int32_t *__restrict__ blk, const int64_t *__restrict__ start,
                 const int64_t *__restrict__ stride, real_t *__restrict__ data)
{
    const int none = ColorTraits<CT>::none;
    const int w = bl + bu + 1, sw = lam + mu + 1, R = w * sw;
    const int64_t total = N * R;
    const int64_t step = (int64_t)gridDim.x * kBlock;
    // Three round trips per slot instead of six: (block number, colour) -> (slab start, block offsets, stride, step size) -> f! values.
    // Every load is unconditional, from an index clamped to something that exists; what does not apply is selected away afterwards.
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += step) {
        const int64_t j = e / R;
        const int s = (int)(e - j * R), d = s / sw, t = s - d * sw;
        const int J = blk[j];
        const int c = color[j];
        const int64_t K = (int64_t)J + d - bu;
        const bool kin = K >= 0 && K < nb;
        const int64_t Kc = kin ? K : 0;
        const bool cin = c != none && c >= c_lo && c < c_hi;
        const int cc = cin ? c : c_lo;
        const int64_t st0 = start[(int64_t)d + (int64_t)w * J];
        const int oJ = off[J], oK = off[Kc], oK1 = off[Kc + 1];
        const int64_t sJ = stride[J];
        const real_t ec = eps[cc];
        const int jj = (int)(j - oJ), k = jj + t - mu, m = kin ? oK1 - oK : 0;
        const bool inb = k >= 0 && k < m;
        // (a slot outside its block reads SOME row of f -- its value is discarded: row oK, clamped into the vector for an empty last block)
        const int64_t rsafe = (int64_t)oK + (inb ? k : 0);
        const real_t v = entry_value<MODE>(FXa, FXb, ld, cc - c_lo, rsafe < N ? rsafe : N - 1, ec);
        if (st0 < 0) continue;                                     // no such block and no slab reserved for it
        // (a layout that reserves the slab of a block outside the matrix -- BlockBandedMatrices' own: (bl+bu+1)(lam+mu+1) rows per
        //  column -- gets its zeros here; the plan zero-fills other layouts before the launch)
        real_t *o = data + st0 + (int64_t)jj * sJ + t;
        if (!inb || c == none) {
            if (c_lo == 0) *o = 0.0;
            continue;
        }
        if (cin) *o = v;
    }
}

// K5b  the same column-range decompression, one WORKGROUP per kCrCols consecutive columns.  A wave per column leaves
//   a wave with ~1.5 loads of work behind two dependent round trips (column metadata, then values): the kernel runs
//   at the latency, not the bandwidth, of the memory system.  Here the metadata of kCrCols columns is fetched with
//   one coalesced load, and their (row, column) work items are flattened so that every thread has kCrU independent
//   gathers in flight.
constexpr int kCrCols = 32;
constexpr int kCrU = 4;
template <typename CT, int MODE, bool VEC>
__global__ void __launch_bounds__(kBlock)
k_decompress_colrange_wg(const CT *__restrict__ color, const int32_t *__restrict__ rlo,
                         const int32_t *__restrict__ cnt, const int64_t *__restrict__ off,
                         const real_t *__restrict__ FXa, const real_t *__restrict__ FXb, int64_t ld,
                         const real_t *__restrict__ eps, int c_lo, int c_hi, int64_t j0, int64_t ncols,
                         real_t *__restrict__ data, int rev)
{
    // VEC: every column's first row, row count and destination are even (found at plan time) and the arrays are
    // 16-B aligned -- a work item is a PAIR of rows (one 16-B load per operand, one 16-B store).
    // FXb == nullptr: the subtrahend is identically zero (imaginary parts of the complex step, a - 0.0 == a).
    constexpr int SH = VEC ? 1 : 0;
    __shared__ int s_c[kCrCols], s_r0[kCrCols], s_end[kCrCols + 1];
    __shared__ int64_t s_off[kCrCols];
    __shared__ real_t s_e[kCrCols];
    const int none = ColorTraits<CT>::none;
    const int64_t jb = (int64_t)(rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * kCrCols;   // see tile_order_reversed()
    const int nloc = (int)((ncols - jb < kCrCols) ? ncols - jb : kCrCols);
    if (threadIdx.x < kCrCols) {
        const int t = threadIdx.x;
        int c = none, n = 0;
        if (t < nloc) {
            c = (int)color[j0 + jb + t];
            n = cnt[jb + t] >> SH;
            s_r0[t] = rlo[jb + t];
            s_off[t] = off[jb + t];
        }
        const bool live = (c != none) & (c >= c_lo) & (c < c_hi);
        const bool zero = (c == none) & (c_lo == 0) & (t < nloc);
        s_c[t] = live ? c - c_lo : (zero ? -1 : -2);      // >= 0 colour of the chunk, -1 write zeros, -2 leave untouched
        s_e[t] = live ? eps[c] : (real_t)1;
        // inclusive prefix of the work items (rows, or row pairs) over the 32 columns (one half-wave)
        int incl = n;
#pragma unroll
        for (int d = 1; d < kCrCols; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (t >= d) incl += o;
        }
        s_end[t + 1] = incl;
        if (t == 0) s_end[0] = 0;
    }
    __syncthreads();
    const int total = s_end[kCrCols];
    for (int q0 = 0; q0 < total; q0 += kBlock * kCrU) {
        d2_t a[kCrU], b[kCrU];
        int col[kCrU], k[kCrU];
#pragma unroll
        for (int u = 0; u < kCrU; ++u) {
            const int q = q0 + u * kBlock + (int)threadIdx.x;
            a[u] = d2_t{0, 0}; b[u] = d2_t{0, 0}; col[u] = -1; k[u] = 0;
            if (q < total) {
                int lo = 0, hi = kCrCols;                 // column of work item q: s_end[lo] <= q < s_end[lo + 1]
#pragma unroll
                for (int it = 0; it < 5; ++it) {
                    const int mid = (lo + hi) >> 1;
                    if (s_end[mid] <= q) lo = mid; else hi = mid;
                }
                col[u] = lo;
                k[u] = (q - s_end[lo]) << SH;
                const int cb = s_c[lo];
                if (cb >= 0) {
                    const int64_t r = (int64_t)s_r0[lo] + k[u];
                    const int64_t at = (int64_t)cb * ld + r;
                    if constexpr (VEC) {
                        if (MODE == 2) {
                            const d2_t p0 = *reinterpret_cast<const d2_t *>(FXa + at * 2);
                            const d2_t p1 = *reinterpret_cast<const d2_t *>(FXa + at * 2 + 2);
                            a[u] = d2_t{p0.y, p1.y};
                        } else {
                            a[u] = *reinterpret_cast<const d2_t *>(FXa + at);
                            if (MODE == 1) b[u] = *reinterpret_cast<const d2_t *>(FXb + at);
                            else if (FXb) b[u] = *reinterpret_cast<const d2_t *>(FXb + r);
                        }
                    } else {
                        if (MODE == 0) { a[u].x = FXa[at]; if (FXb) b[u].x = FXb[r]; }
                        else if (MODE == 1) { a[u].x = FXa[at]; b[u].x = FXb[at]; }
                        else { a[u].x = FXa[at * 2 + 1]; }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kCrU; ++u) {
            if (col[u] < 0) continue;
            const int cb = s_c[col[u]];
            real_t *dst = data + s_off[col[u]] + k[u];
            if (cb >= 0) {
                const real_t e = s_e[col[u]];
                const real_t v0 = (MODE == 0) ? (a[u].x - b[u].x) / e : (MODE == 1) ? (a[u].x - b[u].x) / (2 * e) : a[u].x / e;
                if constexpr (VEC) {
                    const real_t v1 = (MODE == 0) ? (a[u].y - b[u].y) / e : (MODE == 1) ? (a[u].y - b[u].y) / (2 * e) : a[u].y / e;
                    *reinterpret_cast<d2_t *>(dst) = d2_t{v0, v1};
                } else {
                    dst[0] = v0;
                }
            } else if (cb == -1) {
                if constexpr (VEC) *reinterpret_cast<d2_t *>(dst) = d2_t{0, 0};
                else dst[0] = (real_t)0;
            }
        }
    }
}

// K6  dense, uncoloured arm (sparsity === nothing, src/jacobians.jl:548-557 / 590-598 / 626-631):
//   "colour" i perturbs the single entry x[i] with the PER-ELEMENT step
//   eps_i = compute_epsilon(fdtype, x[i], relstep, absstep, dir)  (src/epsilons.jl:26-29,50-53)
//   and J[:, i] = (f(x + eps_i e_i) - f(x)) / eps_i.
__global__ void __launch_bounds__(kBlock)
k_eps_element(const real_t *__restrict__ x, int64_t ncols, double relstep, double absstep, double dir,
              int is_forward, real_t *__restrict__ eps, int pair)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < ncols; i += stride) {
        // abs(x_i): of a complex element (complex-valued x) the modulus
        const real_t ax = pair ? (real_t)hypot((double)x[2 * i], (double)x[2 * i + 1]) : (real_t)fabs(x[i]);
        const real_t a = (real_t)relstep * ax;
        real_t e = (a > (real_t)absstep) ? a : (real_t)absstep;
        if (is_forward) e = e * dir;
        eps[i] = e;
    }
}

template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_decompress_dense(const real_t *__restrict__ FXa, const real_t *__restrict__ FXb, int64_t ld,
                   const real_t *__restrict__ eps, int c_lo, int c_hi, int64_t M, real_t *__restrict__ J)
{
    const int64_t total = (int64_t)(c_hi - c_lo) * M;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
        const int64_t cb = t / M, r = t - cb * M;
        J[(int64_t)(c_lo + cb) * M + r] = entry_value<MODE>(FXa, FXb, ld, (int)cb, r, eps[c_lo + cb]);
    }
}

// Stream-copy ceiling probe (16 B per lane, grid-stride).
__global__ void __launch_bounds__(kBlock)
k_stream_copy(const r2_t *__restrict__ src, r2_t *__restrict__ dst, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

__global__ void __launch_bounds__(kBlock) k_fill(real_t *__restrict__ p, int64_t n, real_t v)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// host-side launchers (called from fdjac_api.hip)
// ---------------------------------------------------------------------------------------------
// The row-window kernels walk their tiles from the LAST one to the first: the f! values were written front to back
// by the launch before, so the end of the batch is what the 256 MiB Infinity Cache still holds when the decompression
// starts (measured in one process on the same buffers, N = 10^7: tridiagonal forward 123.9 -> 118.3 / 119.8 -> 117.4 /
// 115.4 -> 114.5 us depending on buffer placement, 5-point central 307 -> 301 us, block-banded complex step
// (k_decompress_colrange_wg) 112 -> 103 us; neutral when everything fits).
// Same work per tile => same bits.
// every decompression kernel walks its tiles from the last one to the first: the f! batch was written front to back by the launch
// before, so its end is what the 256 MiB Infinity Cache still holds ("tile order")
static inline bool tile_order_reversed() { return true; }

constexpr int64_t kGridCapMult = 8;      // resident workgroups per CU of the grid-stride kernels

// Grid for a grid-stride kernel: at most cap resident workgroups, and -- because these kernels are
// bandwidth bound with identical work per tile -- a block count that divides the tiles into whole
// rounds (e.g. 4883 tiles, cap 2048 -> 3 rounds -> 1628 blocks) so no round runs partly empty.
int balanced_grid(int64_t tiles, int64_t cap)
{
    if (tiles < 1) tiles = 1;
    if (cap < 1) cap = 1;
    const int64_t rounds = (tiles + cap - 1) / cap;
    return (int)((tiles + rounds - 1) / rounds);
}

static inline int grid_for(int64_t work_items, int per_block, int num_cus)
{
    const int64_t tiles = (work_items + per_block - 1) / per_block;
    const int64_t cap = (int64_t)num_cus * kGridCapMult;
    return balanced_grid(tiles, cap);
}

// step-size reduction, register path (C <= kRegColors): the groups [g0, g0 + ng) of the two-level reduction.  final = the launch
// covers all kEpsGroups groups and writes eps itself; otherwise it stops at the group sums (d_gsum) -- a shard.
template <typename CT>
static int launch_eps_groups_t(fd_plan *p, const real_t *x, int g0, int ng, bool final, double relstep, double absstep, double dir)
{
    hipStream_t s = p->ctx->stream;
    const int C = (int)p->C;
    const int ldp = kRegColors;
    if (ng <= 0) return FD_OK;
    EpsGrid eg;
    eg.tpg = p->eps_tpg; eg.bpg = p->eps_bpg; eg.tpb = p->eps_tpb;
    eg.final_groups = final ? ng : 0;
    eg.C = C; eg.is_forward = p->fdtype == FD_FORWARD ? 1 : 0;
    eg.relstep = relstep; eg.absstep = absstep; eg.dir = dir;
#define FD_EPS_REG(NCC, CY, NTT)                                                                                \
    hipLaunchKernelGGL((k_eps_partial_reg<CT, NCC, CY, NTT>), dim3((unsigned)(ng * eg.bpg)), dim3(kBlock), 0, s, x,  \
                       (const CT *)p->d_color, p->N, p->d_partial, p->d_gsum, p->d_tick, ldp, p->cyc_C, p->cyc_shift, g0 * eg.bpg, eg, \
                       p->cx ? 1 : 0, p->d_eps, p->d_eps2)
#define FD_EPS_REG_V(NCC)                                                                                       \
    do {                                                                                                        \
        if (p->cyc_C > 0) { if (p->eps_nt) FD_EPS_REG(NCC, true, true); else FD_EPS_REG(NCC, true, false); }     \
        else { if (p->eps_nt) FD_EPS_REG(NCC, false, true); else FD_EPS_REG(NCC, false, false); }               \
    } while (0)
    if (C <= 4) FD_EPS_REG_V(4); else if (C <= 6) FD_EPS_REG_V(6); else FD_EPS_REG_V(kRegColors);      // (colour sums kept per thread: 4, 6 or 8 -- c3's five colours: 30 -> 26 us)
#undef FD_EPS_REG_V
#undef FD_EPS_REG
    if (final) p->eps2_fresh = p->d_eps2 != nullptr;
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

// the whole reduction of one GPU in the flag form (k_eps_flags); fz: the plan's fused-step buffers of this call's parity
template <typename CT>
static int launch_eps_flags_t(fd_plan *p, const real_t *x, const FusedEps &fz)
{
    hipStream_t s = p->ctx->stream;
    const unsigned grid = (unsigned)(fz.eg.C + fz.nblocks);
#define FD_EPS_FZ(NCC, CY, NTT) hipLaunchKernelGGL((k_eps_flags<CT, NCC, CY, NTT>), dim3(grid), dim3(kBlock), 0, s, x, (const CT *)p->d_color, p->N, fz)
#define FD_EPS_FZ_V(NCC)                                                                                        \
    do {                                                                                                        \
        if (p->cyc_C > 0) { if (p->eps_nt) FD_EPS_FZ(NCC, true, true); else FD_EPS_FZ(NCC, true, false); }       \
        else { if (p->eps_nt) FD_EPS_FZ(NCC, false, true); else FD_EPS_FZ(NCC, false, false); }                 \
    } while (0)
    if (fz.eg.C <= 4) FD_EPS_FZ_V(4); else if (fz.eg.C <= 6) FD_EPS_FZ_V(6); else FD_EPS_FZ_V(kRegColors);
#undef FD_EPS_FZ_V
#undef FD_EPS_FZ
    p->eps2_fresh = p->d_eps2 != nullptr;
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}
int launch_eps_flags(fd_plan *p, const real_t *x, const FusedEps &fz)
{
    return p->color8 ? launch_eps_flags_t<uint8_t>(p, x, fz) : launch_eps_flags_t<int32_t>(p, x, fz);
}

int launch_eps_groups(fd_plan *p, const real_t *x, int g0, int ng, bool final, double relstep, double absstep, double dir)
{
    return p->color8 ? launch_eps_groups_t<uint8_t>(p, x, g0, ng, final, relstep, absstep, dir)
                     : launch_eps_groups_t<int32_t>(p, x, g0, ng, final, relstep, absstep, dir);
}

// level 2 of the two-level reduction on its own (after an exchange of the group sums)
int launch_eps_final(fd_plan *p, double relstep, double absstep, double dir)
{
    hipLaunchKernelGGL(k_eps_final, dim3(1), dim3(64), 0, p->ctx->stream, p->d_gsum, kRegColors, (int)p->C, relstep, absstep, dir,
                       p->fdtype == FD_FORWARD ? 1 : 0, p->d_eps, p->d_eps2);
    p->eps2_fresh = p->d_eps2 != nullptr;
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

// (the many-colour path's second stage: one block per colour over the chunk partials)
int launch_eps_finalize(fd_plan *p, int nparts, int ldp, double relstep, double absstep, double dir)
{
    hipLaunchKernelGGL(k_eps_finalize, dim3((unsigned)p->C), dim3(kBlock), 0, p->ctx->stream, p->d_partial, nparts, ldp, relstep,
                       absstep, dir, p->fdtype == FD_FORWARD ? 1 : 0, p->d_eps, p->d_eps2);
    p->eps2_fresh = p->d_eps2 != nullptr;
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

template <typename CT>
static int launch_eps_t(fd_plan *p, const real_t *x, double relstep, double absstep, double dir)
{
    hipStream_t s = p->ctx->stream;
    const int C = (int)p->C;
    if (C <= kRegColors) return launch_eps_groups_t<CT>(p, x, 0, kEpsGroups, true, relstep, absstep, dir);
    const int nparts = p->seg_chunks;
    hipLaunchKernelGGL(k_eps_partial_seg, dim3((unsigned)((int64_t)C * nparts)), dim3(kBlock), 0, s,
                       x, p->d_perm, p->d_cptr, (int64_t)C, nparts, p->d_partial, p->cx ? 1 : 0);
    return launch_eps_finalize(p, nparts, C, relstep, absstep, dir);
}

// the fused small-problem launch (k_eps_perturb_small); pmode -1 = step sizes only
int launch_eps_perturb_small(fd_plan *p, const real_t *x, double relstep, double absstep, double dir, int pmode,
                             int base_row)
{
    hipStream_t s = p->ctx->stream;
    const int fwd = p->fdtype == FD_FORWARD ? 1 : 0;
#define FD_SMALL(CT, NC, PM)                                                                                        \
    hipLaunchKernelGGL((k_eps_perturb_small<CT, NC, PM>), dim3(1), dim3(kSmallBlock), 0, s, x, (const CT *)p->d_color, \
                       p->N, relstep, absstep, dir, fwd, (int)p->C, p->d_eps, p->d_X, p->ldx, base_row)
#define FD_SMALL_PM(CT, NC)                                                        \
    switch (pmode) {                                                               \
    case 0: FD_SMALL(CT, NC, 0); break;                                            \
    case 1: FD_SMALL(CT, NC, 1); break;                                            \
    case 2: FD_SMALL(CT, NC, 2); break;                                            \
    default: FD_SMALL(CT, NC, -1); break;                                          \
    }
    if (p->color8) { if (p->C <= 4) { FD_SMALL_PM(uint8_t, 4) } else { FD_SMALL_PM(uint8_t, kRegColors) } }
    else { if (p->C <= 4) { FD_SMALL_PM(int32_t, 4) } else { FD_SMALL_PM(int32_t, kRegColors) } }
#undef FD_SMALL_PM
#undef FD_SMALL
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

int launch_eps(fd_plan *p, const real_t *x, double relstep, double absstep, double dir)
{
    if (p->kind == K_DENSE) {
        hipLaunchKernelGGL(k_eps_element, dim3(grid_for(p->C, kBlock, p->ctx->num_cus)), dim3(kBlock), 0, p->ctx->stream,
                           x, p->C, relstep, absstep, dir, p->fdtype == FD_FORWARD ? 1 : 0, p->d_eps, p->cx ? 1 : 0);
        FD_HIP_CHECK(hipGetLastError());
        return FD_OK;
    }
    return p->color8 ? launch_eps_t<uint8_t>(p, x, relstep, absstep, dir)
                     : launch_eps_t<int32_t>(p, x, relstep, absstep, dir);
}

template <typename CT, int MODE>
static int launch_perturb_tm(fd_plan *p, const real_t *x, int c_lo, int B)
{
    const int64_t j0 = p->x0 & ~(int64_t)1;  // even start => 16-B aligned pairs
    const int64_t j1 = p->x1;
    // one pair per thread, one-shot grid: a capped grid-stride launch streams at 4.9 instead of 6.2 TB/s on this part
    // (scripts/ubench/copy_variants.hip; N = 10^7 forward: 64 -> 58 us, profiles/r04_b_*)
    const unsigned g = (unsigned)std::max<int64_t>(1, ((j1 - j0 + 1) / 2 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((k_perturb<CT, MODE>), dim3(g), dim3(kBlock), 0, p->ctx->stream, x,
                       (const CT *)p->d_color, p->d_eps, c_lo, B, j0, j1, p->d_X, p->ldx);
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

int launch_perturb(fd_plan *p, const real_t *x, int c_lo, int B)
{
#define FD_DISPATCH(CT)                                                        \
    switch (p->fdtype) {                                                       \
    case FD_FORWARD: return launch_perturb_tm<CT, 0>(p, x, c_lo, B);           \
    case FD_CENTRAL: return launch_perturb_tm<CT, 1>(p, x, c_lo, B);           \
    default: return launch_perturb_tm<CT, 2>(p, x, c_lo, B);                   \
    }
    if (p->color8) { FD_DISPATCH(uint8_t) } else { FD_DISPATCH(int32_t) }
#undef FD_DISPATCH
}

template <int MODE>
static void launch_window_m(fd_plan *p, const real_t *fx, const real_t *FXa, const real_t *FXb, int c_lo, int c_hi,
                            real_t *out)
{
    hipStream_t s = p->ctx->stream;
    // fx is the plan's own padded, 256-B aligned array unless the caller passed f_in
    const bool fxvec = (MODE != 0) || (fx == p->d_fx) || (fx == p->fx_batch_row) || (fx == p->d_zero);
    // the imaginary parts of an imag-only complex step arrive as a real array with fx = the plan's all-zero vector:
    // a - 0.0 == a, so the kernels are told not to load it at all
    const bool fx_zero = (MODE == 0) && p->d_zero != nullptr && fx == p->d_zero;
    if (fx_zero) FXb = nullptr;
    // LDS pitch between colours: 2 mod 32 elements, so that neighbouring colours start 4 banks apart (entries of neighbouring
    // columns read neighbouring rows of DIFFERENT colours; a pitch of 0 mod 32 puts them all on the same banks)
    const int wp = ((2 * p->win_pairs + 31) & ~31) + 2;
    const int narr = p->win_ncol;
    // (tiles are walked back to front: the end of the f! batch is what the Infinity Cache still holds -- DESIGN section 5)
    const int vok = ((((uintptr_t)out) & kPairMask) == 0 ? 1 : 0) | 4;
    if (p->window2d) {
        const int64_t g2 = 8 * xcd_chunks(p->w2_ntiles);
        const size_t shm2 = sizeof(real_t) * ((size_t)wp * (size_t)narr + kWinMaxCol) + 4 * (size_t)kW2Desc;
#define FD_LAUNCH_W2(NCT, FV)                                                                                      \
        hipLaunchKernelGGL((k_decompress_window2d<MODE, NCT, FV>), dim3((unsigned)g2), dim3(kBlock), shm2, s,        \
                           p->d_wcode, p->d_w2desc, FXa, FXb, p->ldf, p->M, p->d_eps, c_lo, c_hi, out, p->w2_ntiles,  \
                           vok, wp)
        if (p->win_ncol <= 4) { if (fxvec) FD_LAUNCH_W2(4, true); else FD_LAUNCH_W2(4, false); }
        else if (p->win_ncol <= 6) { if (fxvec) FD_LAUNCH_W2(6, true); else FD_LAUNCH_W2(6, false); }
        else { if (fxvec) FD_LAUNCH_W2(kWinMaxCol, true); else FD_LAUNCH_W2(kWinMaxCol, false); }
#undef FD_LAUNCH_W2
        return;
    }
    const int64_t ntl = (p->nnz_local + p->win_tile - 1) / p->win_tile, tile0 = 0;
    if (ntl <= 0) return;
    const int64_t gw = 8 * xcd_chunks(ntl);
    const size_t shmw = sizeof(real_t) * ((size_t)wp * (size_t)narr + kWinMaxCol) + kWinHeadBytes;
#define FD_LAUNCH_WIN(NCT, FV, UU)                                                                              \
    hipLaunchKernelGGL((k_decompress_window<MODE, NCT, FV, UU>), dim3((unsigned)gw), dim3(kBlock), shmw, s,     \
                       (const uint32_t *)p->d_wcode, p->d_wtiles, FXa, FXb, p->ldf, p->M, p->d_eps, c_lo,       \
                       c_hi, out, p->nnz_local, vok, wp, p->win_per_P, p->win_per_S, p->win_per_magic, tile0, ntl,  \
                       p->bd_t0, p->bd_t1, p->band_off, p->band_w, p->band_u, (int)p->band_C, p->band_mw)
#define FD_LAUNCH_WIN_U(NCT, FV) do { if (p->win_tile == 2048) FD_LAUNCH_WIN(NCT, FV, 4); else if (p->win_tile == 1024) FD_LAUNCH_WIN(NCT, FV, 2); else FD_LAUNCH_WIN(NCT, FV, 1); } while (0)
    if (p->win_ncol <= 4) { if (fxvec) FD_LAUNCH_WIN_U(4, true); else FD_LAUNCH_WIN_U(4, false); }
    else if (p->win_ncol <= 6) { if (fxvec) FD_LAUNCH_WIN_U(6, true); else FD_LAUNCH_WIN_U(6, false); }
    else { if (fxvec) FD_LAUNCH_WIN_U(kWinMaxCol, true); else FD_LAUNCH_WIN_U(kWinMaxCol, false); }
#undef FD_LAUNCH_WIN_U
#undef FD_LAUNCH_WIN
}

template <typename CT, int MODE>
static int launch_decompress_tm(fd_plan *p, const real_t *fx, int c_lo, int c_hi, real_t *const *outs)
{
    hipStream_t s = p->ctx->stream;
    const int B = c_hi - c_lo;
    const int64_t ld_eff = p->ldf;
    const real_t *FXa = p->d_FX;
    const real_t *FXb = (MODE == 0) ? fx : FXa + (int64_t)B * ld_eff;  // central: minus points
    const CT *color = (const CT *)p->d_color;
    switch (p->kind) {
    case K_CSC:
    case K_CSC_DENSE:
    case K_COO_DENSE: {
        if (p->nnz_local == 0) break;
        if (p->sorted_gather && p->kind == K_CSC) {
            const bool ldsq = B <= kEpsLdsMax;
            const bool allw = (p->nchunks == 1) && !p->has_none && p->own_c0 == 0 && (p->own_c1 < 0 || p->own_c1 >= p->C);
            const int64_t gq = 8 * xcd_chunks((p->nnz_local + kSortTile - 1) / kSortTile);
            const size_t shmq = sizeof(real_t) * (size_t)(kSortTile + (ldsq ? B : 0)) + (allw ? 0 : (size_t)kSortTile);
            const int vok = ((((uintptr_t)outs[0]) & kPairMask) == 0 ? 1 : 0) | (tile_order_reversed() ? 4 : 0);
#define FD_LAUNCH_SORTED(LL, AW, SS)                                                                                 \
            do { if (MODE == 0 && p->d_fxwin)                                                                          \
                hipLaunchKernelGGL((k_decompress_sorted<CT, MODE, LL, AW, SS, true>), dim3((unsigned)gq), dim3(kBlock), shmq, s, \
                               p->d_rowval, (const CT *)p->d_nzcolor, p->d_spos, FXa, FXb, p->ldf, p->d_eps, c_lo,    \
                               c_hi, outs[0], p->nnz_local, vok, p->d_tile_order, p->d_fxwin);                         \
            else                                                                                                       \
                hipLaunchKernelGGL((k_decompress_sorted<CT, MODE, LL, AW, SS, false>), dim3((unsigned)gq), dim3(kBlock), shmq, s, \
                               p->d_rowval, (const CT *)p->d_nzcolor, p->d_spos, FXa, FXb, p->ldf, p->d_eps, c_lo,    \
                               c_hi, outs[0], p->nnz_local, vok, p->d_tile_order, (const int32_t *)nullptr); } while (0)
            if (p->sorted_gather) {
                if (ldsq && allw) FD_LAUNCH_SORTED(true, true, true);
                else if (ldsq) FD_LAUNCH_SORTED(true, false, true);
                else if (allw) FD_LAUNCH_SORTED(false, true, true);
                else FD_LAUNCH_SORTED(false, false, true);
            } else {
                if (ldsq && allw) FD_LAUNCH_SORTED(true, true, false);
                else if (ldsq) FD_LAUNCH_SORTED(true, false, false);
                else if (allw) FD_LAUNCH_SORTED(false, true, false);
                else FD_LAUNCH_SORTED(false, false, false);
            }
#undef FD_LAUNCH_SORTED
            break;
        }
        if (p->window && p->kind == K_CSC) {
            launch_window_m<MODE>(p, fx, FXa, FXb, c_lo, c_hi, outs[0]);
            break;
        }
        constexpr int U = 2;       // pairs per thread
        const int64_t tile = (int64_t)U * kBlock * 2;
        const int64_t g = 8 * xcd_chunks((p->nnz_local + tile - 1) / tile);
        const bool lds = B <= kEpsLdsMax;
        const size_t shm = lds ? sizeof(real_t) * (size_t)B : 0;
        const int vec_ok = ((((uintptr_t)outs[0]) & kPairMask) == 0 ? 1 : 0) | (tile_order_reversed() ? 4 : 0);
#define FD_LAUNCH_LIST(HD, UU, LL, DEST, VOK)                                                                    \
        hipLaunchKernelGGL((k_decompress_list<CT, MODE, HD, UU, LL>), dim3((unsigned)g), dim3(kBlock), shm, s,   \
                           p->d_rowval, (const CT *)p->d_nzcolor, DEST, FXa, FXb, p->ldf, p->d_eps, c_lo, c_hi, \
                           outs[0], p->nnz_local, VOK)
        if (p->kind == K_CSC) {
            if (lds) FD_LAUNCH_LIST(false, 2, true, nullptr, vec_ok);
            else FD_LAUNCH_LIST(false, 2, false, nullptr, vec_ok);
        } else {
            if (lds) FD_LAUNCH_LIST(true, 2, true, p->d_dest, vec_ok & 4);
            else FD_LAUNCH_LIST(true, 2, false, p->d_dest, vec_ok & 4);
        }
#undef FD_LAUNCH_LIST
        break;
    }
    case K_TRIDIAG: {
        // row-window variant (chosen at plan creation, FD_INFO_WINDOW): few colours, an even first column; otherwise
        // the gather kernel below
        if (p->tri_window) {
            const int64_t nt = (p->col1 - p->col0 + kTriTile - 1) / kTriTile;
            const int64_t du0 = p->col0 > 0 ? p->col0 - 1 : 0;
            const int fxvec = (MODE != 0) || (fx == p->d_fx) || (fx == p->fx_batch_row) || (fx == p->d_zero);
            const int vok = ((((uintptr_t)outs[1]) & kPairMask) == 0 ? 1 : 0) | ((((uintptr_t)outs[0]) & kPairMask) == 0 ? 2 : 0) |
                            (((((uintptr_t)outs[2]) + sizeof(real_t) * (uintptr_t)(p->col0 - du0)) & kPairMask) == 0 ? 4 : 0) |
                            (tile_order_reversed() ? 8 : 0);
            const size_t shmt = sizeof(real_t) * ((size_t)(kTriTile + 4) * (size_t)B + kWinMaxCol);
            // (imag-only complex step: fx is the all-zero vector and is not loaded, see launch_window_m)
            const real_t *fxb_t = ((MODE == 0) && p->d_zero != nullptr && fx == p->d_zero) ? nullptr : FXb;
            hipLaunchKernelGGL((k_decompress_tridiag_window<CT, MODE, 4>), dim3((unsigned)(8 * xcd_chunks(nt))), dim3(kBlock),
                               shmt, s, color, FXa, fxb_t, p->ldf, p->d_eps, c_lo, c_hi, p->N, p->col0, p->col1, outs[0],
                               outs[1], outs[2], fxvec, vok);
            break;
        }
        const int g = grid_for(p->col1 - p->col0, kBlock, p->ctx->num_cus);
        hipLaunchKernelGGL((k_decompress_tridiag<CT, MODE>), dim3(g), dim3(kBlock), 0, s, color, FXa, FXb,
                           p->ldf, p->d_eps, c_lo, c_hi, p->N, p->col0, p->col1, outs[0], outs[1], outs[2]);
        break;
    }
    case K_BANDED: {
        if (p->window) {   // the band's storage order IS an entry list with implicit indices: same kernel as CSC
            launch_window_m<MODE>(p, fx, FXa, FXb, c_lo, c_hi, outs[0]);
            break;
        }
        const int g = grid_for((p->col1 - p->col0) * (p->l + p->u + 1), kBlock, p->ctx->num_cus);
        hipLaunchKernelGGL((k_decompress_banded<CT, MODE>), dim3(g), dim3(kBlock), 0, s, color, FXa, FXb,
                           p->ldf, p->d_eps, c_lo, c_hi, p->M, p->l, p->u, p->col0, p->col1, outs[0]);
        break;
    }
    case K_BBB: {
        const int64_t slots = p->N * (int64_t)(p->bbb_bl + p->bbb_bu + 1) * (p->bbb_lam + p->bbb_mu + 1);
        const int g = grid_for(slots, kBlock, p->ctx->num_cus);
        hipLaunchKernelGGL((k_decompress_bbb<CT, MODE>), dim3(g), dim3(kBlock), 0, s, color, FXa, FXb, p->ldf, p->d_eps, c_lo, c_hi, p->N,
                           p->bbb_nb, p->bbb_bl, p->bbb_bu, p->bbb_lam, p->bbb_mu, p->d_bbb_off, p->d_bbb_blk, p->d_bbb_start, p->d_bbb_stride,
                           outs[0]);
        break;
    }
    case K_DENSE: {
        const int g = grid_for((int64_t)B * p->M, kBlock, p->ctx->num_cus);
        hipLaunchKernelGGL((k_decompress_dense<MODE>), dim3(g), dim3(kBlock), 0, s, FXa, FXb, p->ldf, p->d_eps, c_lo,
                           c_hi, p->M, outs[0]);
        break;
    }
    case K_COLRANGE: {
        // one workgroup per 32 columns (one wave per column measured slower in round 2: 158 vs 127 us on config 5; the scalar
        // instantiation serves layouts whose rows / destinations are not all even)
        const int64_t nc = p->col1 - p->col0;
        // the imaginary parts of an imag-only complex step arrive as a real array with fx = the zero vector
        const real_t *fxb = (MODE == 0 && p->d_zero != nullptr && FXb == p->d_zero) ? nullptr : FXb;
        const bool vec = p->cr_pairs && (((uintptr_t)outs[0]) & kPairMask) == 0 && (MODE != 0 || fxb == nullptr || (((uintptr_t)fxb) & kPairMask) == 0);
        const dim3 gcr((unsigned)((nc + kCrCols - 1) / kCrCols));
        const int rev = tile_order_reversed() ? 1 : 0;
        if (nc > 0) {
            if (vec)
                hipLaunchKernelGGL((k_decompress_colrange_wg<CT, MODE, true>), gcr, dim3(kBlock), 0, s, color, p->d_cr_rlo,
                                   p->d_cr_cnt, p->d_cr_off, FXa, fxb, p->ldf, p->d_eps, c_lo, c_hi, p->col0, nc, outs[0], rev);
            else
                hipLaunchKernelGGL((k_decompress_colrange_wg<CT, MODE, false>), gcr, dim3(kBlock), 0, s, color, p->d_cr_rlo,
                                   p->d_cr_cnt, p->d_cr_off, FXa, fxb, p->ldf, p->d_eps, c_lo, c_hi, p->col0, nc, outs[0], rev);
        }
        break;
    }
    default: break;
    }
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

int launch_decompress(fd_plan *p, const real_t *fx, int c_lo, int c_hi, real_t *const *outs, int mode)
{
#define FD_DISPATCH(CT)                                                                 \
    switch (mode) {                                                                \
    case FD_FORWARD: return launch_decompress_tm<CT, 0>(p, fx, c_lo, c_hi, outs);       \
    case FD_CENTRAL: return launch_decompress_tm<CT, 1>(p, fx, c_lo, c_hi, outs);       \
    default: return launch_decompress_tm<CT, 2>(p, fx, c_lo, c_hi, outs);               \
    }
    if (p->color8) { FD_DISPATCH(uint8_t) } else { FD_DISPATCH(int32_t) }
#undef FD_DISPATCH
}

int launch_fill(fd_ctx *ctx, real_t *ptr, int64_t n, real_t v)
{
    if (n <= 0) return FD_OK;
    hipLaunchKernelGGL(k_fill, dim3(grid_for(n, kBlock, ctx->num_cus)), dim3(kBlock), 0, ctx->stream, ptr, n, v);
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

__global__ void k_scale(real_t *__restrict__ dst, const real_t *__restrict__ src, int64_t n, real_t factor)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = factor * src[i];
}

// dst[i] = factor * src[i] (the doubled step sizes of central differences handed over as f(+) - f(-): 2 eps is exact)
int launch_scale(fd_ctx *ctx, real_t *dst, const real_t *src, int64_t n, real_t factor)
{
    if (n <= 0) return FD_OK;
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, dst, src, n, factor);
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

int launch_stream_copy(fd_ctx *ctx, const void *src, void *dst, int64_t n16)
{
    // one 16-B element per thread, uncapped grid: the fastest copy geometry measured on MI355X (scripts/ubench)
    hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)((n16 + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                       (const r2_t *)src, (r2_t *)dst, n16);
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

}  // namespace fdjac
