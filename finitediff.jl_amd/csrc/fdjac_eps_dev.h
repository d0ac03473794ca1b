// Device code of the step-size reduction shared by its own launches (fdjac_kernels.hip) and by the FUSED step of the built-in storing
// launchers (fdjac_builtin_f.hip): level 0 of the defined two-level sum as a function, and the pieces of the one-launch Jacobian
// (round 6) -- reduction workgroups, ONE finisher workgroup, storing wavefronts that wait for the step sizes.
//
// The fused step hands values from workgroup to workgroup WITHOUT tickets or fences: every handed-over value is its own flag.  A slot
// holds a sentinel (a quiet NaN with a payload no arithmetic produces) until its producer overwrites it with ONE 8-byte (4-byte)
// agent-scope atomic store; the consumer polls the slot with agent-scope atomic loads until it reads something else.  No second word has
// to become visible in order, so nothing is drained and nothing is counted: the dependent chain of the round-5 form (block sum -> drain ->
// ticket -> load -> group sum -> drain -> ticket -> load -> eps: ~6 round trips through memory) shrinks to two (block sums -> finisher,
// step sizes -> storing wavefronts).  Slots are double-buffered by call parity: the finisher of call k resets the slots of call k + 1
// (nobody reads them any more: call k - 1 has completed on the stream).  The ORDER in which values are added is the reduction's
// definition (k_eps_partial_reg) and does not change: same bits.
//
// The SHARDED step (one process per GPU) is the same launch: a rank reduces its own groups, its finishers store the group sums straight
// into every peer's mailbox cells (system-scope stores over xGMI) and poll their own mailbox for the peers' -- again every cell its own
// flag, THREE buffers by epoch: a rank resets, in step e, its cells of step e + 2; a peer can be at most one step ahead (it cannot
// finish e + 1 without this rank's e + 1 sums), so it writes buffers e and e + 1 only, and it starts e + 2 only after this rank's
// launch e + 1 -- hence after launch e with its resets -- has completed.  The halo of a sharded x travels the same way.
//
// Progress: reduction workgroups and the finisher occupy the LOWEST block indices of the launch and never wait for a storing
// wavefront; storing wavefronts wait only for the finisher.  Workgroups are dispatched in index order, so everything a waiting
// wavefront depends on is resident or done; every wait is bounded by a wall-clock timeout that raises the plan's error word
// (FD_ERR_COMM from the next call) and stores NaNs instead of hanging the device.
#pragma once
#include "fdjac_internal.h"

namespace fdjac {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <typename CT> __device__ __forceinline__ void load_color_pair(const CT *p, int &c0, int &c1);
template <> __device__ __forceinline__ void load_color_pair<uint8_t>(const uint8_t *p, int &c0, int &c1)
{
    const unsigned v = *reinterpret_cast<const uint16_t *>(p);  // p is 2-B aligned (even index)
    c0 = (int)(v & 0xFF);
    c1 = (int)(v >> 8);
}
template <> __device__ __forceinline__ void load_color_pair<int32_t>(const int32_t *p, int &c0, int &c1)
{
    const int2 v = *reinterpret_cast<const int2 *>(p);
    c0 = v.x;
    c1 = v.y;
}

constexpr int kEpsU = 4;  // independent 16-B loads in flight per thread

// The reduction is DEFINED as a rank-aligned two-level sum, a function of N alone (see k_eps_partial_reg, fdjac_kernels.hip).
struct EpsGrid {
    int tpg, bpg, tpb;          // tiles per group, blocks per group, tiles per block
    int final_groups;           // > 0: this launch covers that many groups = all of them: its last group writes eps
    int C, is_forward;
    double relstep, absstep, dir;
};

// Level 0 for workgroup `gblock` of the global grid: per-thread accumulation over the block's tiles, the fixed 64-lane shuffle tree,
// the 4 waves in order.  Returns true on wave 0 only; there lane c < NC holds the block's sum of colour c in `s`.
template <typename CT, int NC, bool CYC, bool NT, bool PIPE = true>
__device__ __forceinline__ bool eps_block_sum(const real_t *__restrict__ x, const CT *__restrict__ color, int64_t n, int cyc_C, int cyc_shift,
                                              int gblock, const EpsGrid &eg, int pair, double (*red)[NC], double &s)
{
    const int grp = gblock / eg.bpg, kb = gblock - grp * eg.bpg;
    double acc[NC];   // sums of squares are accumulated in Float64 whatever the element type
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;

    // tile = kEpsU * 512 elements; pair u of thread t sits at tile + u*512 + 2t (dense per instruction)
    const int64_t tile = (int64_t)kEpsU * kBlock * 2;
    const int64_t t0 = (int64_t)grp * eg.tpg + (int64_t)kb * eg.tpb;
    int64_t t1 = t0 + eg.tpb;
    if (t1 > (int64_t)(grp + 1) * eg.tpg) t1 = (int64_t)(grp + 1) * eg.tpg;
    const int64_t base0 = t0 * tile;
    int64_t base_end = t1 * tile;
    if (base_end > n) base_end = n;
    // cyclic colours: colour of this thread's first element, then advanced by (512 mod C) per u
    int rc = 0, du = 0;
    if (CYC) {
        rc = (int)((base0 + threadIdx.x * 2 + cyc_shift) % cyc_C);
        du = (kBlock * 2) % cyc_C;
    }
    // (software-pipelined: the loads of tile k + 1 are in flight while tile k is accumulated -- a block of a sharded reduction walks
    //  up to 5 tiles, and one tile's 16 KB per workgroup in flight left the pass latency-bound; the ORDER of the additions is untouched)
    auto load_tile = [&](int64_t base, r2_t (&v)[kEpsU], int (&c0)[kEpsU], int (&c1)[kEpsU]) {
        if (base + tile <= n) {          // the whole tile lies inside x (wave-uniform; all but the last tile): no load inside a per-lane
                                         // branch -- inside one they are waited for one by one, and with them everything in flight
                                         // (c2 0.0172 -> 0.0166 ms, c3's reduction 26.5 -> 25.4 us, same box; THREE tiles in flight for
                                         // computed colours was measured too: the rank's reduction 6.1 -> 5.6 us but c4's 21.2 -> 23.7)
#pragma unroll
            for (int u = 0; u < kEpsU; ++u) {
                const int64_t i = base + (int64_t)u * kBlock * 2 + threadIdx.x * 2;
                if (CYC) {
                    c0[u] = rc;
                    c1[u] = rc + 1 == cyc_C ? 0 : rc + 1;
                    rc += du;
                    rc = rc >= cyc_C ? rc - cyc_C : rc;
                }
                const r2_t *pv = reinterpret_cast<const r2_t *>(x + i);
                v[u] = NT ? __builtin_nontemporal_load(pv) : *pv;
                if (!CYC) load_color_pair<CT>(color + i, c0[u], c1[u]);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < kEpsU; ++u) {
            const int64_t i = base + (int64_t)u * kBlock * 2 + threadIdx.x * 2;
            if (CYC) {
                c0[u] = rc;
                c1[u] = rc + 1 == cyc_C ? 0 : rc + 1;
                rc += du;
                rc = rc >= cyc_C ? rc - cyc_C : rc;
            }
            if (i + 1 < n) {
                if (NT) v[u] = __builtin_nontemporal_load(reinterpret_cast<const r2_t *>(x + i));
                else v[u] = *reinterpret_cast<const r2_t *>(x + i);
                if (!CYC) load_color_pair<CT>(color + i, c0[u], c1[u]);
            } else if (i < n) {
                v[u] = r2_t{x[i], 0.0};
                if (!CYC) c0[u] = color[i];
                c1[u] = -2;
            } else {
                v[u] = r2_t{0.0, 0.0};
                c0[u] = c1[u] = -2;
            }
        }
    };
    r2_t va[kEpsU], vb[kEpsU];
    int a0[kEpsU], a1[kEpsU], b0[kEpsU], b1[kEpsU];
    auto accumulate = [&](const r2_t (&v)[kEpsU], int (&c0)[kEpsU], int (&c1)[kEpsU]) {
#pragma unroll
        for (int u = 0; u < kEpsU; ++u) {
            if (pair) c1[u] = c0[u];     // complex-valued x: (re, im) of one coloured element -- |x_j|^2 = re^2 + im^2
            const double s0 = (double)v[u].x * (double)v[u].x, s1 = (double)v[u].y * (double)v[u].y;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                acc[c] += (c0[u] == c) ? s0 : 0.0;
                acc[c] += (c1[u] == c) ? s1 : 0.0;
            }
        }
    };
    if (!PIPE) {      // (one tile in flight: 32 registers fewer -- the fused single-GPU step, whose blocks have one or two tiles)
        for (int64_t base = base0; base < base_end; base += tile) {
            load_tile(base, va, a0, a1);
            accumulate(va, a0, a1);
        }
    } else {
    if (base0 < base_end) load_tile(base0, va, a0, a1);
    for (int64_t base = base0; base < base_end; base += 2 * tile) {
        const bool more1 = base + tile < base_end;
        if (more1) load_tile(base + tile, vb, b0, b1);
        accumulate(va, a0, a1);
        if (!more1) break;
        if (base + 2 * tile < base_end) load_tile(base + 2 * tile, va, a0, a1);
        accumulate(vb, b0, b1);
    }
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double w = wave_sum(acc[c]);
        if (lane == 0) red[wave][c] = w;
    }
    __syncthreads();
    if (wave != 0) return false;
    s = 0.0;
    if (lane < NC) {
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) s += red[w][lane];
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// the fused step
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int kFzReplicas = 64;      // the step sizes are published in this many places (one 256-B line each for Float64) so that the
constexpr int kFzPitch = 32;         //   polling wavefronts do not all hammer one memory channel

template <typename T> struct FzBits;
template <> struct FzBits<double> {
    typedef unsigned long long u_t;
    static constexpr u_t sentinel = kFzSentinel64, halo_sentinel = kFzSentinelHalo64;
};
template <> struct FzBits<float> {
    typedef unsigned u_t;
    static constexpr u_t sentinel = kFzSentinel32, halo_sentinel = kFzSentinel32;
};
typedef FzBits<real_t>::u_t rbits_t;
__device__ __forceinline__ real_t fz_from_bits(rbits_t b)
{
    real_t r;
    __builtin_memcpy(&r, &b, sizeof r);
    return r;
}
__device__ __forceinline__ rbits_t fz_to_bits(real_t r)
{
    rbits_t b;
    __builtin_memcpy(&b, &r, sizeof r);
    return b;
}

struct FusedEps {
    EpsGrid eg;
    int nblocks;                     // reduction workgroups of this launch (blockIdx C .. C + nblocks; blockIdx c < C: the finisher of colour c)
    int cyc_C, cyc_shift, pair;
    double *part, *part_next;        // [C][nblocks] block sums, colour-major: this call's parity / the next call's (reset here)
    rbits_t *epsr, *epsr_next;       // [kFzReplicas][kFzPitch] published step sizes, same double buffering
    real_t *eps, *eps2;              // the plan's plain arrays (fd_plan_get_epsilons; later launches of the call)
    int *err;                        // the plan's error word (pinned host memory)
    long long timeout_ticks;         // wall_clock64 ticks (100 MHz)
    long long *trace;                // NULL, or 16 words of wall_clock64 marks (fd_plan_fused_trace: where a launch's time goes)
    // the SHARDED step (nranks > 1): this launch reduces the groups [g0, g0 + ng) of rank `rank`; their sums go straight into the peers'
    // mailboxes and the peers' arrive in this rank's (cells of buffer `buf`, every one its own flag); the halo of a sharded x likewise
    int nranks, rank, g0, ng;
    char *const *peer;               // the peers' mailboxes as mapped here
    char *local;                     // this rank's mailbox
    long long fz_off;                // the fused step's cells inside a mailbox
    int buf, buf_reset;              // this step's buffer; the one it resets for the step after next (-1: loop-back, none)
    real_t *xw;                      // NULL, or x: sharded -- the halo cells [own_begin - halo, own_begin), [own_end, own_end + halo) arrive with the launch
    long long own_begin, own_end;
    int halo;
};
__device__ __forceinline__ char *fz_cells(char *mailbox, const FusedEps &fz, int buf) { return mailbox + fz.fz_off + (long long)buf * kFzBufBytes; }
// trace slots: 0 first reduction workgroup starts (min), 1 last block sum published (max), 2 finisher starts, 3 finisher has every block
// sum, 4 step sizes published, 5 first storing workgroup starts (min), 6 / 7 first / last storing workgroup has the step sizes, 8 last
// storing wavefront done (max), 9 last storing workgroup starts (max)
__device__ __forceinline__ void fz_mark_min(const FusedEps &fz, int k)
{
    if (fz.trace && (threadIdx.x & 63) == 0 && ((blockIdx.x & 31) < 2 || blockIdx.x < kRegColors)) atomicMin((unsigned long long *)fz.trace + k, (unsigned long long)wall_clock64());      // (one workgroup in 16 is sampled)
}
__device__ __forceinline__ void fz_mark_max(const FusedEps &fz, int k)
{
    if (fz.trace && (threadIdx.x & 63) == 0 && ((blockIdx.x & 31) < 2 || blockIdx.x < kRegColors)) atomicMax((unsigned long long *)fz.trace + k, (unsigned long long)wall_clock64());
}

// eight agent-scope 8-byte loads in flight together
__device__ __forceinline__ void fz_load8_agent(const unsigned long long *const (&p)[8], unsigned long long *v)
{
    asm volatile("global_load_dwordx2 %0, %8, off sc1\n\t"
                 "global_load_dwordx2 %1, %9, off sc1\n\t"
                 "global_load_dwordx2 %2, %10, off sc1\n\t"
                 "global_load_dwordx2 %3, %11, off sc1\n\t"
                 "global_load_dwordx2 %4, %12, off sc1\n\t"
                 "global_load_dwordx2 %5, %13, off sc1\n\t"
                 "global_load_dwordx2 %6, %14, off sc1\n\t"
                 "global_load_dwordx2 %7, %15, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
                 : "memory");
}

// a reduction workgroup of the fused step: level 0, published value by value (colour-major: the finisher of colour c reads
// part[c][0 .. nblocks) with dense loads)
template <int NC, bool PIPE, typename CT = uint8_t, bool CYC = true, bool NT = false>
__device__ __forceinline__ void fused_eps_block(const real_t *__restrict__ x, int64_t n, const FusedEps &fz, int gblock, double (*red)[NC],
                                                const CT *__restrict__ color = nullptr)
{
    double s;
    if (threadIdx.x == 0) fz_mark_min(fz, 0);
    // (gblock counts this launch's reduction workgroups; the block of the GLOBAL grid it reduces starts at the rank's first group)
    if (!eps_block_sum<CT, NC, CYC, NT, PIPE>(x, color, n, fz.cyc_C, fz.cyc_shift, fz.g0 * fz.eg.bpg + gblock, fz.eg, fz.pair, red, s)) return;
    const int lane = threadIdx.x & 63;
    if (lane < fz.eg.C) __hip_atomic_store(fz.part + (int64_t)lane * fz.nblocks + gblock, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fz_mark_max(fz, 1);
}

// the finisher of colour c (blockIdx c), by wavefront:
//   0  waits for the colour's block sums -- DENSE loads, lane l takes part[c][l + 64 m], all in flight together -- parks them in its
//      private LDS window, lane g adds group g's in block order (level 1); a sharded step stores every own group's sum into every
//      peer's cell (c, g) right away (a rank must send before it can receive: all ranks run this code);
//   1  (sharded) meanwhile waits for the peers' cells (c, .) of this rank's mailbox -- one dense 512-byte load per poll;
//   2, 3  housekeeping: the next step's slots back to the sentinel, the halo of a sharded x into the neighbours' cells;
//   then ONE barrier, and wavefront 0 adds the 64 group sums in group order (level 2), forms eps[c] and publishes it.
// (Round-6 measurements, N = 10^6 / a rank of 8 at N = 10^7, time from the last block sum to the published step size: one workgroup
//  for all colours, thread (group, colour pair) loading its own addends: 4.1 us / - (1024 scattered 8-byte requests through one CU's
//  memory pipeline); one workgroup per colour, dense loads, two barriers, mailbox polled after the first: 2.0 / 4.0 us; ONE wavefront
//  per colour with per-lane address lists: 3.6 / 6-9 us (scattered again); this form: profiles/r06_fused_trace.md.)
// lds: kFzMaxBlocks + 2 * kEpsGroups doubles.
constexpr int kFzMaxBlocks = kEpsGroups * kEpsBlocksPerGroup;      // 1024
__device__ __forceinline__ void fused_finisher(const FusedEps &fz, int c, double *lds)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nb = fz.nblocks, bpg = fz.eg.bpg;
    const bool sharded = fz.nranks > 1;
    double *gs_peer = lds + kFzMaxBlocks;        // [64] the peers' group sums (wavefront 1)
    int *s_bad = reinterpret_cast<int *>(gs_peer + kEpsGroups);
    if (t == 0) *s_bad = 0;
    double gs = 0.0;
    int ok = 1;
    const long long t0 = wall_clock64();
    if (wave == 0) {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(fz.part + (int64_t)c * nb);
        if (t == 0) fz_mark_max(fz, 2);
        const unsigned long long my_peer = (sharded && lane < fz.nranks) ? (unsigned long long)fz.peer[lane] : 0ull;
        // (eight loads per statement, ONE wait: the compiler waits behind every __hip_atomic_load -- eight dependent round trips instead of
        //  one, measured: 3.8 us for 512 block sums; lanes past the end re-read the colour's first slot.  512 block sums per round.)
        for (int m0 = 0; m0 * 64 < nb; m0 += 8) {
            unsigned long long v[8];
            const unsigned long long *q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = src + ((lane + 64 * (m0 + u) < nb) ? lane + 64 * (m0 + u) : 0);
            for (;;) {
                fz_load8_agent(q, v);
                bool all = true;
#pragma unroll
                for (int u = 0; u < 8; ++u) all = all && v[u] != kFzSentinel64;
                if (__all(all)) break;
                if (wall_clock64() - t0 > fz.timeout_ticks) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (lane + 64 * (m0 + u) < nb) lds[lane + 64 * (m0 + u)] = __longlong_as_double((long long)v[u]);
        }
        __builtin_amdgcn_wave_barrier();     // (one wavefront: its LDS instructions execute in order)
        if (t == 0) fz_mark_max(fz, 3);
        {
            // (the group's block sums: all LDS reads first, then the additions in block order -- "+ 0.0" past the last block changes nothing)
            const bool own = lane >= fz.g0 && lane < fz.g0 + fz.ng;
#pragma unroll
            for (int k0 = 0; k0 < kEpsBlocksPerGroup; k0 += 8) {
                double w[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) w[k] = (own && k0 + k < bpg) ? lds[(lane - fz.g0) * bpg + k0 + k] : 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) gs += w[k];
            }
            if (sharded)
                for (int b = 0; b < fz.nranks; ++b) {
                    // (the peers' mailbox addresses were requested before the wait: lane b holds peer b's)
                    char *pb = reinterpret_cast<char *>(__shfl((unsigned long long)my_peer, b, 64));
                    if (own && b != fz.rank)
                        __hip_atomic_store(reinterpret_cast<double *>(fz_cells(pb, fz, fz.buf)) + c * kEpsGroups + lane, gs, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
                }
        }
    } else if (wave == 1) {
        if (sharded) {
            const bool own = lane >= fz.g0 && lane < fz.g0 + fz.ng;
            const unsigned long long *cell = reinterpret_cast<const unsigned long long *>(fz_cells(fz.local, fz, fz.buf)) + c * kEpsGroups + lane;
            unsigned long long w = 0;
            for (;;) {
                w = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (__all(own || w != kFzSentinel64)) break;
                if (wall_clock64() - t0 > fz.timeout_ticks) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            gs_peer[lane] = __longlong_as_double((long long)w);
        }
    } else {
        // ---- wavefronts 2, 3: housekeeping, no dependence on anything ----
        const int u = t - 128;
        if (sharded && c == 0 && fz.xw && u < 2 * fz.halo) {
            const int h = u % fz.halo, up = u / fz.halo;      // up: to rank + 1 (my last elements = its lower halo); else to rank - 1
            const int target = up ? fz.rank + 1 : fz.rank - 1;
            if (target >= 0 && target < fz.nranks) {
                const real_t v = up ? fz.xw[fz.own_end - fz.halo + h] : fz.xw[fz.own_begin + h];
                rbits_t *cell = reinterpret_cast<rbits_t *>(fz_cells(fz.peer[target], fz, fz.buf) + kFzGsumBytes + (up ? 0 : kFzHaloBytes)) + h;
                __hip_atomic_store(cell, fz_to_bits(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        unsigned long long *pn = reinterpret_cast<unsigned long long *>(fz.part_next + (int64_t)c * nb);
        for (int i = u; i < nb; i += 128) __hip_atomic_store(pn + i, kFzSentinel64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (u < kFzReplicas) __hip_atomic_store(fz.epsr_next + u * kFzPitch + c, FzBits<real_t>::sentinel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sharded && fz.buf_reset >= 0) {      // this rank's cells of the step after next (nobody writes them before this launch is over)
            unsigned long long *cells = reinterpret_cast<unsigned long long *>(fz_cells(fz.local, fz, fz.buf_reset));
            if (u < kEpsGroups) __hip_atomic_store(cells + c * kEpsGroups + u, kFzSentinel64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (c == 0 && u >= 64 && u < 64 + (int)(2 * kFzHaloBytes / 8))
                __hip_atomic_store(cells + kFzGsumBytes / 8 + (u - 64), kFzSentinelHalo64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (!ok) *s_bad = 1;
    // (a bare barrier behind the LDS writes: __syncthreads() is also a fence -- it would wait for the acknowledgement of every store
    //  above, the system-scope stores into the peers' cells among them: 2 us on the way to the step sizes, measured)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wave != 0) return;
    const bool bad = *s_bad != 0;
    // level 2: the 64 group sums in group order (every lane forms the same total)
    if (!(lane >= fz.g0 && lane < fz.g0 + fz.ng)) gs = gs_peer[lane];
    gs_peer[lane] = gs;                       // (all 64 group sums side by side: every lane reads them back as broadcasts -- the reads
    __builtin_amdgcn_wave_barrier();          //  pipeline, 64 cross-lane shuffles do not)
    double tot = 0.0;
#pragma unroll
    for (int k0 = 0; k0 < kEpsGroups; k0 += 16) {
        double w[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) w[k] = gs_peer[k0 + k];
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += w[k];
    }
    real_t e = eps_rule<real_t>(tot, fz.eg.relstep, fz.eg.absstep, fz.eg.dir, fz.eg.is_forward);
    if (bad) e = fz_from_bits(FzBits<real_t>::sentinel ^ 1);      // (a NaN that is not the sentinel: the storing wavefronts go on and store NaNs)
    __hip_atomic_store(fz.epsr + lane * kFzPitch + c, fz_to_bits(e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // replica `lane`
    if (t == 0) {
        if (bad) __hip_atomic_store(fz.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        fz.eps[c] = e;
        if (fz.eps2) fz.eps2[c] = (real_t)2 * e;
        fz_mark_max(fz, 4);
    }
}

// the halo of a sharded x as a storing wavefront at the edge of the rank's columns gets it: element h of the lower (which = 0) / upper
// (1) halo, polled from this rank's own cell (the neighbour's finisher stores it there) and written into x as fd_plan_set_halo promises
__device__ __forceinline__ real_t fused_halo(const FusedEps &fz, int which, int h)
{
    const rbits_t *cell = reinterpret_cast<const rbits_t *>(fz_cells(fz.local, fz, fz.buf) + kFzGsumBytes + (which ? kFzHaloBytes : 0)) + h;
    const long long t0 = wall_clock64();
    rbits_t b;
    for (;;) {
        b = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (b != FzBits<real_t>::halo_sentinel) break;
        if (wall_clock64() - t0 > fz.timeout_ticks) {
            __hip_atomic_store(fz.err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            b = FzBits<real_t>::sentinel ^ 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    const real_t v = fz_from_bits(b);
    fz.xw[which ? fz.own_end + h : fz.own_begin - fz.halo + h] = v;
    return v;
}

// a storing workgroup's wait for the step sizes: ONE wavefront polls (lane c its colour's slot of the workgroup's replica) and parks
// them in LDS; the caller's barrier releases the others.  (Every wavefront polling for itself -- 7 800 of them at N = 10^6 -- kept the
// memory system busy enough to delay what they were waiting for: 18 us per launch instead of 14.)
__device__ __forceinline__ void fused_wait_eps(const FusedEps &fz, int wg_id, real_t *s_eps)
{
    const int lane = threadIdx.x & 63;
    const rbits_t *E = fz.epsr + (wg_id & (kFzReplicas - 1)) * kFzPitch;
    const int c = lane < fz.eg.C ? lane : 0;
    long long t0 = 0;
    rbits_t b = 0;
    fz_mark_min(fz, 5);
    fz_mark_max(fz, 9);
    for (int spin = 0;; ++spin) {
        b = __hip_atomic_load(E + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(b != FzBits<real_t>::sentinel)) break;
        if (spin == 0) t0 = wall_clock64();
        if ((spin & 15) == 15 && wall_clock64() - t0 > fz.timeout_ticks) {
            if (lane == 0) __hip_atomic_store(fz.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            b = FzBits<real_t>::sentinel ^ 1;
            break;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    if (lane < kRegColors) s_eps[lane] = fz_from_bits(b);
    fz_mark_min(fz, 6);
    fz_mark_max(fz, 7);
}

}  // namespace fdjac
