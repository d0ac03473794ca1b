// Built-in device f! families behind the fd_f_launch boundary: the reference's own test
// fixtures (test/coloring_tests.jl:5-13, 99-108, 124-133) and the benchmark configurations of
// BASELINE.json.  Each kernel evaluates `nbatch` independent points (grid.y = point index) and
// only the rows [row_begin,row_end) the plan will consume.  Real and complex (for the
// complex-step arm, src/jacobians.jl:623-648) instantiations share one template.
//
// Operation order follows the fixtures literally (e.g. (x[i-1] - 2x[i]) + x[i+1]) and the
// library is built with -ffp-contract=off, so a linear fixture reproduces the CPU values bit
// for bit.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "fdjac_internal.h"
#include "fdjac_eps_dev.h"

namespace fdjac {

constexpr real_t kTwo = 2, kFour = 4;   // literals in the element type (Julia's 2x / 4x stay in eltype(x))
struct cd {
    real_t re, im;
};
__device__ __forceinline__ cd operator+(cd a, cd b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cd operator*(cd a, cd b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cd operator*(real_t s, cd a) { return {s * a.re, s * a.im}; }
__device__ __forceinline__ cd operator+(cd a, real_t s) { return {a.re + s, a.im}; }
__device__ __forceinline__ cd operator-(cd a, real_t s) { return {a.re - s, a.im}; }

template <typename T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ real_t zero_of<real_t>() { return 0.0; }
template <> __device__ __forceinline__ cd zero_of<cd>() { return {0.0, 0.0}; }

__device__ __forceinline__ real_t sin_of(real_t a) { return sin(a); }
__device__ __forceinline__ cd sin_of(cd a)
{
    // sin(a+ib) = sin a cosh b + i cos a sinh b
    return {sin(a.re) * cosh(a.im), cos(a.re) * sinh(a.im)};
}

// dx[i] = x[i-1] - 2x[i] + x[i+1]  (+ x[i]^2 * x[i+1] when NL), zero beyond the ends.
template <typename T, bool NL>
__global__ void __launch_bounds__(kBlock)
k_f_tridiag(T *__restrict__ fx, const T *__restrict__ x, int64_t n, int64_t xs, int64_t fs, int64_t r0, int64_t r1)
{
    const T *xb = x + (int64_t)blockIdx.y * xs;
    T *fb = fx + (int64_t)blockIdx.y * fs;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = r0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < r1; i += stride) {
        const T xm = i > 0 ? xb[i - 1] : zero_of<T>();
        const T xp = i + 1 < n ? xb[i + 1] : zero_of<T>();
        const T xi = xb[i];
        T v = (xm - kTwo * xi) + xp;
        if (NL) v = v + (xi * xi) * xp;
        fb[i] = v;
    }
}

// Same fixture, two rows per thread: one aligned 16-B load of (x[i], x[i+1]) plus the two scalar
// neighbours (L1 hits), one 16-B store.  Needs even i and 16-B aligned batch bases.
template <bool NL>
__global__ void __launch_bounds__(kBlock)
k_f_tridiag_v2(real_t *__restrict__ fx, const real_t *__restrict__ x, int64_t n, int64_t xs, int64_t fs, int64_t r0,
               int64_t r1)
{
    const real_t *xb = x + (int64_t)blockIdx.y * xs;
    real_t *fb = fx + (int64_t)blockIdx.y * fs;
    const int64_t stride = (int64_t)gridDim.x * kBlock * 2;
    for (int64_t i = r0 + ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 2; i < r1; i += stride) {
        if (i + 1 < n) {
            const r2_t c = *reinterpret_cast<const r2_t *>(xb + i);
            const real_t xm = i > 0 ? xb[i - 1] : 0.0;
            const real_t xq = i + 2 < n ? xb[i + 2] : 0.0;
            real_t v0 = (xm - kTwo * c.x) + c.y;
            real_t v1 = (c.x - kTwo * c.y) + xq;
            if (NL) {
                v0 = v0 + (c.x * c.x) * c.y;
                v1 = v1 + (c.y * c.y) * xq;
            }
            *reinterpret_cast<r2_t *>(fb + i) = r2_t{v0, v1};
        } else {
            const real_t xm = i > 0 ? xb[i - 1] : 0.0;
            const real_t xi = xb[i];
            fb[i] = (xm - kTwo * xi) + 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Lazy-point evaluation (fd_f_launch_lazy): each thread loads the base x values and colours its
// rows depend on ONCE, then evaluates the fixture at every point of the batch by perturbing in
// registers -- x~ = x + eps_b*[color==b] (src/jacobians.jl:562 / 603-604 / 633) -- so the perturbed
// points never exist in memory and x is read once for all colours.  Values are bit-identical to
// evaluating the materialised points.  MODE 0 forward, 1 central (+ then -), 2 complex step.
// ---------------------------------------------------------------------------------------------
// the wavefront's index in its workgroup, told to the compiler as the wave-uniform value it is: tile numbers, row starts, interior / edge
// verdicts and base addresses derived from it then live in scalar registers and are computed by the scalar unit.  One box, events
// (scripts/ab_all.sh): block-coupled store 49.6 -> 48.1 us (Float32 29.4 -> 28.9), Float32 5-point store 57.9 -> 57.2 -- used there;
// the Float64 5-point store 100 -> 127-132 us (its interior / edge / boundary window loads become real branches) and the tridiagonal
// stores unchanged -- not used there.
__device__ __forceinline__ int wave_index_scalar() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// the subtraction of fd_lazy_points.diff: an IEEE a - b of the two stored values, never contracted into a producer
__device__ __forceinline__ real_t sub_exact(real_t a, real_t b)
{
#pragma clang fp contract(off)
    return a - b;
}

template <typename T, bool NL> __device__ __forceinline__ T tridiag_row(T xm, T xi, T xp)
{
    T v = (xm - kTwo * xi) + xp;
    if (NL) v = v + (xi * xi) * xp;
    return v;
}

template <typename CT, int MODE, bool NL>
__global__ void __launch_bounds__(kBlock)
k_f_tridiag_lazy(real_t *__restrict__ fx, int64_t fs, real_t *__restrict__ base_out, const real_t *__restrict__ x,
                 const CT *__restrict__ color, const real_t *__restrict__ eps, int c_lo, int B, int64_t n, int64_t r0,
                 int64_t r1, int imag_only, int diff)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock * 2;
    for (int64_t i = r0 + ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 2; i < r1; i += stride) {
        // base values x[i-1..i+2] and their colours relative to the batch (-1: never perturbed)
        real_t xv[4];
        int cv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t j = i - 1 + k;
            const bool in = (j >= 0) & (j < n);
            xv[k] = in ? x[in ? j : 0] : 0.0;
            const int c = in ? (int)color[in ? j : 0] : -1;
            cv[k] = (c == (int)(CT)(-1) || c < 0) ? -1 : c - c_lo;   // "none" is all-ones in CT
        }
        const bool two = i + 1 < n;
        real_t b0 = 0.0, b1 = 0.0;                    // f(x) at rows i, i+1 (written to base_out, or the subtrahend of diff)
        if (base_out || (diff && MODE == 0)) {
            b0 = tridiag_row<real_t, NL>(xv[0], xv[1], xv[2]);
            if (two) b1 = tridiag_row<real_t, NL>(xv[1], xv[2], xv[3]);
        }
        if (base_out) {
            if (two) *reinterpret_cast<r2_t *>(base_out + i) = r2_t{b0, b1};
            else base_out[i] = b0;
        }
        for (int b = 0; b < B; ++b) {
            const real_t e = eps[c_lo + b];
            real_t d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = (cv[k] == b) ? e : 0.0;
            if (MODE == 2) {
                cd p[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) p[k] = cd{xv[k], d[k]};
                const cd v0 = tridiag_row<cd, NL>(p[0], p[1], p[2]);
                if (imag_only) {   // imaginary parts as a real array (fd_lazy_points.imag_only)
                    real_t *dst = fx + (int64_t)b * fs + i;
                    if (two) {
                        const cd v1 = tridiag_row<cd, NL>(p[1], p[2], p[3]);
                        *reinterpret_cast<r2_t *>(dst) = r2_t{v0.im, v1.im};
                    } else {
                        dst[0] = v0.im;
                    }
                } else {
                    real_t *dst = fx + ((int64_t)b * fs + i) * 2;
                    *reinterpret_cast<r2_t *>(dst) = r2_t{v0.re, v0.im};
                    if (two) {
                        const cd v1 = tridiag_row<cd, NL>(p[1], p[2], p[3]);
                        *reinterpret_cast<r2_t *>(dst + 2) = r2_t{v1.re, v1.im};
                    }
                }
            } else if (diff) {
                // differences (fd_lazy_points.diff): f(x + d) - f(x), or f(x + d) - f(x - d), one array per colour
                real_t p[4], q[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { p[k] = xv[k] + d[k]; q[k] = xv[k] - d[k]; }
                const real_t s0 = MODE == 1 ? tridiag_row<real_t, NL>(q[0], q[1], q[2]) : b0;
                const real_t v0 = sub_exact(tridiag_row<real_t, NL>(p[0], p[1], p[2]), s0);
                real_t v1 = 0.0;
                if (two) {
                    const real_t s1 = MODE == 1 ? tridiag_row<real_t, NL>(q[1], q[2], q[3]) : b1;
                    v1 = sub_exact(tridiag_row<real_t, NL>(p[1], p[2], p[3]), s1);
                }
                real_t *dst = fx + (int64_t)b * fs + i;
                if (two) *reinterpret_cast<r2_t *>(dst) = r2_t{v0, v1};
                else dst[0] = v0;
            } else {
#pragma unroll
                for (int sgn = 0; sgn < (MODE == 1 ? 2 : 1); ++sgn) {
                    real_t p[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) p[k] = sgn == 0 ? xv[k] + d[k] : xv[k] - d[k];
                    const real_t v0 = tridiag_row<real_t, NL>(p[0], p[1], p[2]);
                    real_t *dst = fx + (int64_t)(sgn * B + b) * fs + i;
                    if (two) {
                        const real_t v1 = tridiag_row<real_t, NL>(p[1], p[2], p[3]);
                        *reinterpret_cast<r2_t *>(dst) = r2_t{v0, v1};
                    } else {
                        dst[0] = v0;
                    }
                }
            }
        }
    }
}

// fd_lazy_points.store (include/fdjac_device.h): the tridiagonal fixture evaluated at the lazily perturbed points of a batch,
// difference quotients formed and stored into the banded Jacobian by this launch -- nothing follows it.  Operations per entry:
// those of the plain path (sub_exact, IEEE division), so the stored values have its bits.
//
// k_f_tridiag_store_wave (round 3; the default for an exactly tridiagonal band with all colours in one batch): COLUMN-centric --
// lane t of a wavefront owns the columns j = jw + 2t, j + 1, loads x[j-2 .. j+3] as three aligned 16-B pairs, evaluates the
// fixture on each column's three rows at x and at x +- eps_c e_j (the points differ from the colour's point x +- eps_c mask_c
// only in columns of colour c that do not touch these rows -- the plan verified C >= 3 cyclic colours on the exact band -- so
// every operand, including the "+ 0.0" of the unperturbed coordinates, and every operation is that of the coloured
// evaluation: same bits), and hands its six quotients to fd_band_emit_wave: staged in a wave-private LDS window in storage
// order, written as dense aligned non-temporal 16-B stores.  No workgroup barrier, no colour reads (cyclic colours are
// arithmetic), no position arithmetic per entry.  N = 10^7: 53-60 us for 320 MB (0.67-0.76 of the 8 TB/s peak; a pure
// 1:3 read:write stream of the same bytes takes 50 us) against 104-114 us for round 2's forms below, whose waves lived
// ~6 us each for one 8- or 16-B x load (scripts/ubench/fused_store_probe.hip, profiles/r03_a_fused_store_probe.txt).
//
// k_f_tridiag_lazy_store (round 2; kept for wider bands, colour chunks and colour ownership, CSC only): rows owned by
// workgroups, differences -> LDS [colour][row], the workgroup walks its storage positions.
__device__ __forceinline__ r2_t ld_pair_guarded(const real_t *__restrict__ x, int64_t a, int64_t n)
{
    if (a >= 0 && a + 1 < n) return *reinterpret_cast<const r2_t *>(x + a);
    r2_t v = {0, 0};
    if (a >= 0 && a < n) v.x = x[a];
    return v;
}

// the six quotients of a lane's two columns (rows j-1 .. j+1 of column j, then of column j + 1)
template <int MODE, bool NL>
__device__ __forceinline__ void tridiag_pair_quotients(const r2_t &L, const r2_t &Cc, const r2_t &R, real_t ea, real_t eb, real_t (&q)[6])
{
    const real_t xv[6] = {L.x, L.y, Cc.x, Cc.y, R.x, R.y};
#pragma unroll
    for (int o = 0; o < 2; ++o) {                             // column j + o: x[j + o] +- e, every other coordinate x + 0.0 / x - 0.0
        const real_t e = o == 0 ? ea : eb;
        const real_t ed = MODE == 1 ? 2 * e : e;
        real_t p[5], m[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { const real_t dlt = k == 2 ? e : (real_t)0; p[k] = xv[o + k] + dlt; m[k] = xv[o + k] - dlt; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const real_t plus = tridiag_row<real_t, NL>(p[k], p[k + 1], p[k + 2]);
            const real_t sub = MODE == 1 ? tridiag_row<real_t, NL>(m[k], m[k + 1], m[k + 2])
                                         : tridiag_row<real_t, NL>(xv[o + k], xv[o + k + 1], xv[o + k + 2]);
            q[3 * o + k] = sub_exact(plus, sub) / ed;      // (the shared-reciprocal division measured 2 % slower here: bandwidth, not issue, bounds this kernel)
        }
    }
}

template <int MODE, bool NL>
__global__ void __launch_bounds__(kBlock)
k_f_tridiag_store_wave(const real_t *__restrict__ x, const real_t *__restrict__ eps, int64_t n, fd_band_store bst, int64_t jstart)
{
    __shared__ __attribute__((aligned(16))) real_t s_win[kBlock / 64][FD_BAND_WAVE_LDS(3)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nwaves = (bst.col_end - jstart + 127) / 128;
    const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    if (gw >= nwaves) return;
    const int64_t jw = jstart + gw * 128;                    // even: the x pairs are 16-B aligned
    const int64_t j = jw + 2 * lane;
    const r2_t Cc = ld_pair_guarded(x, j, n), L = ld_pair_guarded(x, j - 2, n), R = ld_pair_guarded(x, j + 2, n);
    // cyclic colours: column j has colour (j + shift) mod C, column j + 1 the next one
    const int c0w = (int)((jw + bst.shift) % bst.C);         // (wave-uniform 64-bit modulo: scalar unit)
    const int c0 = (int)((uint32_t)(c0w + 2 * lane) % (uint32_t)bst.C);
    const int c1 = c0 + 1 == bst.C ? 0 : c0 + 1;
    const real_t ea = eps[c0], eb = eps[c1];
    real_t q[6];
    tridiag_pair_quotients<MODE, NL>(L, Cc, R, ea, eb, q);
    fd_band_emit_wave<real_t, 3, true>(&bst, s_win[wave], jw, q);      // (non-temporal: nothing re-reads nzval in this call)
}

// THE FUSED STEP (round 6): the whole Jacobian in ONE launch -- blockIdx 0 is the finisher of the step-size reduction, blockIdx
// 1 .. nblocks are its reduction workgroups (fdjac_eps_dev.h), everything after that stores: the wavefronts of k_f_tridiag_store_wave,
// which load their x, then wait for the step sizes of their colours to be published.  Same operations per value as the two-launch
// form: same bits.  N = 10^6: one launch instead of two, and a hand-off of two memory round trips instead of six.
template <int MODE, bool NL, int NC, bool PIPE>
__global__ void __launch_bounds__(kBlock)
k_f_tridiag_fused(const real_t *__restrict__ x, int64_t n, fd_band_store bst, int64_t jstart, FusedEps fz)
{
    // one LDS area, carved by role: storing workgroups (4 wave windows + the step sizes), reduction workgroups, finishers
    constexpr int kWinBytes = (kBlock / 64) * FD_BAND_WAVE_LDS(3) * (int)sizeof(real_t);
    constexpr int kLdsBytes = kWinBytes + 64 > (kFzMaxBlocks + 2 * kEpsGroups) * 8 ? kWinBytes + 64 : (kFzMaxBlocks + 2 * kEpsGroups) * 8;
    __shared__ __attribute__((aligned(16))) char s_lds[kLdsBytes];
    real_t (*s_win)[FD_BAND_WAVE_LDS(3)] = reinterpret_cast<real_t (*)[FD_BAND_WAVE_LDS(3)]>(s_lds);
    double *s_red = reinterpret_cast<double *>(s_lds);
    const int b = (int)blockIdx.x, nfin = fz.eg.C;
    if (b < nfin) { fused_finisher(fz, b, s_red); return; }
    if (b < nfin + fz.nblocks) { fused_eps_block<NC, PIPE>(x, n, fz, b - nfin, reinterpret_cast<double (*)[NC]>(s_red)); return; }
    // a storing workgroup: its wavefronts walk the 128-column tiles gw, gw + stride, ... (the launcher keeps the whole grid resident,
    // so nobody is dispatched after the step sizes are out); the first tile's x is in flight while the workgroup waits
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = b - nfin - fz.nblocks;
    const int64_t nwaves = (bst.col_end - jstart + 127) / 128;
    const int64_t gstride = (int64_t)((int)gridDim.x - nfin - fz.nblocks) * (kBlock / 64);
    int64_t gw = (int64_t)cb * (kBlock / 64) + wave;
    int64_t jw = jstart + gw * 128;
    r2_t Cc = {0, 0}, L = {0, 0}, R = {0, 0};
    // (a sharded x: the two lanes at the edges of the rank's columns take the neighbours' halo from the mailbox cells)
    auto load_x = [&](int64_t jt, r2_t &c_, r2_t &l_, r2_t &r_) {
        const int64_t j = jt + 2 * lane;
        c_ = ld_pair_guarded(x, j, n); l_ = ld_pair_guarded(x, j - 2, n); r_ = ld_pair_guarded(x, j + 2, n);
        if (fz.xw && fz.nranks > 1) {
            if (j == fz.own_begin && fz.rank > 0) { l_.x = fused_halo(fz, 0, 0); l_.y = fused_halo(fz, 0, 1); }
            if (j + 2 == fz.own_end && fz.rank + 1 < fz.nranks) { r_.x = fused_halo(fz, 1, 0); r_.y = fused_halo(fz, 1, 1); }
        }
    };
    if (gw < nwaves) load_x(jw, Cc, L, R);
    real_t *s_eps = reinterpret_cast<real_t *>(s_lds + kWinBytes);
    if (wave == 0) fused_wait_eps(fz, cb, s_eps);
    __syncthreads();
    for (; gw < nwaves; gw += gstride) {
        const int c0w = (int)((jw + bst.shift) % bst.C);
        const int c0 = (int)((uint32_t)(c0w + 2 * lane) % (uint32_t)bst.C);
        const int c1 = c0 + 1 == bst.C ? 0 : c0 + 1;
        real_t q[6];
        tridiag_pair_quotients<MODE, NL>(L, Cc, R, s_eps[c0], s_eps[c1], q);
        fd_band_emit_wave<real_t, 3, true>(&bst, s_win[wave], jw, q);
        // (the next tile's x requested BEFORE this tile is computed was measured: 12 more registers -- 6 instead of 8 wavefronts per
        //  SIMD -- for 0.3 us of a rank's 7.5-us storing phase, and 0.8 us lost at N = 10^6)
        jw = jstart + (gw + gstride) * 128;
        if (gw + gstride < nwaves) load_x(jw, Cc, L, R);
    }
    if (fz.trace && (blockIdx.x & 31) < 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); fz_mark_max(fz, 8); }
}

#ifdef FDJAC_F32
// Float32: the same kernel with FOUR columns per lane -- x as three aligned 16-B quads, twelve quotients, fd_band_emit_wave4's 16-B
// stores (a wavefront of the pair form moves 1.5 KB, half of the Float64 wavefront's bytes for the same instructions: N = 10^7
// 34.6 us for 160 MB).  Same operations per entry as the pair form: same bits.
typedef real_t r4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ r4_t ld_quad_guarded(const real_t *__restrict__ x, int64_t a, int64_t n)
{
    if (a >= 0 && a + 3 < n) return *reinterpret_cast<const r4_t *>(x + a);
    r4_t v = {0, 0, 0, 0};
    if (a >= 0 && a < n) v.x = x[a];
    if (a + 1 >= 0 && a + 1 < n) v.y = x[a + 1];
    if (a + 2 >= 0 && a + 2 < n) v.z = x[a + 2];
    if (a + 3 >= 0 && a + 3 < n) v.w = x[a + 3];
    return v;
}
template <int MODE, bool NL>
__device__ __forceinline__ void tridiag_quad_quotients(const r4_t &L, const r4_t &Cc, const r4_t &R, const real_t (&ev)[4], real_t (&q)[12])
{
    const real_t xv[8] = {L.z, L.w, Cc.x, Cc.y, Cc.z, Cc.w, R.x, R.y};      // x[j - 2 .. j + 5]
#pragma unroll
    for (int o = 0; o < 4; ++o) {                             // column j + o: x[j + o] +- e, every other coordinate x + 0.0 / x - 0.0
        const real_t e = ev[o];
        const real_t ed = MODE == 1 ? 2 * e : e;
        real_t p[5], m[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { const real_t dlt = k == 2 ? e : (real_t)0; p[k] = xv[o + k] + dlt; m[k] = xv[o + k] - dlt; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const real_t plus = tridiag_row<real_t, NL>(p[k], p[k + 1], p[k + 2]);
            const real_t sub = MODE == 1 ? tridiag_row<real_t, NL>(m[k], m[k + 1], m[k + 2])
                                         : tridiag_row<real_t, NL>(xv[o + k], xv[o + k + 1], xv[o + k + 2]);
            q[3 * o + k] = sub_exact(plus, sub) / ed;
        }
    }
}
template <int MODE, bool NL>
__global__ void __launch_bounds__(kBlock)
k_f_tridiag_store_wave4(const real_t *__restrict__ x, const real_t *__restrict__ eps, int64_t n, fd_band_store bst, int64_t jstart)
{
    __shared__ __attribute__((aligned(16))) real_t s_win[kBlock / 64][FD_BAND_WAVE4_LDS(3)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t nwaves = (bst.col_end - jstart + 255) / 256;
    const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + wave;
    if (gw >= nwaves) return;
    const int64_t jw = jstart + gw * 256;                    // a multiple of 4: the x quads are 16-B aligned
    const int64_t j = jw + 4 * lane;
    const r4_t Cc = ld_quad_guarded(x, j, n), L = ld_quad_guarded(x, j - 4, n), R = ld_quad_guarded(x, j + 4, n);
    const int c0w = (int)((jw + bst.shift) % bst.C);
    int cc = (int)((uint32_t)(c0w + 4 * lane) % (uint32_t)bst.C);
    real_t ev[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) { ev[o] = eps[cc]; cc = cc + 1 == bst.C ? 0 : cc + 1; }
    real_t q[12];
    tridiag_quad_quotients<MODE, NL>(L, Cc, R, ev, q);
    fd_band_emit_wave4<real_t, 3, true>(&bst, s_win[wave], jw, q);
}
// the fused step (k_f_tridiag_fused), four columns per lane
template <int MODE, bool NL, int NC, bool PIPE>
__global__ void __launch_bounds__(kBlock)
k_f_tridiag_fused4(const real_t *__restrict__ x, int64_t n, fd_band_store bst, int64_t jstart, FusedEps fz)
{
    constexpr int kWinBytes = (kBlock / 64) * FD_BAND_WAVE4_LDS(3) * (int)sizeof(real_t);
    constexpr int kLdsBytes = kWinBytes + 64 > (kFzMaxBlocks + 2 * kEpsGroups) * 8 ? kWinBytes + 64 : (kFzMaxBlocks + 2 * kEpsGroups) * 8;
    __shared__ __attribute__((aligned(16))) char s_lds[kLdsBytes];
    real_t (*s_win)[FD_BAND_WAVE4_LDS(3)] = reinterpret_cast<real_t (*)[FD_BAND_WAVE4_LDS(3)]>(s_lds);
    double *s_red = reinterpret_cast<double *>(s_lds);
    const int b = (int)blockIdx.x, nfin = fz.eg.C;
    if (b < nfin) { fused_finisher(fz, b, s_red); return; }
    if (b < nfin + fz.nblocks) { fused_eps_block<NC, PIPE>(x, n, fz, b - nfin, reinterpret_cast<double (*)[NC]>(s_red)); return; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = b - nfin - fz.nblocks;
    const int64_t nwaves = (bst.col_end - jstart + 255) / 256;
    const int64_t gstride = (int64_t)((int)gridDim.x - nfin - fz.nblocks) * (kBlock / 64);
    int64_t gw = (int64_t)cb * (kBlock / 64) + wave;
    int64_t jw = jstart + gw * 256;
    r4_t Cc = {0, 0, 0, 0}, L = {0, 0, 0, 0}, R = {0, 0, 0, 0};
    auto load_x = [&](int64_t jt, r4_t &c_, r4_t &l_, r4_t &r_) {
        const int64_t j = jt + 4 * lane;
        c_ = ld_quad_guarded(x, j, n); l_ = ld_quad_guarded(x, j - 4, n); r_ = ld_quad_guarded(x, j + 4, n);
        if (fz.xw && fz.nranks > 1) {
            if (j == fz.own_begin && fz.rank > 0) { l_.z = fused_halo(fz, 0, 0); l_.w = fused_halo(fz, 0, 1); }
            if (j + 4 == fz.own_end && fz.rank + 1 < fz.nranks) { r_.x = fused_halo(fz, 1, 0); r_.y = fused_halo(fz, 1, 1); }
        }
    };
    if (gw < nwaves) load_x(jw, Cc, L, R);
    real_t *s_eps = reinterpret_cast<real_t *>(s_lds + kWinBytes);
    if (wave == 0) fused_wait_eps(fz, cb, s_eps);
    __syncthreads();
    for (; gw < nwaves; gw += gstride) {
        const int c0w = (int)((jw + bst.shift) % bst.C);
        int c = (int)((uint32_t)(c0w + 4 * lane) % (uint32_t)bst.C);
        real_t ev[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) { ev[o] = s_eps[c]; c = c + 1 == bst.C ? 0 : c + 1; }
        real_t q[12];
        tridiag_quad_quotients<MODE, NL>(L, Cc, R, ev, q);
        fd_band_emit_wave4<real_t, 3, true>(&bst, s_win[wave], jw, q);
        jw = jstart + (gw + gstride) * 256;
        if (gw + gstride < nwaves) load_x(jw, Cc, L, R);
    }
    if (fz.trace && (blockIdx.x & 31) < 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); fz_mark_max(fz, 8); }
}
#endif

// (32-bit index arithmetic: the launcher checked that every entry / row / column number is below 2^31)
__device__ __forceinline__ int band_colptr32(int j, int l, int u, int M)
{
    const int w = l + u + 1;
    const int nt = j < u ? j : u;
    const int top = nt * u - nt * (nt - 1) / 2;
    const int b0 = M - l, f0 = b0 > 0 ? b0 : 0;
    int bot = 0;
    if (j > f0) { const int nn = j - f0, a = f0 - b0 + 1; bot = nn * a + nn * (nn - 1) / 2; }
    return w * j - top - bot;
}
// Rows are owned by workgroups (2 * kBlock each: every thread evaluates its row pair at the base and at every point of the
// batch, differences -> LDS [colour][row], dense 16-B LDS stores); then the workgroup walks the storage positions its rows
// occupy -- one contiguous range of nzval -- and every position finds its (column, row) by the band's arithmetic (one
// multiply-shift in the interior, colptr's closed form in the corner workgroups), reads its difference, divides, and is
// written with its neighbour as a dense 16-B pair; positions of that range whose row belongs to the next workgroup are
// left to it.
template <typename CT, int MODE, bool NL, int BS>
__global__ void __launch_bounds__(BS)
k_f_tridiag_lazy_store(const real_t *__restrict__ x, const CT *__restrict__ color, const real_t *__restrict__ eps, int c_lo, int B,
                       int n, int r0, int r1, real_t *__restrict__ outp, int M, int N, int eb, int cb, int ce, int l, int u, int C,
                       int shift, int pitch, uint64_t mw, uint64_t mc)
{
    extern __shared__ real_t s_val[];                          // [B][pitch] differences of the workgroup's rows, then [B] divisors
    real_t *s_ed = s_val + (size_t)B * pitch;
    const int R0 = r0 + (int)blockIdx.x * (2 * BS);
    if (R0 >= r1) return;
    const int R1 = R0 + 2 * BS < r1 ? R0 + 2 * BS : r1;
    const int w = l + u + 1, top_full = u * (u + 1) / 2;
    if ((int)threadIdx.x < B) {
        const real_t e = eps[c_lo + threadIdx.x];
        s_ed[threadIdx.x] = MODE == 1 ? 2 * e : e;
    }
    // ---- phase A
    const int i = R0 + 2 * (int)threadIdx.x;
    if (i < R1) {
        real_t xv[4];
        int cv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = i - 1 + k;
            const bool in = (j >= 0) & (j < n);
            xv[k] = in ? x[in ? j : 0] : 0.0;
            const int c = in ? (int)color[in ? j : 0] : -1;
            cv[k] = (c == (int)(CT)(-1) || c < 0) ? -1 : c - c_lo;
        }
        real_t b0 = 0.0, b1 = 0.0;
        if (MODE == 0) {
            b0 = tridiag_row<real_t, NL>(xv[0], xv[1], xv[2]);
            b1 = tridiag_row<real_t, NL>(xv[1], xv[2], xv[3]);
        }
        for (int b = 0; b < B; ++b) {
            const real_t e = eps[c_lo + b];
            real_t p[4], q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const real_t d = (cv[k] == b) ? e : 0.0; p[k] = xv[k] + d; q[k] = xv[k] - d; }
            const real_t s0 = MODE == 1 ? tridiag_row<real_t, NL>(q[0], q[1], q[2]) : b0;
            const real_t s1 = MODE == 1 ? tridiag_row<real_t, NL>(q[1], q[2], q[3]) : b1;
            *reinterpret_cast<r2_t *>(s_val + b * pitch + 2 * (int)threadIdx.x) =
                r2_t{sub_exact(tridiag_row<real_t, NL>(p[0], p[1], p[2]), s0), sub_exact(tridiag_row<real_t, NL>(p[1], p[2], p[3]), s1)};
        }
    }
    __syncthreads();
    // ---- phase B: the storage positions of rows [R0, R1) (local to the column range [cb, ce))
    const int jlo = R0 - l > cb ? R0 - l : cb, jhi = R1 - 1 + u < ce - 1 ? R1 - 1 + u : ce - 1;     // columns that touch those rows
    if (jhi < jlo) return;
    const int plo = (band_colptr32(jlo, l, u, M) - eb) & ~1, phi = band_colptr32(jhi + 1, l, u, M) - eb;   // [plo, phi)
    const bool interior = jlo >= u && jhi + l <= M - 1;        // no column cut off by the matrix edge
    const bool vec = ((((uintptr_t)outp) & kPairMask) == 0);
    auto column_of = [&](int g) -> int {
        int j = (int)fd_div31((uint32_t)(g + top_full), mw);
        if (j >= N) j = N - 1;
        while (j > 0 && band_colptr32(j, l, u, M) > g) --j;
        while (j + 1 < N && band_colptr32(j + 1, l, u, M) <= g) ++j;
        return j;
    };
    for (int pp = plo + 2 * (int)threadIdx.x; pp < phi; pp += 2 * BS) {
        real_t qv[2];
        bool wr[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int g = pp + h + eb;
            int j, row;
            if (interior) {
                const uint32_t Q = (uint32_t)(g + top_full);
                const uint32_t jq = fd_div31(Q, mw);
                j = (int)jq;
                row = j - u + (int)(Q - jq * (uint32_t)w);
            } else {
                const int gg = g < phi + eb ? g : phi + eb - 1;
                j = column_of(gg);
                row = (j - u > 0 ? j - u : 0) + (gg - band_colptr32(j, l, u, M));
            }
            const uint32_t cj = (uint32_t)(j + shift);
            const int b = (int)(cj - fd_div31(cj, mc) * (uint32_t)C) - c_lo;
            const bool live = pp + h >= 0 && pp + h < phi && row >= R0 && row < R1 && b >= 0 && b < B && j >= cb && j < ce;
            const int at = live ? b * pitch + (row - R0) : 0;
            qv[h] = s_val[at] / s_ed[live ? b : 0];
            wr[h] = live;
        }
        if (wr[0] & wr[1] & vec) *reinterpret_cast<r2_t *>(outp + pp) = r2_t{qv[0], qv[1]};
        else {
            if (wr[0]) outp[pp] = qv[0];
            if (wr[1]) outp[pp + 1] = qv[1];
        }
    }
}

// 5-point stencils on an nx (fast) x ny grid.  CLAMP = false: zero-Dirichlet Laplacian
// w + e + s + n - 4x ; CLAMP = true: the reference's clamped-edge sum x + x[i-1] + x[i+1] + x[j-1] + x[j+1].
// (SK: 0 = Laplacian, 1 = clamped sum, 2 = Laplacian + x[k]^2 * x[k+1]: a nonlinear variant whose J depends on x.)
template <typename T, int SK>
__global__ void __launch_bounds__(kBlock)
k_f_stencil5(T *__restrict__ fx, const T *__restrict__ x, int64_t nx, int64_t ny, int64_t xs, int64_t fs,
             int64_t r0, int64_t r1)
{
    const T *xb = x + (int64_t)blockIdx.y * xs;
    T *fb = fx + (int64_t)blockIdx.y * fs;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = r0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; k < r1; k += stride) {
        const int64_t j = k / nx, i = k - j * nx;
        if (SK == 1) {
            const int64_t im = i > 0 ? i - 1 : 0, ip = i + 1 < nx ? i + 1 : nx - 1;
            const int64_t jm = j > 0 ? j - 1 : 0, jp = j + 1 < ny ? j + 1 : ny - 1;
            fb[k] = (((xb[k] + xb[im + nx * j]) + xb[ip + nx * j]) + xb[i + nx * jm]) + xb[i + nx * jp];
        } else {
            const T w = i > 0 ? xb[k - 1] : zero_of<T>();
            const T e = i + 1 < nx ? xb[k + 1] : zero_of<T>();
            const T s = j > 0 ? xb[k - nx] : zero_of<T>();
            const T n = j + 1 < ny ? xb[k + nx] : zero_of<T>();
            T v = (((w + e) + s) + n) - kFour * xb[k];
            if (SK == 2) v = v + (xb[k] * xb[k]) * e;    // lap5_nl: + x[k]^2 * x[k+1] (J depends on x)
            fb[k] = v;
        }
    }
}

// Same stencils, two rows per thread (nx even): 16-B loads of the centre / south / north pairs, two
// scalar loads for the west / east neighbours, one 16-B store.  One-shot launch with the XCD-aware
// tile mapping: the rows k-nx, k, k+nx that share x lines are evaluated by the same XCD, so each
// line of x enters one L2 instead of three.
template <int SK>
__global__ void __launch_bounds__(kBlock)
k_f_stencil5_v2(real_t *__restrict__ fx, const real_t *__restrict__ x, int64_t nx, int64_t ny, int64_t xs, int64_t fs,
                int64_t r0, int64_t r1)
{
    const real_t *xb = x + (int64_t)blockIdx.y * xs;
    real_t *fb = fx + (int64_t)blockIdx.y * fs;
    const int64_t ntiles = (r1 - r0 + 2 * kBlock - 1) / (2 * kBlock);
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int64_t k = r0 + tile * (2 * kBlock) + threadIdx.x * 2;   // r0 even, nx even => k, k+1 share a grid row
    if (k >= r1) return;
    const int64_t j = k / nx, i = k - j * nx;
    const r2_t c = *reinterpret_cast<const r2_t *>(xb + k);
    const bool hs = j > 0, hn = j + 1 < ny, hw = i > 0, he = i + 2 < nx;
    r2_t s = r2_t{0.0, 0.0}, n = r2_t{0.0, 0.0};
    if (hs) s = *reinterpret_cast<const r2_t *>(xb + k - nx);
    if (hn) n = *reinterpret_cast<const r2_t *>(xb + k + nx);
    const real_t w = hw ? xb[k - 1] : 0.0;
    const real_t e = he ? xb[k + 2] : 0.0;
    real_t v0, v1;
    if (SK == 1) {
        const real_t w0 = hw ? w : c.x, e1 = he ? e : c.y;
        const real_t s0 = hs ? s.x : c.x, s1 = hs ? s.y : c.y, n0 = hn ? n.x : c.x, n1 = hn ? n.y : c.y;
        v0 = (((c.x + w0) + c.y) + s0) + n0;
        v1 = (((c.y + c.x) + e1) + s1) + n1;
    } else {
        v0 = (((w + c.y) + s.x) + n.x) - kFour * c.x;
        v1 = (((c.x + e) + s.y) + n.y) - kFour * c.y;
        if (SK == 2) { v0 = v0 + (c.x * c.x) * c.y; v1 = v1 + (c.y * c.y) * e; }
    }
    *reinterpret_cast<r2_t *>(fb + k) = r2_t{v0, v1};
}

// Lazy-point version of the 5-point stencils (see k_f_tridiag_lazy): the 8 base values a pair of rows
// depends on are loaded once, every point of the batch is evaluated from registers.
// v[8] = {c0, c1, s0, s1, n0, n1, w, e} = x at {k, k+1, k-nx, k-nx+1, k+nx, k+nx+1, k-1, k+2}.
template <typename T, int SK>
__device__ __forceinline__ void stencil5_pair(const T *v, bool hs, bool hn, bool hw, bool he, T &o0, T &o1)
{
    if (SK == 1) {
        const T w0 = hw ? v[6] : v[0], e1 = he ? v[7] : v[1];
        const T s0 = hs ? v[2] : v[0], s1 = hs ? v[3] : v[1], n0 = hn ? v[4] : v[0], n1 = hn ? v[5] : v[1];
        o0 = (((v[0] + w0) + v[1]) + s0) + n0;
        o1 = (((v[1] + v[0]) + e1) + s1) + n1;
    } else {
        const T z = zero_of<T>();
        const T w = hw ? v[6] : z, e = he ? v[7] : z;
        const T s0 = hs ? v[2] : z, s1 = hs ? v[3] : z, n0 = hn ? v[4] : z, n1 = hn ? v[5] : z;
        o0 = (((w + v[1]) + s0) + n0) - kFour * v[0];
        o1 = (((v[0] + e) + s1) + n1) - kFour * v[1];
        if (SK == 2) { o0 = o0 + (v[0] * v[0]) * v[1]; o1 = o1 + (v[1] * v[1]) * e; }
    }
}

template <typename CT, int MODE, int SK>
__global__ void __launch_bounds__(kBlock)
k_f_stencil5_lazy(real_t *__restrict__ fx, int64_t fs, real_t *__restrict__ base_out, const real_t *__restrict__ x,
                  const CT *__restrict__ color, const real_t *__restrict__ eps, int c_lo, int B, int64_t nx, int64_t ny,
                  int64_t r0, int64_t r1, int imag_only, int diff)
{
    const int64_t ntiles = (r1 - r0 + 2 * kBlock - 1) / (2 * kBlock);
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int64_t k = r0 + tile * (2 * kBlock) + threadIdx.x * 2;
    if (k >= r1) return;
    const int64_t j = k / nx, i = k - j * nx;
    const bool hs = j > 0, hn = j + 1 < ny, hw = i > 0, he = i + 2 < nx;
    const int64_t idx[8] = {k, k + 1, k - nx, k - nx + 1, k + nx, k + nx + 1, k - 1, k + 2};
    const bool ok[8] = {true, true, hs, hs, hn, hn, hw, he};
    real_t xv[8];
    int cv[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int64_t at = ok[m] ? idx[m] : k;
        xv[m] = x[at];
        const int c = (int)color[at];
        cv[m] = (!ok[m] || c == (int)(CT)(-1) || c < 0) ? -1 : c - c_lo;
    }
    real_t b0 = 0.0, b1 = 0.0;                        // f(x) at rows k, k+1 (written to base_out, or the subtrahend of diff)
    if (base_out || (diff && MODE == 0)) stencil5_pair<real_t, SK>(xv, hs, hn, hw, he, b0, b1);
    if (base_out) *reinterpret_cast<r2_t *>(base_out + k) = r2_t{b0, b1};
    for (int b = 0; b < B; ++b) {
        const real_t e = eps[c_lo + b];
        real_t d[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) d[m] = (cv[m] == b) ? e : 0.0;
        if (MODE == 2) {
            cd p[8], o0, o1;
#pragma unroll
            for (int m = 0; m < 8; ++m) p[m] = cd{xv[m], d[m]};
            stencil5_pair<cd, SK>(p, hs, hn, hw, he, o0, o1);
            if (imag_only) {
                *reinterpret_cast<r2_t *>(fx + (int64_t)b * fs + k) = r2_t{o0.im, o1.im};
            } else {
                real_t *dst = fx + ((int64_t)b * fs + k) * 2;
                *reinterpret_cast<r2_t *>(dst) = r2_t{o0.re, o0.im};
                *reinterpret_cast<r2_t *>(dst + 2) = r2_t{o1.re, o1.im};
            }
        } else if (diff) {
            // differences (fd_lazy_points.diff): f(x + d) - f(x), or f(x + d) - f(x - d), one array per colour
            real_t p[8], q[8], o0, o1, s0 = b0, s1 = b1;
#pragma unroll
            for (int m = 0; m < 8; ++m) { p[m] = xv[m] + d[m]; q[m] = xv[m] - d[m]; }
            stencil5_pair<real_t, SK>(p, hs, hn, hw, he, o0, o1);
            if (MODE == 1) stencil5_pair<real_t, SK>(q, hs, hn, hw, he, s0, s1);
            *reinterpret_cast<r2_t *>(fx + (int64_t)b * fs + k) = r2_t{sub_exact(o0, s0), sub_exact(o1, s1)};
        } else {
#pragma unroll
            for (int sgn = 0; sgn < (MODE == 1 ? 2 : 1); ++sgn) {
                real_t p[8], o0, o1;
#pragma unroll
                for (int m = 0; m < 8; ++m) p[m] = sgn == 0 ? xv[m] + d[m] : xv[m] - d[m];
                stencil5_pair<real_t, SK>(p, hs, hn, hw, he, o0, o1);
                *reinterpret_cast<r2_t *>(fx + (int64_t)(sgn * B + b) * fs + k) = r2_t{o0, o1};
            }
        }
    }
}

// block-coupled: sig_b = sum_j w_j x_b[j], w_j = (j+1)/bs; one wave per block, fixed-order tree.
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_f_block_sigma(T *__restrict__ sig, const T *__restrict__ x, int64_t nb, int64_t bs, int64_t xs, int64_t b0, int64_t b1)
{
    const T *xb = x + (int64_t)blockIdx.y * xs;
    T *sb = sig + (int64_t)blockIdx.y * nb;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * kBlock) >> 6;
    for (int64_t b = b0 + wave; b < b1; b += nw) {
        T acc = zero_of<T>();
        for (int64_t j = lane; j < bs; j += 64) acc = acc + ((real_t)(j + 1) / (real_t)bs) * xb[b * bs + j];
        // serial-order-independent but fixed: tree over lanes
        for (int off = 32; off > 0; off >>= 1) {
            if constexpr (sizeof(T) == sizeof(real_t)) {
                real_t o = __shfl_down(*reinterpret_cast<real_t *>(&acc), off, 64);
                acc = acc + *reinterpret_cast<T *>(&o);
            } else {
                cd *a = reinterpret_cast<cd *>(&acc);
                cd o{__shfl_down(a->re, off, 64), __shfl_down(a->im, off, 64)};
                acc = acc + *reinterpret_cast<T *>(&o);
            }
        }
        if (lane == 0) sb[b] = acc;
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_f_block_apply(T *__restrict__ fx, const T *__restrict__ x, const T *__restrict__ sig, int64_t nb, int64_t bs,
                int64_t xs, int64_t fs, int64_t r0, int64_t r1)
{
    const T *xb = x + (int64_t)blockIdx.y * xs;
    const T *sb = sig + (int64_t)blockIdx.y * nb;
    T *fb = fx + (int64_t)blockIdx.y * fs;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = r0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; k < r1; k += stride) {
        const int64_t b = k / bs;
        const T sm = b > 0 ? sb[b - 1] : zero_of<T>();
        const T sp = b + 1 < nb ? sb[b + 1] : zero_of<T>();
        const T S = (sm + sb[b]) + sp;
        fb[k] = xb[k] * S + sin_of(xb[k]);
    }
}

// y[k] = (x1-3)^2 + x1*x2 + (x2+4)^2 - 3 with x1 = x[k], x2 = x[n+k]
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_f_nonsquare(T *__restrict__ fx, const T *__restrict__ x, int64_t n, int64_t xs, int64_t fs, int64_t r0, int64_t r1)
{
    const T *xb = x + (int64_t)blockIdx.y * xs;
    T *fb = fx + (int64_t)blockIdx.y * fs;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = r0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; k < r1; k += stride) {
        const T a = xb[k], b = xb[n + k];
        fb[k] = ((((a - 3.0) * (a - 3.0)) + a * b) + ((b + 4.0) * (b + 4.0))) - 3.0;
    }
}

struct BuiltinF {
    uint32_t magic = 0xFD0F00D5u;
    fd_ctx *ctx = nullptr;
    int family = 0;
    int64_t prm[3] = {0, 0, 0};
    int64_t M = 0, N = 0;
    int32_t *d_srow = nullptr, *d_scol = nullptr;   // FD_F_SPARSE: the pattern by rows (rowptr[M + 1], ascending columns), device
    int32_t *d_sdest = nullptr;                     //   per entry of that list: its slot in the CSC order it was built from (k_f_sparse_store_rows)
    std::vector<int32_t> h_colptr;                  //   that CSC pattern's column offsets (0-based), host
    struct RowsMemo {                               //   what the launcher knows about a plan (fd_csc_store.plan_serial): is its pattern mine?
        int verdict = 0;
        bool pending = false;
        hipEvent_t ev = nullptr;
        unsigned long long *h_note = nullptr;       //   pinned
    };
    std::atomic<int64_t> row_stores{0};
    std::unordered_map<unsigned long long, RowsMemo> rows_memo;
    std::mutex rows_mutex;
    std::atomic<int64_t> launches{0}, points{0};
    void *d_sig = nullptr;  // block-coupled sigma scratch
    int64_t sig_cap = 0;    // in (re,im)-capable elements
};

int balanced_grid(int64_t tiles, int64_t cap);

// grid.x covers `rows` work items of one point, grid.y = points; whole rounds (see balanced_grid)
static inline dim3 grid2(int64_t rows, int64_t nbatch, int num_cus)
{
    // one work item per thread, uncapped (N = 10^7 tridiagonal f!: 65 -> 59 us per launch against a capped grid, profiles/r04_b_*)
    (void)num_cus;
    const int64_t tiles = (rows + kBlock - 1) / kBlock;
    return dim3((unsigned)balanced_grid(tiles, (int64_t)1 << 30), (unsigned)nbatch, 1);
}

#include "fdjac_functor_f.hip"

template <typename T>
static int launch_family(BuiltinF *b, void *fx, const void *x, int64_t nbatch, int64_t xs, int64_t fs, int64_t r0,
                         int64_t r1, hipStream_t s)
{
    if (r1 <= r0) return 0;
    if (b->family == FD_F_LAP7 || b->family == FD_F_SPARSE) return functor_family_launch<T>(b, fx, x, nbatch, xs, fs, r0, r1, s);
    const int ncu = b->ctx->num_cus;
    T *fxp = (T *)fx;
    const T *xp = (const T *)x;
    const dim3 g = grid2(r1 - r0, nbatch, ncu);
    switch (b->family) {
    case FD_F_TRIDIAG:
    case FD_F_TRIDIAG_NL: {
        const bool nl = b->family == FD_F_TRIDIAG_NL;
        if constexpr (sizeof(T) == sizeof(real_t)) {
            const bool aligned = ((((uintptr_t)fx) | ((uintptr_t)x)) & kPairMask) == 0 && (xs % 2 == 0 || nbatch == 1) &&
                                 (fs % 2 == 0 || nbatch == 1);
            if (aligned) {
                const int64_t r0e = r0 & ~(int64_t)1;
                const dim3 g2 = grid2((r1 - r0e + 1) / 2, nbatch, ncu);
                if (nl)
                    hipLaunchKernelGGL((k_f_tridiag_v2<true>), g2, dim3(kBlock), 0, s, (real_t *)fx, (const real_t *)x,
                                       b->prm[0], xs, fs, r0e, r1);
                else
                    hipLaunchKernelGGL((k_f_tridiag_v2<false>), g2, dim3(kBlock), 0, s, (real_t *)fx, (const real_t *)x,
                                       b->prm[0], xs, fs, r0e, r1);
                break;
            }
        }
        if (nl)
            hipLaunchKernelGGL((k_f_tridiag<T, true>), g, dim3(kBlock), 0, s, fxp, xp, b->prm[0], xs, fs, r0, r1);
        else
            hipLaunchKernelGGL((k_f_tridiag<T, false>), g, dim3(kBlock), 0, s, fxp, xp, b->prm[0], xs, fs, r0, r1);
        break;
    }
    case FD_F_LAP5:
    case FD_F_LAP5_NL:
    case FD_F_CLAMP5: {
        const int sk = b->family == FD_F_CLAMP5 ? 1 : b->family == FD_F_LAP5_NL ? 2 : 0;
        if constexpr (sizeof(T) == sizeof(real_t)) {
            const bool ok = ((((uintptr_t)fx) | ((uintptr_t)x)) & kPairMask) == 0 && (xs % 2 == 0 || nbatch == 1) &&
                            (fs % 2 == 0 || nbatch == 1) && (b->prm[0] % 2 == 0);
            if (ok) {
                const int64_t r0e = r0 & ~(int64_t)1;
                const int64_t ntiles = (r1 - r0e + 2 * kBlock - 1) / (2 * kBlock);
                const dim3 g2((unsigned)(8 * xcd_chunks(ntiles)), (unsigned)nbatch, 1);
#define FD_ST_V2(SKK) hipLaunchKernelGGL((k_f_stencil5_v2<SKK>), g2, dim3(kBlock), 0, s, (real_t *)fx, (const real_t *)x, \
                                       b->prm[0], b->prm[1], xs, fs, r0e, r1)
                if (sk == 1) FD_ST_V2(1); else if (sk == 2) FD_ST_V2(2); else FD_ST_V2(0);
#undef FD_ST_V2
                break;
            }
        }
#define FD_ST(SKK) hipLaunchKernelGGL((k_f_stencil5<T, SKK>), g, dim3(kBlock), 0, s, fxp, xp, b->prm[0], b->prm[1], xs, fs, r0, r1)
        if (sk == 1) FD_ST(1); else if (sk == 2) FD_ST(2); else FD_ST(0);
#undef FD_ST
        break;
    }
    case FD_F_BLOCKCOUPLED: {
        const int64_t nb = b->prm[0], bs = b->prm[1];
        if (b->sig_cap < nbatch * nb) {
            if (b->d_sig) (void)hipFree(b->d_sig);
            b->d_sig = nullptr;
            if (hipMalloc(&b->d_sig, (size_t)(nbatch * nb) * 2 * sizeof(real_t)) != hipSuccess) return 2;
            b->sig_cap = nbatch * nb;
        }
        const int64_t b0 = std::max<int64_t>(r0 / bs - 1, 0), b1 = std::min<int64_t>((r1 + bs - 1) / bs + 1, nb);
        const dim3 gs = grid2((b1 - b0) * 64, nbatch, ncu);
        hipLaunchKernelGGL((k_f_block_sigma<T>), gs, dim3(kBlock), 0, s, (T *)b->d_sig, xp, nb, bs, xs, b0, b1);
        hipLaunchKernelGGL((k_f_block_apply<T>), g, dim3(kBlock), 0, s, fxp, xp, (const T *)b->d_sig, nb, bs, xs, fs, r0, r1);
        break;
    }
    case FD_F_NONSQUARE:
        hipLaunchKernelGGL((k_f_nonsquare<T>), g, dim3(kBlock), 0, s, fxp, xp, b->prm[0], xs, fs, r0, r1);
        break;
    default: return 3;
    }
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

static int builtin_launch(void *fctx, void *fx, const void *x, int64_t nbatch, int64_t x_stride, int64_t fx_stride,
                          int64_t row_begin, int64_t row_end, int is_complex, void *stream)
{
    BuiltinF *b = (BuiltinF *)fctx;
    if (!b || b->magic != 0xFD0F00D5u) return 1;
    if (nbatch <= 0) return 0;
    if (nbatch > 65535) return 5;
    b->launches.fetch_add(1);
    b->points.fetch_add(nbatch);
    const int64_t r0 = std::max<int64_t>(row_begin, 0), r1 = std::min<int64_t>(row_end, b->M);
    if (is_complex)
        return launch_family<cd>(b, fx, x, nbatch, x_stride, fx_stride, r0, r1, (hipStream_t)stream);
    return launch_family<real_t>(b, fx, x, nbatch, x_stride, fx_stride, r0, r1, (hipStream_t)stream);
}

// storing workgroups of a fused launch: as many as tiles, but never more than fit on the device next to the reduction's (what the
// kernel's registers and LDS allow per CU, asked of the runtime once per kernel): a workgroup dispatched only after others have left
// would start its life after the step sizes are out -- a second round of memory latency at the end of the launch
template <typename K>
static unsigned fused_grid(K kernel, unsigned tiles_wg, int prefix)
{
    static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    static std::mutex mu;
    static std::unordered_map<const void *, int> per_cu;
    int occ = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = per_cu.find((const void *)kernel);
        if (it == per_cu.end()) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlock, 0) != hipSuccess || nb < 1) nb = 4;
            (void)hipGetLastError();
            it = per_cu.emplace((const void *)kernel, nb).first;
        }
        occ = it->second;
    }
    const int64_t room = (int64_t)cus * occ - prefix;
    const int64_t cap = room > cus ? room : cus;
    if ((int64_t)tiles_wg <= cap) return (unsigned)prefix + tiles_wg;
    const int64_t rounds = ((int64_t)tiles_wg + cap - 1) / cap;
    return (unsigned)prefix + (unsigned)(((int64_t)tiles_wg + rounds - 1) / rounds);
}

template <typename CT>
static int lazy_tridiag_launch(BuiltinF *b, void *fx, const fd_lazy_points *lp, int64_t fs, int64_t r0, int64_t r1,
                               hipStream_t s)
{
    const int64_t r0e = r0 & ~(int64_t)1;
    // one pair of rows per thread, uncapped grid (measured faster than a capped grid-stride launch here)
    int64_t g = ((r1 - r0e + 1) / 2 + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    const bool nl = b->family == FD_F_TRIDIAG_NL;
    const int mode = lp->is_complex ? 2 : (lp->pts == 2 ? 1 : 0);
#define FD_LAZY(MODE, NL)                                                                                           \
    hipLaunchKernelGGL((k_f_tridiag_lazy<CT, MODE, NL>), dim3((unsigned)g), dim3(kBlock), 0, s, (real_t *)fx, fs,    \
                       (real_t *)lp->base_out, (const real_t *)lp->x, (const CT *)lp->color, (const real_t *)lp->eps, lp->c_lo,     \
                       lp->ncolors, b->prm[0], r0e, r1, lp->imag_only, mode != 2 ? lp->diff : 0)
    if (lp->store) {
        // the launch stores the Jacobian itself (include/fdjac_device.h); one-shot launches
        if (lp->store_kind != FD_STORE_BAND) return FD_LAZY_DECLINED;
        const fd_band_store bst = *(const fd_band_store *)lp->store;
        if (bst.elem_bytes != (int)sizeof(real_t) || mode == 2) return FD_LAZY_DECLINED;
        if (bst.N != b->prm[0] || bst.M != b->prm[0]) return FD_LAZY_DECLINED;   // a plan of another problem size than this fixture's
        const int wband = bst.l + bst.u + 1;
        // an exactly tridiagonal band with every colour in this batch: the column-centric wave kernel, any layout; colour chunks,
        // colour ownership and wider bands take the row-owned kernel below
        const bool wave_ok = bst.l == 1 && bst.u == 1 && bst.M == bst.N && lp->c_lo == 0 &&
                             lp->ncolors == bst.C && bst.C >= 3 && (((uintptr_t)lp->x) & kPairMask) == 0 && bst.col_end > bst.col_begin;
#ifdef FDJAC_F32
        if (wave_ok && (((uintptr_t)lp->x) & 15) == 0) {      // Float32: four columns per lane
            const int64_t jstart = bst.col_begin & ~(int64_t)3;
            const int64_t nwaves = (bst.col_end - jstart + 255) / 256;
            const unsigned gw = (unsigned)((nwaves + kBlock / 64 - 1) / (kBlock / 64));
            if (lp->eps_job) {
                const FusedEps fz = *(const FusedEps *)lp->eps_job;
#define FD_LAZY_FZ4P(MODE, NL, NCC, PP) hipLaunchKernelGGL((k_f_tridiag_fused4<MODE, NL, NCC, PP>), dim3(fused_grid(k_f_tridiag_fused4<MODE, NL, NCC, PP>, gw, fz.eg.C + fz.nblocks)), dim3(kBlock), 0, s, (const real_t *)lp->x, b->prm[0], bst, jstart, fz)
#define FD_LAZY_FZ4(MODE, NL)                                                                                                              \
                do { if (fz.eg.C <= 4) { if (fz.eg.tpb > 2) FD_LAZY_FZ4P(MODE, NL, 4, true); else FD_LAZY_FZ4P(MODE, NL, 4, false); }               \
                     else { if (fz.eg.tpb > 2) FD_LAZY_FZ4P(MODE, NL, kRegColors, true); else FD_LAZY_FZ4P(MODE, NL, kRegColors, false); } } while (0)
                if (mode == 0) { if (nl) FD_LAZY_FZ4(0, true); else FD_LAZY_FZ4(0, false); }
                else { if (nl) FD_LAZY_FZ4(1, true); else FD_LAZY_FZ4(1, false); }
#undef FD_LAZY_FZ4
#undef FD_LAZY_FZ4P
                return hipGetLastError() == hipSuccess ? 0 : 4;
            }
#define FD_LAZY_SW4(MODE, NL)                                                                                      \
            hipLaunchKernelGGL((k_f_tridiag_store_wave4<MODE, NL>), dim3(gw), dim3(kBlock), 0, s, (const real_t *)lp->x, \
                               (const real_t *)lp->eps, b->prm[0], bst, jstart)
            if (mode == 0) { if (nl) FD_LAZY_SW4(0, true); else FD_LAZY_SW4(0, false); }
            else { if (nl) FD_LAZY_SW4(1, true); else FD_LAZY_SW4(1, false); }
#undef FD_LAZY_SW4
            return hipGetLastError() == hipSuccess ? 0 : 4;
        }
#endif
        if (wave_ok) {
            const int64_t jstart = bst.col_begin & ~(int64_t)1;
            const int64_t nwaves = (bst.col_end - jstart + 127) / 128;
            const unsigned gw = (unsigned)((nwaves + kBlock / 64 - 1) / (kBlock / 64));
            if (lp->eps_job) {
                // the fused step: this launch also runs the step-size reduction (finisher + reduction workgroups first, see k_f_tridiag_fused)
                const FusedEps fz = *(const FusedEps *)lp->eps_job;
#define FD_LAZY_FZP(MODE, NL, NCC, PP) hipLaunchKernelGGL((k_f_tridiag_fused<MODE, NL, NCC, PP>), dim3(fused_grid(k_f_tridiag_fused<MODE, NL, NCC, PP>, gw, fz.eg.C + fz.nblocks)), dim3(kBlock), 0, s, (const real_t *)lp->x, b->prm[0], bst, jstart, fz)
#define FD_LAZY_FZ(MODE, NL)                                                                                                              \
                do { if (fz.eg.C <= 4) { if (fz.eg.tpb > 2) FD_LAZY_FZP(MODE, NL, 4, true); else FD_LAZY_FZP(MODE, NL, 4, false); }               \
                     else { if (fz.eg.tpb > 2) FD_LAZY_FZP(MODE, NL, kRegColors, true); else FD_LAZY_FZP(MODE, NL, kRegColors, false); } } while (0)
                if (mode == 0) { if (nl) FD_LAZY_FZ(0, true); else FD_LAZY_FZ(0, false); }
                else { if (nl) FD_LAZY_FZ(1, true); else FD_LAZY_FZ(1, false); }
#undef FD_LAZY_FZ
#undef FD_LAZY_FZP
                return hipGetLastError() == hipSuccess ? 0 : 4;
            }
#define FD_LAZY_SW(MODE, NL)                                                                                       \
            hipLaunchKernelGGL((k_f_tridiag_store_wave<MODE, NL>), dim3(gw), dim3(kBlock), 0, s, (const real_t *)lp->x,  \
                               (const real_t *)lp->eps, b->prm[0], bst, jstart)
            if (mode == 0) { if (nl) FD_LAZY_SW(0, true); else FD_LAZY_SW(0, false); }
            else { if (nl) FD_LAZY_SW(1, true); else FD_LAZY_SW(1, false); }
#undef FD_LAZY_SW
            return hipGetLastError() == hipSuccess ? 0 : 4;
        }
        if (lp->eps_job) return FD_LAZY_DECLINED;      // (the fused step exists for the wave kernels only: the library runs the reduction itself)
        if (bst.layout != FD_BAND_CSC) return FD_LAZY_DECLINED;
        const int bs = kBlock;      // (one-wave workgroups, BS = 64, measured slower: 111-119 vs 105 us at N = 10^7)
        const int pitch = 2 * bs + 2;
        const size_t shm = sizeof(real_t) * ((size_t)lp->ncolors * (size_t)pitch + (size_t)lp->ncolors + 2);
        const int64_t nnz_all = fd_band_colptr(&bst, bst.N);
        if (shm > (size_t)60 * 1024 || nnz_all + (int64_t)wband * wband + 64 >= ((int64_t)1 << 31) || bst.N + bst.C + 64 >= ((int64_t)1 << 31) ||
            bst.M + wband + 64 >= ((int64_t)1 << 31) || lp->ncolors > kBlock) return FD_LAZY_DECLINED;
        const unsigned gs = (unsigned)((r1 - r0e + 2 * bs - 1) / (2 * bs));
#define FD_LAZY_ST(MODE, NL)                                                                                       \
        hipLaunchKernelGGL((k_f_tridiag_lazy_store<CT, MODE, NL, kBlock>), dim3(gs), dim3(kBlock), shm, s, (const real_t *)lp->x, \
                           (const CT *)lp->color, (const real_t *)lp->eps, lp->c_lo, lp->ncolors, (int)b->prm[0], (int)r0e, (int)r1, \
                           (real_t *)bst.out, (int)bst.M, (int)bst.N, (int)bst.entry_begin, (int)bst.col_begin, (int)bst.col_end,     \
                           bst.l, bst.u, bst.C, bst.shift, pitch, fd_magic31((uint32_t)wband), fd_magic31((uint32_t)bst.C))
        if (mode == 0) { if (nl) FD_LAZY_ST(0, true); else FD_LAZY_ST(0, false); }
        else { if (nl) FD_LAZY_ST(1, true); else FD_LAZY_ST(1, false); }
#undef FD_LAZY_ST
        return hipGetLastError() == hipSuccess ? 0 : 4;
    }
    if (mode == 0) { if (nl) FD_LAZY(0, true); else FD_LAZY(0, false); }
    else if (mode == 1) { if (nl) FD_LAZY(1, true); else FD_LAZY(1, false); }
    else { if (nl) FD_LAZY(2, true); else FD_LAZY(2, false); }
#undef FD_LAZY
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// fd_lazy_points.store with a fd_stencil5_store (include/fdjac_device.h): the 5-point fixtures evaluated at the lazily perturbed
// points, difference quotients formed and stored into the CSC nzval of the stencil by this launch -- nothing follows it.
// COLUMN-centric like k_f_tridiag_store_wave: lane t of a wavefront owns the grid columns (i, j), (i + 1, j), i = i0 + 2t,
// loads the 6 x 5 window of x around them (11 aligned 16-B loads; the vertical re-reads come from L2: consecutive grid rows run
// on one XCD), and evaluates the five rows each column touches at x +- eps e_k.  Seen from one of those rows the colour's point
// x +- eps_c mask_c differs from x in column k only (the plan verified that colorvec is a valid colouring of the exact stencil),
// so every operand -- the "+ 0.0" of the unperturbed coordinates included -- and every operation (stencil5 row, sub_exact,
// IEEE division) is that of the colour-batched evaluation: same bits (scripts/ubench/stencil_store_probe.hip checks the form
// against the reference's loop order on the host).  The wavefront's 128 x 5 quotients leave through fd_stencil5_emit_wave.
template <typename T, int SK> __device__ __forceinline__ T stencil5_row(T c, T w, T e, T s, T n)
{
    T v = (((w + e) + s) + n) - kFour * c;
    if (SK == 2) v = v + (c * c) * e;
    return v;
}

// a / b for a divisor shared by several quotients, y = 1 / b computed once: one multiplication and two FMA correction steps give
// the CORRECTLY ROUNDED quotient (Markstein: q1 is a faithful rounding of a / b, so q2 = RN(a / b) when nothing over- or
// underflows; zero, tiny, huge and non-finite operands take the true division) -- the bits of IEEE a / b, about half its
// instructions (scripts/ubench/exact_div_probe.hip: 5e10 random and next-to-tie operand pairs, 0 mismatches).  Float64 only.
// `bok` = div_shared_ok(b), tested ONCE per divisor: with |b| in [2^-100, 2^100] and |a| in [2^-800, 2^800] the first quotient lies in
// [2^-900, 2^900] by itself -- two comparisons per quotient instead of four.  A/B on one box (scripts/ab_c5.sh, ab_c3.sh): the
// block-coupled kernel 49.9 -> 48.5 us (used there); the 5-point kernel 98 -> 100 us (not used there).
__device__ __forceinline__ bool div_shared_ok(real_t b) { const real_t mb = fabs(b); return mb >= (real_t)0x1p-100 && mb <= (real_t)0x1p100; }
// b = +-2^k with 2^k and 2^-k normal numbers: then a * (1 / b) IS a / b for every a -- one rounding of the same real number either way
// (scaling by a power of two is exact until the result leaves the normal range, and there both forms round the same value once).
// The complex step's divisor is eps(T) = 2^-52 / 2^-23 unless the caller chose another one.
__device__ __forceinline__ bool div_is_pow2(real_t b)
{
    if constexpr (sizeof(real_t) == 8) {
        const unsigned long long u = (unsigned long long)__double_as_longlong((double)b), ex = (u >> 52) & 0x7FFull;
        return (u & 0x000FFFFFFFFFFFFFull) == 0 && ex >= 2 && ex <= 2044;
    } else {
        const unsigned u = __float_as_uint((float)b), ex = (u >> 23) & 0xFFu;
        return (u & 0x007FFFFFu) == 0 && ex >= 2 && ex <= 252;
    }
}
template <bool FAST> __device__ __forceinline__ real_t div_shared(real_t a, real_t b, real_t y);
template <bool FAST> __device__ __forceinline__ real_t div_shared(real_t a, real_t b, real_t y, bool bok)
{
    if constexpr (FAST && sizeof(real_t) == 8) {
        const double ma = fabs(a);
        if (!(bok && ma >= 0x1p-800 && ma <= 0x1p800)) return a / b;
        const double q0 = a * y;
        const double r0 = __builtin_fma(-b, q0, a);
        const double q1 = __builtin_fma(r0, y, q0);
        const double r1 = __builtin_fma(-b, q1, a);
        return __builtin_fma(r1, y, q1);
    } else {
        return a / b;
    }
}
template <bool FAST> __device__ __forceinline__ real_t div_shared(real_t a, real_t b, real_t y)
{
    if constexpr (FAST && sizeof(real_t) == 8) {
        const double q0 = a * y;
        const double m = fabs(q0), ma = fabs(a);
        if (!(m >= 0x1p-900 && m <= 0x1p900 && ma >= 0x1p-900 && ma <= 0x1p900)) return a / b;
        const double r0 = __builtin_fma(-b, q0, a);
        const double q1 = __builtin_fma(r0, y, q0);
        const double r1 = __builtin_fma(-b, q1, a);
        return __builtin_fma(r1, y, q1);
    } else {
        return a / b;
    }
}

// the five quotients of column (ii, j) from the window W (rows j-2 .. j+2, columns i-2 .. i+3 of x, zero outside the grid), o =
// ii - i.  INTERIOR: every neighbour of every evaluated row exists (no guards).
// NUMER: q receives the five DIFFERENCES (the caller divides: k_f_stencil5_store_wave4).
// YGIVEN: the caller hands in yd = 1 / ed (one reciprocal per COLOUR, fetched from the colour's lane, instead of one per column).
template <int MODE, int SK, bool INTERIOR, bool FASTDIV, int WC, bool NUMER = false, bool YGIVEN = false>
__device__ __forceinline__ void stencil5_column_quotients(const real_t (&W)[5][WC], int o, int ii, int j, int nx, int ny, real_t e, real_t *q,
                                                          real_t yd_in = 0)
{
    const real_t ed = MODE == 1 ? 2 * e : e;
    const real_t yd = YGIVEN ? yd_in : (FASTDIV && sizeof(real_t) == 8) ? (real_t)1 / ed : (real_t)0;
    const real_t xc = W[2][2 + o], pc = xc + e, mc = xc - e;
    const bool hw = INTERIOR || ii > 0, he = INTERIOR || ii < nx - 1, hs = INTERIOR || j > 0, hn = INTERIOR || j < ny - 1;
    const real_t z = 0;
    // PV: what the plus point holds at an unperturbed coordinate (x + 0.0); MV: the minus point (x - 0.0 == x) -- and, for
    // forward differences, the base point x itself
#define PV(dj, di) (W[(dj) + 2][(di) + 2 + o] + z)
#define MV(dj, di) W[(dj) + 2][(di) + 2 + o]
    const real_t mcc = MODE == 1 ? mc : xc;
    {   // row (ii, j-1): its north neighbour is the perturbed coordinate
        const bool rhs = INTERIOR || j - 1 > 0;
        const real_t pl = stencil5_row<real_t, SK>(PV(-1, 0), hw ? PV(-1, -1) : z, he ? PV(-1, 1) : z, rhs ? PV(-2, 0) : z, pc);
        const real_t mi = stencil5_row<real_t, SK>(MV(-1, 0), hw ? MV(-1, -1) : z, he ? MV(-1, 1) : z, rhs ? MV(-2, 0) : z, mcc);
        q[0] = NUMER ? sub_exact(pl, mi) : div_shared<FASTDIV>(sub_exact(pl, mi), ed, yd);
    }
    {   // row (ii-1, j): east
        const bool rhw = INTERIOR || ii - 1 > 0;
        const real_t pl = stencil5_row<real_t, SK>(PV(0, -1), rhw ? PV(0, -2) : z, pc, hs ? PV(-1, -1) : z, hn ? PV(1, -1) : z);
        const real_t mi = stencil5_row<real_t, SK>(MV(0, -1), rhw ? MV(0, -2) : z, mcc, hs ? MV(-1, -1) : z, hn ? MV(1, -1) : z);
        q[1] = NUMER ? sub_exact(pl, mi) : div_shared<FASTDIV>(sub_exact(pl, mi), ed, yd);
    }
    {   // row (ii, j): centre
        const real_t pl = stencil5_row<real_t, SK>(pc, hw ? PV(0, -1) : z, he ? PV(0, 1) : z, hs ? PV(-1, 0) : z, hn ? PV(1, 0) : z);
        const real_t mi = stencil5_row<real_t, SK>(mcc, hw ? MV(0, -1) : z, he ? MV(0, 1) : z, hs ? MV(-1, 0) : z, hn ? MV(1, 0) : z);
        q[2] = NUMER ? sub_exact(pl, mi) : div_shared<FASTDIV>(sub_exact(pl, mi), ed, yd);
    }
    {   // row (ii+1, j): west
        const bool rhe = INTERIOR || ii + 1 < nx - 1;
        const real_t pl = stencil5_row<real_t, SK>(PV(0, 1), pc, rhe ? PV(0, 2) : z, hs ? PV(-1, 1) : z, hn ? PV(1, 1) : z);
        const real_t mi = stencil5_row<real_t, SK>(MV(0, 1), mcc, rhe ? MV(0, 2) : z, hs ? MV(-1, 1) : z, hn ? MV(1, 1) : z);
        q[3] = NUMER ? sub_exact(pl, mi) : div_shared<FASTDIV>(sub_exact(pl, mi), ed, yd);
    }
    {   // row (ii, j+1): south
        const bool rhn = INTERIOR || j + 1 < ny - 1;
        const real_t pl = stencil5_row<real_t, SK>(PV(1, 0), hw ? PV(1, -1) : z, he ? PV(1, 1) : z, pc, rhn ? PV(2, 0) : z);
        const real_t mi = stencil5_row<real_t, SK>(MV(1, 0), hw ? MV(1, -1) : z, he ? MV(1, 1) : z, mcc, rhn ? MV(2, 0) : z);
        q[4] = NUMER ? sub_exact(pl, mi) : div_shared<FASTDIV>(sub_exact(pl, mi), ed, yd);
    }
#undef PV
#undef MV
}

template <typename CT, int MODE, int SK>
__global__ void __launch_bounds__(kBlock, 6)       // a register budget for six waves per SIMD (78 VGPRs, no spills: 128 -> 121 us in round 3)
k_f_stencil5_store_wave(const real_t *__restrict__ x, const real_t *__restrict__ eps, fd_stencil5_store st, int64_t jrow0, int64_t jrow1)
{
    // two columns per lane: aligned 16-B loads of x, 128 columns per wavefront
    constexpr int TW = 128, WC = 6;
    constexpr bool FASTDIV = true;      // the shared-reciprocal exact division (the IEEE sequence spills under the register budget: 337 us)
    __shared__ __attribute__((aligned(16))) real_t s_win[kBlock / 64][FD_STENCIL5_WAVE_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nx = (int)st.nx, ny = (int)st.ny;
    const int TPR = (nx + TW - 1) / TW;
    const int64_t ntiles = (jrow1 - jrow0) * TPR, ngroups = (ntiles + kBlock / 64 - 1) / (kBlock / 64);
    const int64_t grp = xcd_tile(blockIdx.x, ngroups);
    if (grp >= ngroups) return;
    const int64_t wt = grp * (kBlock / 64) + wave;
    if (wt >= ntiles) return;
    // (32-bit division: the launcher declines grids with 2^31 tiles or more; a 64-bit one is ~150 instructions per wavefront)
    const unsigned wrow = (unsigned)wt / (unsigned)TPR;
    const int j = (int)(jrow0 + wrow), i0 = (int)((unsigned)wt - wrow * (unsigned)TPR) * TW;
    // (scalar copies of j and i0 for the addresses alone -- readfirstlane -- measured no change: 100.6 / 103.0 us, scripts/ab_c3.sh)
    const int i = i0 + 2 * lane;
    const int64_t k = (int64_t)j * nx + i;
    const bool act = i < nx;
    // a tile whose whole neighbourhood lies inside the grid needs no guards (wave-uniform)
    const bool interior = j >= 2 && j + 2 < ny && i0 >= 2 && i0 + TW + 2 <= nx;
    // window rows j-2 .. j+2, columns i-2 .. i+3 (zero outside the grid; of rows j+-2 only the lane's own columns are used)
    real_t W[5][WC];
    // The first and the last tile of a grid row (2 of 32 at nx = 4000) with two rows above and three below: every address of the
    // window is inside the array, so the loads stay unconditional (an idle lane reads the tile's first column) and the coordinates
    // outside the grid are SELECTED to 0 -- loads inside per-lane conditionals are waited for one by one (the 7-point kernel's
    // lesson, profiles/r04_y_lap7_taken_apart.md).
    const bool edge_tile = !interior && j >= 2 && j + 3 < ny;
    if (edge_tile) {
        const int64_t kl = act ? k : (int64_t)j * nx + i0;
#pragma unroll
        for (int dj = -2; dj <= 2; ++dj) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                r2_t v = {0, 0};
                if (!((dj == -2 || dj == 2) && c != 1)) {
                    const int ic = i + 2 * (c - 1);
                    const r2_t w = *reinterpret_cast<const r2_t *>(x + (kl + (int64_t)dj * nx + 2 * (c - 1)));
                    v.x = (act && ic >= 0 && ic < nx) ? w.x : (real_t)0;
                    v.y = (act && ic + 1 >= 0 && ic + 1 < nx) ? w.y : (real_t)0;
                }
                W[dj + 2][2 * c] = v.x; W[dj + 2][2 * c + 1] = v.y;
            }
        }
    } else
#pragma unroll
    for (int dj = -2; dj <= 2; ++dj) {
        const bool rowok = interior || (act && j + dj >= 0 && j + dj < ny);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r2_t v = {0, 0};
            if (!((dj == -2 || dj == 2) && c != 1)) {
                const int ic = i + 2 * (c - 1);
                const int64_t kc = k + (int64_t)dj * nx + 2 * (c - 1);
                if (interior || (rowok && ic >= 0 && ic + 1 < nx)) v = *reinterpret_cast<const r2_t *>(x + kc);
                else if (rowok) { if (ic >= 0 && ic < nx) v.x = x[kc]; if (ic + 1 >= 0 && ic + 1 < nx) v.y = x[kc + 1]; }
            }
            W[dj + 2][2 * c] = v.x; W[dj + 2][2 * c + 1] = v.y;
        }
    }
    int cpair[2] = {0, 0};
    if (act) { cpair[0] = (int)((const CT *)st.color)[k]; cpair[1] = (int)((const CT *)st.color)[k + 1]; }
    real_t q[10];
    // (one reciprocal per COLOUR fetched from the colour's lane, as the Float32 kernel below does: 101 -> 105 us here, scripts/ab_c3.sh)
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        if (interior) stencil5_column_quotients<MODE, SK, true, FASTDIV, WC>(W, o, i + o, j, nx, ny, eps[cpair[o]], q + 5 * o);
        else stencil5_column_quotients<MODE, SK, false, FASTDIV, WC>(W, o, i + o, j, nx, ny, eps[cpair[o]], q + 5 * o);
    }
    fd_stencil5_emit_wave<real_t, true, 2>(&st, s_win[wave], j, i0, q);
}

#ifdef FDJAC_F32
// Float32 (round 6): FOUR columns per lane -- 16-byte loads of x and 16-byte stores of the values (a Float32 pair moves 512 B per
// instruction, half of what the memory pipeline takes), 256 columns per wavefront -- and the division through Float64:
//   (float)((double)a * (1.0 / (double)b))  has the bits of the Float32  a / b  whenever that quotient is a NORMAL number: the double
// product is within 2^-52 of a / b, and a quotient of two 24-bit significands that is not itself a Float32 lies at least 2^-49 (relative)
// from every rounding boundary (scripts/ubench/exact_div32_probe.hip: 5e10 pairs, 0 mismatches; denormal quotients can differ).  The
// reciprocal belongs to the COLOUR: lane c of every wavefront forms 1 / (double)(2 eps_c) once and the columns fetch theirs by
// ds_bpermute; a column whose five quotients are not all normal numbers (zero, denormal, infinite, NaN -- or a step outside
// [2^-100, 2^100], whose reciprocal is handed out as NaN) takes the true divisions.  Three conversions / products and a class test
// per quotient instead of the ten instructions of v_div_scale .. v_div_fixup.
__device__ __forceinline__ void div5_through_f64(const float (&a)[5], float b, double y, float *q)
{
    bool ok = true;
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const float v = (float)((double)a[m] * y);
        q[m] = v;
        ok = ok && __builtin_amdgcn_classf(v, 0x108);          // -normal | +normal
    }
    if (!ok) {
#pragma unroll
        for (int m = 0; m < 5; ++m) q[m] = a[m] / b;
    }
}
template <typename CT, int MODE, int SK>
__global__ void __launch_bounds__(kBlock, 4)
k_f_stencil5_store_wave4(const real_t *__restrict__ x, const real_t *__restrict__ eps, fd_stencil5_store st, int64_t jrow0, int64_t jrow1)
{
    constexpr int TW = 256, WC = 8;
    __shared__ __attribute__((aligned(16))) real_t s_win[kBlock / 64][FD_STENCIL5_WAVE4_LDS];
    const int lane = threadIdx.x & 63, wave = wave_index_scalar();
    const int nx = (int)st.nx, ny = (int)st.ny;                 // nx is a multiple of 4 (launcher): a lane's four columns share a grid row
    const int TPR = (nx + TW - 1) / TW;
    const int64_t ntiles = (jrow1 - jrow0) * TPR, ngroups = (ntiles + kBlock / 64 - 1) / (kBlock / 64);
    const int64_t grp = xcd_tile(blockIdx.x, ngroups);
    if (grp >= ngroups) return;
    const int64_t wt = grp * (kBlock / 64) + wave;
    if (wt >= ntiles) return;
    const unsigned wrow = (unsigned)wt / (unsigned)TPR;
    const int j = (int)(jrow0 + wrow), i0 = (int)((unsigned)wt - wrow * (unsigned)TPR) * TW;
    const int i = i0 + 4 * lane;
    const int64_t k = (int64_t)j * nx + i;
    const bool act = i < nx;
    // the colours' steps and reciprocals, one colour per lane (st.C <= 64: launcher)
    real_t e_l = 1;
    double y_l = 0;
    if (lane < st.C) {
        e_l = eps[lane];
        const real_t ed = MODE == 1 ? 2 * e_l : e_l, mb = fabsf(ed);
        y_l = (mb >= 0x1p-100f && mb <= 0x1p100f) ? 1.0 / (double)ed : __builtin_nan("");
    }
    const bool interior = j >= 2 && j + 2 < ny && i0 >= 4 && i0 + TW + 4 <= nx;
    // window rows j-2 .. j+2, columns i-2 .. i+5 (zero outside the grid; of rows j+-2 only the lane's own columns are used)
    real_t W[5][WC];
    const bool edge_tile = !interior && j >= 2 && j + 3 < ny;   // every address below is inside the array: unconditional loads + selects
    if (interior || edge_tile) {
        const int64_t kl = act ? k : (int64_t)j * nx + i0;
#pragma unroll
        for (int dj = -2; dj <= 2; ++dj) {
            const real_t *row = x + (kl + (int64_t)dj * nx);
            const r4_t B = *reinterpret_cast<const r4_t *>(row);
            r4_t A = {0, 0, 0, 0}, Cq = {0, 0, 0, 0};
            if (dj >= -1 && dj <= 1) { A = *reinterpret_cast<const r4_t *>(row - 4); Cq = *reinterpret_cast<const r4_t *>(row + 4); }
            const bool okA = interior || (act && i >= 4), okB = interior || act, okC = interior || (act && i + 4 < nx);
            W[dj + 2][0] = okA ? A.z : (real_t)0; W[dj + 2][1] = okA ? A.w : (real_t)0;
            W[dj + 2][2] = okB ? B.x : (real_t)0; W[dj + 2][3] = okB ? B.y : (real_t)0;
            W[dj + 2][4] = okB ? B.z : (real_t)0; W[dj + 2][5] = okB ? B.w : (real_t)0;
            W[dj + 2][6] = okC ? Cq.x : (real_t)0; W[dj + 2][7] = okC ? Cq.y : (real_t)0;
        }
    } else {
        // the two first and the three last grid rows: every coordinate guarded
#pragma unroll
        for (int dj = -2; dj <= 2; ++dj) {
            const bool rowok = act && j + dj >= 0 && j + dj < ny;
#pragma unroll
            for (int c = 0; c < WC; ++c) {
                const int ic = i + c - 2;
                real_t v = 0;
                if (rowok && ic >= 0 && ic < nx && !((dj == -2 || dj == 2) && (c < 2 || c > 5))) v = x[k + (int64_t)dj * nx + (c - 2)];
                W[dj + 2][c] = v;
            }
        }
    }
    int cq[4] = {0, 0, 0, 0};
    if (act) {
        if constexpr (sizeof(CT) == 1) {
            const unsigned cw = *reinterpret_cast<const unsigned *>((const unsigned char *)st.color + k);       // (k is a multiple of 4)
            cq[0] = (int)(cw & 255u); cq[1] = (int)((cw >> 8) & 255u); cq[2] = (int)((cw >> 16) & 255u); cq[3] = (int)(cw >> 24);
        } else {
#pragma unroll
            for (int o = 0; o < 4; ++o) cq[o] = (int)((const CT *)st.color)[k + o];
        }
    }
    real_t q[20];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const real_t e = __shfl(e_l, cq[o], 64);
        const double y = __shfl(y_l, cq[o], 64);
        real_t num[5];
        if (interior) stencil5_column_quotients<MODE, SK, true, false, WC, true>(W, o, i + o, j, nx, ny, e, num);
        else stencil5_column_quotients<MODE, SK, false, false, WC, true>(W, o, i + o, j, nx, ny, e, num);
        div5_through_f64(num, MODE == 1 ? 2 * e : e, y, q + 5 * o);
    }
    fd_stencil5_emit_wave4<real_t, true>(&st, s_win[wave], j, i0, q);
}
#endif

// fd_bbb_store (round 5): the 5-point families on an nx x ny grid storing into BandedBlockBandedMatrix data -- ny blocks of nx rows,
// block bandwidths (1, 1), sub-block bandwidths (1, 1): the reference's own fixture (test/coloring_tests.jl:99-115;
// ext/FiniteDiffBlockBandedMatricesExt.jl:16-42).  Thread k = (i, j) owns column k: it loads the 13 coordinates its five rows read, forms
// the five quotients exactly as the CSC storing kernel does (stencil5_column_quotients) and writes the NINE slots of its column -- block
// j-1: (0, q_S, 0), block j: (q_W, q_C, q_E), block j+1: (0, q_N, 0); slots of rows outside their block and whole columns without a colour
// are 0, as k_decompress_bbb writes them.  A slot whose row does not depend on the column is exactly 0 in the hand-over path too (the
// plan verified that the colouring is valid for the nine-point BBB pattern: no perturbed column reaches such a row).
template <typename CT, int MODE, int SK>
__global__ void __launch_bounds__(kBlock) k_f_stencil5_store_bbb(const real_t *__restrict__ x, const real_t *__restrict__ eps, fd_bbb_store st, int nx, int ny)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= st.N) return;
    const int j = (int)(k / nx), i = (int)(k - (int64_t)j * nx);
    real_t W[5][5];
#pragma unroll
    for (int dj = -2; dj <= 2; ++dj)
#pragma unroll
        for (int di = -2; di <= 2; ++di) {
            W[dj + 2][di + 2] = 0;
            if ((dj < 0 ? -dj : dj) + (di < 0 ? -di : di) > 2) continue;      // (only the 13 points within L1 distance 2 are read)
            const int ii = i + di, jj = j + dj;
            const bool in = ii >= 0 && ii < nx && jj >= 0 && jj < ny;
            const real_t v = x[in ? (int64_t)jj * nx + ii : k];              // (unconditional load from a clamped index)
            W[dj + 2][di + 2] = in ? v : (real_t)0;
        }
    const int c = (int)((const CT *)st.color)[k];
    const bool none = c == (int)(CT)(-1);
    real_t q[5] = {0, 0, 0, 0, 0};
    const real_t e = none ? (real_t)1 : eps[c];
    if (!none) stencil5_column_quotients<MODE, SK, false, true, 5>(W, 0, i, j, nx, ny, e, q);
    // an in-block row that does not depend on the column: the hand-over path divides an exact +0.0 difference by the step -- the sign
    // of that zero follows the step's (dir = -1)
    const real_t zq = (real_t)0 / (MODE == 1 ? 2 * e : e);
    // rows k - nx, k - 1, k, k + 1, k + nx; those that do not exist hold garbage in q: they are never stored
    real_t *out = (real_t *)st.out;
    const long long sJ = st.stride[j];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const long long st0 = st.start[d + 3 * (long long)j];
        if (st0 < 0) continue;
        const int K = j + d - 1;                                          // (bu = 1)
        const bool kin = K >= 0 && K < ny;
        real_t *o = out + st0 + (long long)i * sJ;                         // slots t = 0, 1, 2 <-> rows i - 1, i, i + 1 of block K (mu = 1)
        const real_t mid = d == 0 ? q[0] : d == 1 ? q[2] : q[4];
        const real_t lo = d == 1 ? q[1] : zq, hi = d == 1 ? q[3] : zq;
        o[0] = (kin && !none && i - 1 >= 0) ? lo : (real_t)0;
        o[1] = (kin && !none) ? mid : (real_t)0;
        o[2] = (kin && !none && i + 1 < nx) ? hi : (real_t)0;
    }
}

template <typename CT>
static int lazy_stencil5_launch(BuiltinF *b, void *fx, const fd_lazy_points *lp, int64_t fs, int64_t r0, int64_t r1,
                                hipStream_t s)
{
    const int64_t r0e = r0 & ~(int64_t)1;
    const int64_t ntiles = (r1 - r0e + 2 * kBlock - 1) / (2 * kBlock);
    const unsigned g = (unsigned)(8 * xcd_chunks(ntiles));
    const int sk = b->family == FD_F_CLAMP5 ? 1 : b->family == FD_F_LAP5_NL ? 2 : 0;
    const int mode = lp->is_complex ? 2 : (lp->pts == 2 ? 1 : 0);
    if (lp->store) {
        // the launch stores the stencil's CSC Jacobian itself (fd_stencil5_store): every colour in one batch, the Laplacian
        // fixtures (the clamped sum's boundary rows reach themselves twice: not this pattern's arithmetic)
        if (lp->store_kind == FD_STORE_BBB && mode != 2 && sk != 1) {
            // BandedBlockBandedMatrix data: ny blocks of nx rows, (1, 1) / (1, 1) bandwidths -- the structure of this family's Jacobian
            const fd_bbb_store bb = *(const fd_bbb_store *)lp->store;
            if (bb.elem_bytes != (int)sizeof(real_t) || bb.block_size != b->prm[0] || bb.nblk != b->prm[1] || bb.bl != 1 || bb.bu != 1 || bb.lam != 1 ||
                bb.mu != 1 || lp->c_lo != 0 || lp->ncolors != bb.C || bb.color_bytes != (int)sizeof(CT) || bb.N >= ((int64_t)1 << 31))
                return FD_LAZY_DECLINED;
            const unsigned gb = (unsigned)((bb.N + kBlock - 1) / kBlock);
#define FD_S5B(MODE, SKK)                                                                                                  \
            hipLaunchKernelGGL((k_f_stencil5_store_bbb<CT, MODE, SKK>), dim3(gb), dim3(kBlock), 0, s, (const real_t *)lp->x,     \
                               (const real_t *)lp->eps, bb, (int)b->prm[0], (int)b->prm[1])
            if (mode == 0) { if (sk == 2) FD_S5B(0, 2); else FD_S5B(0, 0); }
            else { if (sk == 2) FD_S5B(1, 2); else FD_S5B(1, 0); }
#undef FD_S5B
            return hipGetLastError() == hipSuccess ? 0 : 4;
        }
        if (lp->store_kind != FD_STORE_STENCIL5 || mode == 2 || sk == 1) return FD_LAZY_DECLINED;
        const fd_stencil5_store st = *(const fd_stencil5_store *)lp->store;
        if (st.elem_bytes != (int)sizeof(real_t) || st.nx != b->prm[0] || st.ny != b->prm[1] || lp->c_lo != 0 || lp->ncolors != st.C ||
            st.color_bytes != (int)sizeof(CT) || (((uintptr_t)lp->x) & kPairMask) != 0 || st.col_end <= st.col_begin)
            return FD_LAZY_DECLINED;
        const int64_t jrow0 = st.col_begin / st.nx, jrow1 = (st.col_end - 1) / st.nx + 1;      // grid rows with local columns
        const int64_t ntiles = (jrow1 - jrow0) * ((st.nx + 127) / 128), ngroups = (ntiles + kBlock / 64 - 1) / (kBlock / 64);
        const unsigned gs = (unsigned)(8 * xcd_chunks(ngroups));
        if (ntiles >= ((int64_t)1 << 31) || st.nx * st.ny >= ((int64_t)1 << 31)) return FD_LAZY_DECLINED;   // (32-bit tile / grid arithmetic in the kernel)
        // ONE form (round 3 measured the others, profiles/r03_h_stencil_store_ab.txt; the probes live in scripts/ubench/): two columns per
        // lane, non-temporal stores, the shared-reciprocal exact division, guard-free interior tiles, a register budget for six waves per SIMD
#ifdef FDJAC_F32
        {   // Float32: four columns per lane, the division through Float64 (k_f_stencil5_store_wave4)
            const char *sw = fdjac::test_switch("FDJAC_S5_WAVE4");
            const bool quad_ok = st.nx % 4 == 0 && st.C <= 64 && (((uintptr_t)lp->x) & 15) == 0 && (((uintptr_t)st.color) & 3) == 0 &&
                                 !(sw && *sw && atoi(sw) == 0);
            const int64_t nt4 = (jrow1 - jrow0) * ((st.nx + 255) / 256), ng4 = (nt4 + kBlock / 64 - 1) / (kBlock / 64);
            if (quad_ok) {
                const unsigned g4 = (unsigned)(8 * xcd_chunks(ng4));
#define FD_S54(MODE, SKK)                                                                                            \
                hipLaunchKernelGGL((k_f_stencil5_store_wave4<CT, MODE, SKK>), dim3(g4), dim3(kBlock), 0, s, (const real_t *)lp->x, \
                                   (const real_t *)lp->eps, st, jrow0, jrow1)
                if (mode == 0) { if (sk == 2) FD_S54(0, 2); else FD_S54(0, 0); }
                else { if (sk == 2) FD_S54(1, 2); else FD_S54(1, 0); }
#undef FD_S54
                return hipGetLastError() == hipSuccess ? 0 : 4;
            }
        }
#endif
#define FD_S5(MODE, SKK)                                                                                            \
        hipLaunchKernelGGL((k_f_stencil5_store_wave<CT, MODE, SKK>), dim3(gs), dim3(kBlock), 0, s, (const real_t *)lp->x, \
                           (const real_t *)lp->eps, st, jrow0, jrow1)
        if (mode == 0) { if (sk == 2) FD_S5(0, 2); else FD_S5(0, 0); }
        else { if (sk == 2) FD_S5(1, 2); else FD_S5(1, 0); }
#undef FD_S5
        return hipGetLastError() == hipSuccess ? 0 : 4;
    }
#define FD_LAZY(MODE, CL)                                                                                          \
    hipLaunchKernelGGL((k_f_stencil5_lazy<CT, MODE, CL>), dim3(g), dim3(kBlock), 0, s, (real_t *)fx, fs,            \
                       (real_t *)lp->base_out, (const real_t *)lp->x, (const CT *)lp->color, (const real_t *)lp->eps, lp->c_lo,    \
                       lp->ncolors, b->prm[0], b->prm[1], r0e, r1, lp->imag_only, mode != 2 ? lp->diff : 0)
#define FD_LAZY_SK(MODE) do { if (sk == 1) FD_LAZY(MODE, 1); else if (sk == 2) FD_LAZY(MODE, 2); else FD_LAZY(MODE, 0); } while (0)
    if (mode == 0) FD_LAZY_SK(0);
    else if (mode == 1) FD_LAZY_SK(1);
    else FD_LAZY_SK(2);
#undef FD_LAZY_SK
#undef FD_LAZY
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// Lazy-point version of the block-coupled family.  One workgroup owns kBcG consecutive blocks: phase A forms
// sig_b of those blocks and their two neighbours for EVERY point of the batch (x~ built in registers, the same
// lane-order + shuffle-tree sum as k_f_block_sigma), phase B evaluates the rows.  x / colours are read once for all
// points, sin(x) / cos(x) once per row (a point differs from the base in at most the row's own coordinate), and the
// perturbed points are never written: the complex-step BlockBanded configuration drops from perturb 88 us + f! 596 us
// to one ~memory-bound launch.  Bit-identical to perturb + k_f_block_sigma + k_f_block_apply.
constexpr int kBcG = 6;
template <typename T> __device__ __forceinline__ T tree_sum64(T acc)
{
    for (int off = 32; off > 0; off >>= 1) {
        if constexpr (sizeof(T) == sizeof(real_t)) {
            real_t o = __shfl_down(*reinterpret_cast<real_t *>(&acc), off, 64);
            acc = acc + *reinterpret_cast<T *>(&o);
        } else {
            cd *a = reinterpret_cast<cd *>(&acc);
            cd o{__shfl_down(a->re, off, 64), __shfl_down(a->im, off, 64)};
            acc = acc + *reinterpret_cast<T *>(&o);
        }
    }
    return acc;
}
// lanes of ONE wave exchange data through LDS: LDS instructions of a wave execute in order, this keeps the compiler
// from reordering them across the hand-over
__device__ __forceinline__ void bc_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int MODE> struct BcT { typedef real_t type; };
template <> struct BcT<2> { typedef cd type; };
__device__ __forceinline__ real_t bc_make(real_t x, real_t d, int sgn, real_t) { return sgn == 0 ? x + d : x - d; }
__device__ __forceinline__ cd bc_make(real_t x, real_t d, int, cd) { return cd{x, d}; }

template <typename CT, int MODE>
__global__ void __launch_bounds__(kBlock)
k_f_blockcoupled_lazy(real_t *__restrict__ fx, int64_t fs, real_t *__restrict__ base_out, const real_t *__restrict__ x,
                      const CT *__restrict__ color, const real_t *__restrict__ eps, int c_lo, int B, int64_t nb, int bs,
                      int64_t blk0, int64_t blk1, int64_t r0, int64_t r1, int imag_only)
{
    typedef typename BcT<MODE>::type T;
    extern __shared__ real_t s_bc[];
    constexpr int pts = MODE == 1 ? 2 : 1;
    constexpr int NBLK = kBcG + 2;
    const int PB = B * pts + 1;                      // last slot: the unperturbed point (base_out, forward only)
    T *sig = reinterpret_cast<T *>(s_bc);            // [NBLK][PB]            sigma of every point
    T *tree = sig + (size_t)NBLK * PB;               // [NBLK][128]           the batch-base summation tree (levels 0..5)
    int *owner = reinterpret_cast<int *>(tree + (size_t)NBLK * 128);   // [waves][B]  duplicate-colour detection
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t g0 = blk0 + (int64_t)blockIdx.x * kBcG;
    const real_t w = (real_t)(lane + 1) / (real_t)bs;
    const bool want_base = (MODE == 0) && (base_out != nullptr);
    int *own = owner + wave * B;

    // phase A: sig of blocks g0-1 .. g0+kBcG for every point.  A point differs from the batch's base values
    // (x_j + 0.0) in the lanes whose colour it is -- normally ONE lane per block (the columns of a dense block all
    // conflict), and then only the six partial sums on that leaf's path to the root of the fixed summation tree
    // change: keep the tree of the base values in LDS and let the lane re-add along its path.  (Blocks with a repeated
    // colour take the general path: one full tree per point.)  Same additions as k_f_block_sigma => same bits.
    for (int lb = wave; lb < NBLK; lb += kBlock / 64) {
        const int64_t bb = g0 - 1 + lb;
        const bool inb = (bb >= 0) & (bb < nb);
        const bool act = inb & (lane < bs);
        real_t xj = 0.0;
        int cj = -1;
        if (act) {
            xj = x[bb * bs + lane];
            const int c = (int)color[bb * bs + lane];
            cj = (c == (int)(CT)(-1) || c < 0) ? -1 : c - c_lo;
            if (cj >= B) cj = -1;
        }
        T *tr = tree + (size_t)lb * 128;
        T *sg = sig + (size_t)lb * PB;
        // base tree of the batch: leaves 0.0 + w*(x + 0.0) (what an unperturbed lane of any point contributes)
        T acc = act ? zero_of<T>() + w * bc_make(xj, 0.0, 0, T{}) : zero_of<T>();
        tr[lane] = acc;
        int base_at = 64;
        for (int off = 32; off > 0; off >>= 1) {
            if constexpr (sizeof(T) == sizeof(real_t)) {
                real_t o = __shfl_down(*reinterpret_cast<real_t *>(&acc), off, 64);
                acc = acc + *reinterpret_cast<T *>(&o);
            } else {
                cd *a = reinterpret_cast<cd *>(&acc);
                cd o{__shfl_down(a->re, off, 64), __shfl_down(a->im, off, 64)};
                acc = acc + *reinterpret_cast<T *>(&o);
            }
            if (lane < off && off > 1) tr[base_at + lane] = acc;   // level partials: 32 @64, 16 @96, 8 @112, 4 @120, 2 @124
            base_at += off;
        }
        T root = acc;   // valid in lane 0
        if constexpr (sizeof(T) == sizeof(real_t)) {
            real_t r = __shfl(*reinterpret_cast<real_t *>(&root), 0, 64);
            root = *reinterpret_cast<T *>(&r);
        } else {
            cd *a = reinterpret_cast<cd *>(&root);
            cd r{__shfl(a->re, 0, 64), __shfl(a->im, 0, 64)};
            root = *reinterpret_cast<T *>(&r);
        }
        bc_wave_sync();   // the tree written by this wave's lanes is read by other lanes below
        // duplicate colours inside the block?
        for (int q = lane; q < B; q += 64) own[q] = -1;
        bc_wave_sync();
        if (cj >= 0) own[cj] = lane;
        bc_wave_sync();
        const bool dup = (cj >= 0) && (own[cj] != lane);
        const bool any_dup = __builtin_amdgcn_ballot_w64(dup) != 0;
        if (!any_dup) {
            for (int q = lane; q < PB - 1; q += 64) sg[q] = inb ? root : zero_of<T>();
            bc_wave_sync();
            if (cj >= 0) {
                const real_t e = eps[c_lo + cj];
#pragma unroll
                for (int sgn = 0; sgn < pts; ++sgn) {
                    T n = zero_of<T>() + w * bc_make(xj, e, sgn, T{});
                    int idx = lane;
                    n = n + tr[idx ^ 32]; idx &= 31;
                    n = n + tr[64 + (idx ^ 16)]; idx &= 15;
                    n = n + tr[96 + (idx ^ 8)]; idx &= 7;
                    n = n + tr[112 + (idx ^ 4)]; idx &= 3;
                    n = n + tr[120 + (idx ^ 2)]; idx &= 1;
                    n = n + tr[124 + (idx ^ 1)];
                    sg[sgn * B + cj] = n;
                }
            }
        } else {
            for (int q = 0; q < PB - 1; ++q) {
                T term = zero_of<T>();
                if (act) {
                    const int b = q < B ? q : q - B;
                    const real_t d = (cj == b) ? eps[c_lo + b] : 0.0;
                    term = zero_of<T>() + w * bc_make(xj, d, q < B ? 0 : 1, T{});
                }
                term = tree_sum64<T>(term);
                if (lane == 0) sg[q] = inb ? term : zero_of<T>();
            }
        }
        if (want_base) {   // the base evaluation f(x) uses x itself (not x + 0.0); MODE 0 only, T = real_t
            real_t tb = act ? 0.0 + w * xj : 0.0;
            tb = tree_sum64<real_t>(tb);
            if (lane == 0) reinterpret_cast<real_t *>(sg)[PB - 1] = inb ? tb : 0.0;
        }
    }
    __syncthreads();

    for (int lb = 1 + wave; lb <= kBcG; lb += kBlock / 64) {
        const int64_t bb = g0 - 1 + lb;
        if (bb >= blk1 || bb >= nb) continue;
        const int64_t k = bb * bs + lane;
        if (!(lane < bs && k >= r0 && k < r1)) continue;
        const real_t xk = x[k];
        const int c = (int)color[k];
        const int ck = (c == (int)(CT)(-1) || c < 0) ? -1 : c - c_lo;
        const bool mine = (ck >= 0) & (ck < B);
        const real_t e = mine ? eps[c_lo + ck] : 0.0;
        // values every point shares, and the ones of the row's own colour
        const real_t x0 = xk + 0.0;                 // what an unperturbed point of the batch holds (x + 0.0)
        real_t s0, sP = 0.0, sM = 0.0, c0 = 0.0, ch = 1.0, sh = 0.0;
        const real_t ch0 = cosh(0.0 * xk), sh0 = sinh(0.0 * xk);   // cosh(0), sinh(0) through the same library calls
        if (MODE == 2) {
            s0 = sin(xk); c0 = cos(xk);
            if (mine) { ch = cosh(e); sh = sinh(e); }
        } else {
            s0 = sin(x0);
            if (mine) { sP = sin(xk + e); if (MODE == 1) sM = sin(xk - e); }
        }
        const T *sm = sig + (size_t)(lb - 1) * PB, *sc = sig + (size_t)lb * PB, *sp = sig + (size_t)(lb + 1) * PB;
        for (int q = 0; q < PB - 1; ++q) {
            const int b = q < B ? q : q - B;
            const bool hit = mine & (ck == b);
            const T S = (sm[q] + sc[q]) + sp[q];
            if constexpr (MODE == 2) {
                const cd xt{xk, hit ? e : (real_t)0};
                const cd sn{s0 * (hit ? ch : ch0), c0 * (hit ? sh : sh0)};
                const cd v = xt * S + sn;
                if (imag_only) fx[(int64_t)q * fs + k] = v.im;
                else *reinterpret_cast<r2_t *>(fx + ((int64_t)q * fs + k) * 2) = r2_t{v.re, v.im};
            } else {
                const real_t d = hit ? e : 0.0;
                const real_t xt = q < B ? xk + d : xk - d;
                const real_t sn = hit ? (q < B ? sP : sM) : s0;
                fx[(int64_t)q * fs + k] = xt * S + sn;
            }
        }
        if (want_base) {
            const real_t *sb = reinterpret_cast<const real_t *>(sig);
            const real_t S = (sb[(size_t)(lb - 1) * PB + PB - 1] + sb[(size_t)lb * PB + PB - 1]) + sb[(size_t)(lb + 1) * PB + PB - 1];
            base_out[k] = xk * S + sin(xk);
        }
    }
}

// fd_lazy_points.store with a fd_colrange_store (include/fdjac_device.h), complex step: the block-coupled fixture evaluated at
// the lazily perturbed points and imag(f) / eps stored into the BlockBandedMatrix data by this launch (config 5: 245 MB of stored
// values written once, instead of written as f! values, read back and written again -- 0.195 -> 0.06 ms per Jacobian).
// Phase A is k_f_blockcoupled_lazy's: sigma of every point of the batch for the blocks g0-2 .. g0+kBcS+1 (the tree of the base
// values in LDS, a perturbed lane re-adds along its leaf-to-root path).  Phase B is column-centric: for column (b, j) -- point q = its
// colour -- and a row k of block b' in b-1 .. b+1 the value is imag(x~_k * (sig_{b'-1} + sig_{b'} + sig_{b'+1}) + sin(x~_k)) at point
// q, the operations of k_f_blockcoupled_lazy's phase B on the same operands (the plan verified that no other column of colour q
// touches those rows), divided by eps_q: same bits as the hand-over path.
#ifndef FD_BCS
#define FD_BCS 4
#endif
#ifndef FD_BCT
#define FD_BCT 512
#endif
constexpr int kBcT = FD_BCT;      // threads per workgroup
constexpr int kBcS = FD_BCS;      // block-columns per workgroup (2 / 3 / 4 / 6 / 8 / 10 measured: 57 / 62 / 57 / 63 / 67 / 75 us on config 5)
// complex items of the LDS region phase A uses for its trees and owners and phase B for the S sums
__host__ __device__ constexpr size_t bcs_shared_items(int B)
{
    const size_t item = 2 * sizeof(real_t);
    const size_t a = (size_t)(kBcS + 2) * (size_t)B, b = (size_t)(kBcT / 64) * 128 + ((size_t)(kBcT / 64) * (size_t)B * 4 + item - 1) / item;
    return a > b ? a : b;
}
__device__ __forceinline__ int bc_lane_int(int v, int i) { return __builtin_amdgcn_readlane(v, i); }
__device__ __forceinline__ long long bc_lane_i64(long long v, int i)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v & 0xffffffffu), i);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), i);
    return (long long)(((unsigned long long)hi << 32) | lo);
}
#ifndef FD_BC_RPL32
#define FD_BC_RPL32 2      // Float32: row pairs of four columns (29.2 us at config 5) or row quads of eight (29.9 us), same box
#endif
// RPL consecutive rows of one column: one 16-byte store (Float64 pair / Float32 quad; a Float32 destination that is only 8-byte aligned: two pairs)
template <int RPL> __device__ __forceinline__ void bc_store_rows(real_t *d, const real_t (&v)[RPL], bool al)
{
    if constexpr (RPL == 2) {
        __builtin_nontemporal_store(r2_t{v[0], v[1]}, reinterpret_cast<r2_t *>(d));
    } else {
        typedef real_t rq_t __attribute__((ext_vector_type(4)));
        if (al) __builtin_nontemporal_store(rq_t{v[0], v[1], v[2], v[3]}, reinterpret_cast<rq_t *>(d));
        else {
            __builtin_nontemporal_store(r2_t{v[0], v[1]}, reinterpret_cast<r2_t *>(d));
            __builtin_nontemporal_store(r2_t{v[2], v[3]}, reinterpret_cast<r2_t *>(d + 2));
        }
    }
}
// QUAD (round 5; blocks of exactly 32 rows, even destinations: fd_colrange_store.pairs): a lane of phase B is a ROW PAIR of one of
// FOUR adjacent columns -- every store is 16 bytes per lane, 256 contiguous bytes per 16 lanes, half the store instructions of the
// two-column form; the same operations per element, so the same bits.  Float32 (round 6): row pairs too (8-byte stores, 33.8 -> 29.2 us
// at config 5; FD_BC_RPL32 = 4 makes it a ROW QUAD of one of EIGHT adjacent columns, 16-byte stores: 29.9 us on the same box).
template <typename CT, bool TWO, bool QUAD = false>
__global__ void __launch_bounds__(kBcT)
k_f_blockcoupled_store(const real_t *__restrict__ x, const CT *__restrict__ color, const real_t *__restrict__ eps, int c_lo, int B,
                       int64_t nb, int bs, int64_t blk0, int64_t blk1, fd_colrange_store st)
{
    typedef cd T;
    extern __shared__ real_t s_bcs[];
    constexpr int NBLK = kBcS + 4, NW = kBcT / 64, NIT = (NBLK + NW - 1) / NW;
    const int PB = B + 1;
    constexpr int RS = TWO ? 32 : 64;                // rows of a block in the row arrays
    constexpr int NU = (2 * kBcS + NW - 1) / NW;     // units of phase B per wave
    T *sig = reinterpret_cast<T *>(s_bcs);            // [NBLK][PB]   sigma of every point
    T *ssum = sig + (size_t)NBLK * PB;               // [kBcS + 2][B] (sig[k-1] + sig[k]) + sig[k+1] of the row blocks g0-1 .. g0+kBcS;
    T *tree = ssum;                                  //   before that, phase A's [NW][128] batch-base summation trees
    int *owner = reinterpret_cast<int *>(tree + (size_t)NW * 128);      //   and [NW][B] owners
    real_t *rx = reinterpret_cast<real_t *>(ssum + bcs_shared_items(B));   // [(kBcS + 2) * RS] x of the rows of those blocks
    real_t *rc = rx + (kBcS + 2) * RS;               //   cos(x)
    real_t *rz = rc + (kBcS + 2) * RS;               //   cos(x) * sinh(0 * x): imag(sin(x~)) of a row the point does not perturb
    real_t *ce = rz + (kBcS + 2) * RS;               // [B] step size of every colour of the batch,
    real_t *cy = ce + B;                             //     its reciprocal (div_shared),
    real_t *cs = cy + B;                             //     sinh(eps): imag(sin(x + i eps)) = cos(x) sinh(eps)
    const int lane = threadIdx.x & 63, wave = wave_index_scalar();
    const int64_t g0 = blk0 + (int64_t)blockIdx.x * kBcS;
    const real_t w = (real_t)(lane + 1) / (real_t)bs;
    int *own = owner + wave * B;
    const int ucols = (bs + 1) / 2;                  // columns of a unit of phase B (half a block-column)

    // every global load of phase A up front (the blocks g0-2 .. g0+kBcS+1 this wave sums: x and the colour of their columns)
    real_t xa[NIT];
    int ca[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int lb = wave + it * NW;
        const int64_t bb = g0 - 2 + lb;
        const bool act = (lb < NBLK) & (bb >= 0) & (bb < nb) & (lane < bs);
        xa[it] = 0.0; ca[it] = -1;
        if (act) {
            xa[it] = x[bb * bs + lane];
            const int c = (int)color[bb * bs + lane];
            int cj = (c == (int)(CT)(-1) || c < 0) ? -1 : c - c_lo;
            if (cj >= B) cj = -1;
            ca[it] = cj;
        }
    }
    // PACK (Float32, blocks of <= 32 rows): x, cos(x), cos(x) sinh(0 x) of the rows phase B evaluates -- blocks g0-1 .. g0+kBcS -- are formed
    // by FULL wavefronts, two blocks each (lanes 0-31 / 32-63), instead of by the half-empty wavefront that sums the block: half the wave
    // instructions of the library cosine; the same function of the same operand: same bits.  One box, events (scripts/ab_c5.sh): Float32
    // 29.0 -> 25.8 us at config 5 -- used; Float64 48.2 -> 50.8 us -- not used.
    constexpr bool PACK = TWO && sizeof(real_t) == 4;
    real_t xr = 0;
    bool actr = false;
    const int lbr = 1 + 2 * wave + (lane >> 5);
    if constexpr (PACK) {
        const int64_t bbr = g0 - 2 + lbr;
        actr = (lbr <= kBcS + 2) & (bbr >= 0) & (bbr < nb) & ((lane & 31) < bs);
        if (actr) xr = x[bbr * bs + (lane & 31)];
    }
    // ... and of phase B: lane i holds the colour and the destination of column i of each unit the wave will store
    int uq[NU];
    long long ud[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int u = wave + k * NW;
        const int cstart = (u & 1) * ucols, ncol = min(bs, cstart + ucols) - cstart;
        const int64_t bcol = g0 + (u >> 1);
        uq[k] = -1; ud[k] = 0;
        if (u < 2 * kBcS && bcol < blk1 && bcol < nb && lane < ncol) {
            const int64_t j = bcol * bs + cstart + lane;
            if (j >= st.col_begin && j < st.col_end) {
                const int c = (int)color[j];
                const int q = (c == (int)(CT)(-1) || c < 0) ? -1 : c - c_lo;
                if (q >= 0 && q < B) { uq[k] = q; ud[k] = st.dest[j - st.col_begin]; }       // (else: another batch's colour)
            }
        }
    }
    int notp2 = 0;
    for (int q = threadIdx.x; q < B; q += kBcT) {
        const real_t e = eps[c_lo + q];
        ce[q] = e; cy[q] = (real_t)1 / e; cs[q] = sinh(e);
        notp2 |= div_is_pow2(e) ? 0 : 1;
    }
    // every step size of the batch a power of two (the complex step's default, eps(T)): imag * (1 / eps) is imag / eps, bit for bit --
    // no division at all.  Float32 only (whose quotients are true divisions: 39.3 -> 33.2 us at 10^4 blocks of 32 x 32, same box); for
    // Float64, where the quotient already is a product and two FMA corrections, dropping them measured no gain (52.2 against 51.5 us)
    const bool p2all = sizeof(real_t) == 4 && __syncthreads_or(notp2) == 0;
    if (sizeof(real_t) != 4) __syncthreads();

    if constexpr (PACK) {
        if (lbr <= kBcS + 2) {                         // (wave-uniform up to the two halves: the wavefronts 0 .. (kBcS + 1) / 2)
            const int at = (lbr - 1) * RS + (lane & 31);
            const real_t c0 = actr ? cos(xr) : (real_t)0;
            rx[at] = xr;
            rc[at] = c0;
            const real_t z = 0.0 * xr;                          // +-0 for a finite x: sinh(+-0) = +-0 (non-finite x: the library call)
            rz[at] = actr ? c0 * (z == (real_t)0 ? z : sinh(z)) : (real_t)0;
        }
    }
    // ---- phase A (as k_f_blockcoupled_lazy, MODE 2): sig of blocks g0-2 .. g0+kBcS+1 for every point of the batch
    real_t *tr = reinterpret_cast<real_t *>(tree) + (size_t)wave * 128;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int lb = wave + it * NW;
        if (lb >= NBLK) break;
        const int64_t bb = g0 - 2 + lb;
        const bool inb = (bb >= 0) & (bb < nb);
        const bool act = inb & (lane < bs);
        const real_t xj = xa[it];
        const int cj = ca[it];
        if (!PACK && lb >= 1 && lb <= kBcS + 2) {    // the rows phase B evaluates: x, cos once per row
            const int at = (lb - 1) * RS + lane;
            const real_t c0 = act ? cos(xj) : (real_t)0;
            if (lane < RS) {
            rx[at] = xj;
            rc[at] = c0;
            const real_t z = 0.0 * xj;                          // +-0 for a finite x: sinh(+-0) = +-0 (non-finite x: the library call)
            rz[at] = act ? c0 * (z == (real_t)0 ? z : sinh(z)) : (real_t)0;
            }
        }
        T *sg = sig + (size_t)lb * PB;
        // the base tree holds REAL parts only: the imaginary part of every base leaf is 0.0 + w * 0.0 = +0 and so is every partial
        // sum of them, and v + (+0) = v for the v = 0.0 + w * eps a perturbed lane starts from -- the bits of the complex tree
        real_t acc = act ? (real_t)0 + w * xj : (real_t)0;
        bc_wave_sync();                              // (the previous block's tree is no longer read)
        tr[lane] = acc;
        int base_at = 64;
        for (int off = 32; off > 0; off >>= 1) {
            acc = acc + __shfl_down(acc, off, 64);
            if (lane < off && off > 1) tr[base_at + lane] = acc;
            base_at += off;
        }
        const T root{__shfl(acc, 0, 64), (real_t)0};
        for (int q = lane; q < B; q += 64) own[q] = -1;
        bc_wave_sync();
        if (cj >= 0) own[cj] = lane;
        bc_wave_sync();
        const bool dup = (cj >= 0) && (own[cj] != lane);
        const bool any_dup = __builtin_amdgcn_ballot_w64(dup) != 0;
        if (!any_dup) {
            for (int q = lane; q < PB - 1; q += 64) sg[q] = inb ? root : zero_of<T>();
            bc_wave_sync();
            if (cj >= 0) {
                const real_t e = ce[cj];
                real_t n = (real_t)0 + w * xj;
                int idx = lane;
                n = n + tr[idx ^ 32]; idx &= 31;
                n = n + tr[64 + (idx ^ 16)]; idx &= 15;
                n = n + tr[96 + (idx ^ 8)]; idx &= 7;
                n = n + tr[112 + (idx ^ 4)]; idx &= 3;
                n = n + tr[120 + (idx ^ 2)]; idx &= 1;
                n = n + tr[124 + (idx ^ 1)];
                sg[cj] = T{n, (real_t)0 + w * e};
            }
        } else {
            for (int q = 0; q < PB - 1; ++q) {
                T term = zero_of<T>();
                if (act) {
                    const real_t d = (cj == q) ? ce[q] : 0.0;
                    term = zero_of<T>() + w * bc_make(xj, d, 0, T{});
                }
                term = tree_sum64<T>(term);
                if (lane == 0) sg[q] = inb ? term : zero_of<T>();
            }
        }
    }
    __syncthreads();

    // S of every (row block, point): (sig[k-1] + sig[k]) + sig[k+1], the sum k_f_blockcoupled_lazy's phase B forms, once per group
    for (int idx = threadIdx.x; idx < (kBcS + 2) * B; idx += kBcT) {
        const int lbk = 1 + idx / B, q = idx - (lbk - 1) * B;
        const T *sm = sig + (size_t)(lbk - 1) * PB + q;
        ssum[idx] = (sm[0] + sm[PB]) + sm[2 * PB];
    }
    __syncthreads();

    // ---- phase B, column-centric.  A wave takes a unit = half of the columns of one block-column; a lane is a ROW of the block (and,
    // TWO: blocks of <= 32 rows, one of two adjacent columns): it keeps x, cos(x) sinh(0 x) of its row in the three row blocks
    // b-1, b, b+1 in registers across the unit's columns and per column evaluates its three entries at the column's own point and
    // stores them (32 lanes = 256 contiguous bytes per row block; a column is one contiguous run).
    real_t *outp = (real_t *)st.out;
    if constexpr (QUAD) {
        // RPL rows per lane: a row pair of one of FOUR columns (Float64) or a row quad of one of EIGHT columns (Float32) -- 16 bytes per store
        constexpr int RPL = sizeof(real_t) == 4 ? FD_BC_RPL32 : 2, LPC = 32 / RPL, CPI = 64 / LPC;
        const int cq = lane / LPC, r2 = lane % LPC;                       // column of the iteration's CPI, rows RPL r2 .. RPL r2 + RPL - 1
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int u = wave + k * NW;
            if (u >= 2 * kBcS) break;
            const int lb = 2 + (u >> 1), cstart = (u & 1) * ucols, ncol = min(bs, cstart + ucols) - cstart;
            const int64_t bcol = g0 - 2 + lb;
            if (bcol >= blk1 || bcol >= nb || ncol <= 0) continue;
            const int q_l = uq[k];
            const long long dest_l = ud[k];
            real_t xk[3][RPL], rzk[3][RPL], rcm[RPL];
            bool mv[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int64_t kb = bcol - 1 + m;
                mv[m] = (kb >= 0) & (kb < nb);
                const int at = (lb - 2 + m) * RS + RPL * r2;
#pragma unroll
                for (int h = 0; h < RPL; ++h) { xk[m][h] = rx[at + h]; rzk[m][h] = rz[at + h]; }
            }
#pragma unroll
            for (int h = 0; h < RPL; ++h) rcm[h] = rc[(lb - 1) * RS + RPL * r2 + h];
            const int first = bcol > 0 ? 0 : 1;
            const T *su = ssum + (size_t)(lb - 2) * B;
            for (int it = 0; CPI * it < ncol; ++it) {
                const int ci = CPI * it + cq;
                const int src = ci < ncol ? ci : 0;
                const int qs = __shfl(q_l, src, 64);
                const unsigned dlo = (unsigned)__shfl((int)(unsigned)((unsigned long long)dest_l & 0xffffffffu), src, 64);
                const unsigned dhi = (unsigned)__shfl((int)(unsigned)((unsigned long long)dest_l >> 32), src, 64);
                const bool cv = (ci < ncol) & (qs >= 0);
                const int q = cv ? qs : 0;
                real_t *dst = outp + (long long)(((unsigned long long)dhi << 32) | dlo) + RPL * r2;
                const real_t e = ce[q], ye = cy[q], sh = cs[q];
                const int jl = cstart + ci;
                real_t vim[3][RPL], qv[3][RPL];
                bool fast = p2all || (sizeof(real_t) == 8 && div_shared_ok(e));
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const T S = su[(size_t)m * B + q];
#pragma unroll
                    for (int h = 0; h < RPL; ++h) {
                        const bool hit = (m == 1) & (RPL * r2 + h == jl);
                        const real_t xim = hit ? e : (real_t)0;
                        const real_t snim = hit ? rcm[h] * sh : rzk[m][h];
                        vim[m][h] = (xk[m][h] * S.im + xim * S.re) + snim;
                        qv[m][h] = vim[m][h] * ye;
                        if (!p2all) {
                            const real_t ma = fabs(vim[m][h]);        // (the divisor's range is tested once per column: div_shared_ok)
                            fast = fast & (!mv[m] | ((ma >= (real_t)0x1p-800) & (ma <= (real_t)0x1p800)));
                        }
                    }
                }
                if (p2all) {
                    // (qv is the quotient already)
                } else if (fast) {
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int h = 0; h < RPL; ++h) {
                            const real_t r0 = __builtin_fma(-e, qv[m][h], vim[m][h]);
                            const real_t q1 = __builtin_fma(r0, ye, qv[m][h]);
                            const real_t r1 = __builtin_fma(-e, q1, vim[m][h]);
                            qv[m][h] = __builtin_fma(r1, ye, q1);
                        }
                } else {
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int h = 0; h < RPL; ++h) qv[m][h] = vim[m][h] / e;
                }
                // (Float32: a destination that is even but not a multiple of four -- pairs there)
                const bool al = RPL == 2 || (dlo & 3u) == 0;
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    if (cv & mv[m]) {
                        bc_store_rows<RPL>(dst + (m - first) * bs, qv[m], al);
                    }
            }
        }
        return;
    }
    const int half = TWO ? lane >> 5 : 0, r = TWO ? lane & 31 : lane;
    const bool ract = r < bs;
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int u = wave + k * NW;
        if (u >= 2 * kBcS) break;
        const int lb = 2 + (u >> 1), cstart = (u & 1) * ucols, ncol = min(bs, cstart + ucols) - cstart;
        const int64_t bcol = g0 - 2 + lb;
        if (bcol >= blk1 || bcol >= nb || ncol <= 0) continue;
        const int q_l = uq[k];
        const long long dest_l = ud[k];
        // the lane's rows: block bcol - 1 + m, row r of it (block bandwidths (1, 1), equal blocks)
        real_t xk[3], rzk[3];
        bool mv[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int64_t kb = bcol - 1 + m;
            mv[m] = ract & (kb >= 0) & (kb < nb);
            const int at = (lb - 2 + m) * RS + r;
            xk[m] = rx[at]; rzk[m] = rz[at];
        }
        const real_t rcm = rc[(lb - 1) * RS + r];
        const int first = bcol > 0 ? 0 : 1;                               // the column's stored rows start at block max(bcol - 1, 0)
        const T *su = ssum + (size_t)(lb - 2) * B;
        for (int it = 0; it * (TWO ? 2 : 1) < ncol; ++it) {
            const int ci = TWO ? 2 * it + half : it;                      // column of the unit
            const int src = ci < ncol ? ci : 0;
            const int qs = __shfl(q_l, src, 64);
            const unsigned dlo = (unsigned)__shfl((int)(unsigned)((unsigned long long)dest_l & 0xffffffffu), src, 64);
            const unsigned dhi = (unsigned)__shfl((int)(unsigned)((unsigned long long)dest_l >> 32), src, 64);
            const bool cv = (ci < ncol) & (qs >= 0);
            const int q = cv ? qs : 0;
            real_t *dst = outp + (long long)(((unsigned long long)dhi << 32) | dlo) + r;
            const real_t e = ce[q], ye = cy[q], sh = cs[q];
            const int jl = cstart + ci;                                   // the column's own row is row jl of the middle block
            // the three entries of the lane as independent chains (nothing but the stores is predicated)
            real_t vim[3], qv[3];
            bool fast = p2all || (sizeof(real_t) == 8 && div_shared_ok(e));
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const T S = su[(size_t)m * B + q];
                const bool hit = (m == 1) & (r == jl);
                // imag(x~ * S + sin(x~)), x~ = (xk, hit ? e : 0):  (x~.re * S.im + x~.im * S.re) + cos(xk) * sinh(x~.im)
                const real_t xim = hit ? e : (real_t)0;
                const real_t snim = hit ? rcm * sh : rzk[m];
                vim[m] = (xk[m] * S.im + xim * S.re) + snim;
                qv[m] = vim[m] * ye;
                if (!p2all) {
                    const real_t ma = fabs(vim[m]);
                    fast = fast & (!mv[m] | ((ma >= (real_t)0x1p-800) & (ma <= (real_t)0x1p800)));
                }
            }
            // vim / e: div_shared's correctly rounded quotient (two FMA corrections of vim * (1 / e)); true division out of its range
            if (p2all) {
                // (a power-of-two step: qv is the quotient already)
            } else if (fast) {
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const real_t r0 = __builtin_fma(-e, qv[m], vim[m]);
                    const real_t q1 = __builtin_fma(r0, ye, qv[m]);
                    const real_t r1 = __builtin_fma(-e, q1, vim[m]);
                    qv[m] = __builtin_fma(r1, ye, q1);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 3; ++m) qv[m] = vim[m] / e;
            }
#pragma unroll
            for (int m = 0; m < 3; ++m)
                if (cv & mv[m]) __builtin_nontemporal_store(qv[m], dst + (m - first) * bs);
        }
    }
}

static size_t bcs_lds_bytes(int ncolors, int bs)
{
    const size_t rs = bs <= 32 ? 32 : 64;
    return ((size_t)(kBcS + 4) * (size_t)(ncolors + 1) + bcs_shared_items(ncolors)) * 2 * sizeof(real_t) +
           (size_t)3 * (kBcS + 2) * rs * sizeof(real_t) + (size_t)(3 * ncolors + 2) * sizeof(real_t);
}

// LDS of k_f_blockcoupled_lazy: sigma of every point + the base summation tree, per block of the group, + owners
static size_t bc_lds_bytes(int ncolors, int pts, bool cplx)
{
    const size_t el = cplx ? 2 * sizeof(real_t) : sizeof(real_t);
    return (size_t)(kBcG + 2) * ((size_t)(ncolors * pts + 1) + 128) * el + (size_t)(kBlock / 64) * (size_t)ncolors * 4;
}

template <typename CT>
static int lazy_blockcoupled_launch(BuiltinF *b, void *fx, const fd_lazy_points *lp, int64_t fs, int64_t r0, int64_t r1,
                                    hipStream_t s)
{
    const int64_t nb = b->prm[0], bs = b->prm[1];
    const int mode = lp->is_complex ? 2 : (lp->pts == 2 ? 1 : 0);
    if (lp->store) {
        // the launch stores imag(f) / eps into the block-banded data itself (fd_colrange_store): complex step, the block structure of
        // this fixture (dense blocks of bs <= 64 rows, block bandwidths (1, 1))
        if (lp->store_kind != FD_STORE_COLRANGE || mode != 2) return FD_LAZY_DECLINED;
        const fd_colrange_store st = *(const fd_colrange_store *)lp->store;
        if (st.elem_bytes != (int)sizeof(real_t) || st.nblk != nb || st.block_size != bs || st.bl != 1 || st.bu != 1 || bs > 64 ||
            st.color_bytes != (int)sizeof(CT) || st.col_end <= st.col_begin || bcs_lds_bytes(lp->ncolors, (int)bs) > (size_t)64 * 1024)
            return FD_LAZY_DECLINED;
        const int64_t cb0 = st.col_begin / bs, cb1 = (st.col_end - 1) / bs + 1;      // blocks with local columns
        const int64_t gs = (cb1 - cb0 + kBcS - 1) / kBcS;
                if (bs == 32 && st.pairs && (((uintptr_t)st.out) & (sizeof(real_t) == 4 ? 4 * FD_BC_RPL32 - 1 : 15)) == 0)      // row pairs (or Float32 row quads)
            hipLaunchKernelGGL((k_f_blockcoupled_store<CT, true, true>), dim3((unsigned)gs), dim3(kBcT), bcs_lds_bytes(lp->ncolors, (int)bs), s, (const real_t *)lp->x,
                               (const CT *)lp->color, (const real_t *)lp->eps, lp->c_lo, lp->ncolors, nb, (int)bs, cb0, cb1, st);
        else if (bs <= 32)
            hipLaunchKernelGGL((k_f_blockcoupled_store<CT, true>), dim3((unsigned)gs), dim3(kBcT), bcs_lds_bytes(lp->ncolors, (int)bs), s, (const real_t *)lp->x,
                               (const CT *)lp->color, (const real_t *)lp->eps, lp->c_lo, lp->ncolors, nb, (int)bs, cb0, cb1, st);
        else
            hipLaunchKernelGGL((k_f_blockcoupled_store<CT, false>), dim3((unsigned)gs), dim3(kBcT), bcs_lds_bytes(lp->ncolors, (int)bs), s, (const real_t *)lp->x,
                               (const CT *)lp->color, (const real_t *)lp->eps, lp->c_lo, lp->ncolors, nb, (int)bs, cb0, cb1, st);
        return hipGetLastError() == hipSuccess ? 0 : 4;
    }
    const int64_t blk0 = r0 / bs, blk1 = (r1 - 1) / bs + 1;
    const int64_t g = (blk1 - blk0 + kBcG - 1) / kBcG;
    const size_t shm = bc_lds_bytes(lp->ncolors, lp->pts, mode == 2);
#define FD_LAZY(MODE)                                                                                               \
    hipLaunchKernelGGL((k_f_blockcoupled_lazy<CT, MODE>), dim3((unsigned)g), dim3(kBlock), shm, s, (real_t *)fx, fs, \
                       (real_t *)lp->base_out, (const real_t *)lp->x, (const CT *)lp->color, (const real_t *)lp->eps, lp->c_lo,     \
                       lp->ncolors, nb, (int)bs, blk0, blk1, r0, r1, lp->imag_only)
    if (mode == 0) FD_LAZY(0); else if (mode == 1) FD_LAZY(1); else FD_LAZY(2);
#undef FD_LAZY
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// ---- lazy-point launchers of a JVP (fd_f_launch_lazy_jvp): f at x + eps*v / x - eps*v, perturbed while loading ----
// Points are formed as x[j] + (eps*v[j]) and x[j] - (eps*v[j]) exactly as k_jvp_points writes them, rows are evaluated
// with the same helpers as the plain kernels => bit-identical to the materialised path.
template <bool NL>
__global__ void __launch_bounds__(kBlock)
k_f_tridiag_lazy_jvp(real_t *__restrict__ fx, int64_t fs, real_t *__restrict__ base_out, const real_t *__restrict__ x,
                     const real_t *__restrict__ v, const real_t *__restrict__ eps, int central, int64_t n,
                     real_t *__restrict__ qout)
{
    const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 2;
    if (i >= n) return;
    const real_t e = eps[0];
    real_t xv[4], ev[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t j = i - 1 + k;
        const bool in = (j >= 0) & (j < n);
        xv[k] = in ? x[in ? j : 0] : 0.0;
        ev[k] = in ? e * v[in ? j : 0] : 0.0;
    }
    const bool two = i + 1 < n;
    if (qout) {   // fd_lazy_jvp_points.quotient_out: (f(x + eps v) - f(x)) / eps  or  (f(x + eps v) - f(x - eps v)) / (2 eps)
        real_t pp[4], pm[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { pp[k] = xv[k] + ev[k]; pm[k] = central ? xv[k] - ev[k] : xv[k]; }
        const real_t ed = central ? 2 * e : e;
        const real_t q0 = sub_exact(tridiag_row<real_t, NL>(pp[0], pp[1], pp[2]), tridiag_row<real_t, NL>(pm[0], pm[1], pm[2])) / ed;
        if (two) *reinterpret_cast<r2_t *>(qout + i) =
                     r2_t{q0, sub_exact(tridiag_row<real_t, NL>(pp[1], pp[2], pp[3]), tridiag_row<real_t, NL>(pm[1], pm[2], pm[3])) / ed};
        else qout[i] = q0;
        return;
    }
    if (base_out) {
        const real_t b0 = tridiag_row<real_t, NL>(xv[0], xv[1], xv[2]);
        if (two) *reinterpret_cast<r2_t *>(base_out + i) = r2_t{b0, tridiag_row<real_t, NL>(xv[1], xv[2], xv[3])};
        else base_out[i] = b0;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (m == 1 && !central) break;
        // forward: member 0 = x + eps v; central: member 0 = x - eps v, member 1 = x + eps v
        const bool minus = central && m == 0;
        real_t p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = minus ? xv[k] - ev[k] : xv[k] + ev[k];
        real_t *dst = fx + (int64_t)m * fs + i;
        const real_t v0 = tridiag_row<real_t, NL>(p[0], p[1], p[2]);
        if (two) *reinterpret_cast<r2_t *>(dst) = r2_t{v0, tridiag_row<real_t, NL>(p[1], p[2], p[3])};
        else dst[0] = v0;
    }
}

template <int SK>
__global__ void __launch_bounds__(kBlock)
k_f_stencil5_lazy_jvp(real_t *__restrict__ fx, int64_t fs, real_t *__restrict__ base_out, const real_t *__restrict__ x,
                      const real_t *__restrict__ v, const real_t *__restrict__ eps, int central, int64_t nx, int64_t ny,
                      real_t *__restrict__ qout)
{
    const int64_t n = nx * ny;
    const int64_t ntiles = (n + 2 * kBlock - 1) / (2 * kBlock);
    const int64_t tile = xcd_tile(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const int64_t k = tile * (2 * kBlock) + threadIdx.x * 2;
    if (k >= n) return;
    const int64_t j = k / nx, i = k - j * nx;
    const bool hs = j > 0, hn = j + 1 < ny, hw = i > 0, he = i + 2 < nx;
    const int64_t idx[8] = {k, k + 1, k - nx, k - nx + 1, k + nx, k + nx + 1, k - 1, k + 2};
    const bool ok[8] = {true, true, hs, hs, hn, hn, hw, he};
    const real_t e = eps[0];
    real_t xv[8], ev[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int64_t at = ok[m] ? idx[m] : k;
        xv[m] = x[at];
        ev[m] = e * v[at];
    }
    if (qout) {   // fd_lazy_jvp_points.quotient_out
        real_t pp[8], pm[8], a0, a1, s0, s1;
#pragma unroll
        for (int q = 0; q < 8; ++q) { pp[q] = xv[q] + ev[q]; pm[q] = central ? xv[q] - ev[q] : xv[q]; }
        stencil5_pair<real_t, SK>(pp, hs, hn, hw, he, a0, a1);
        stencil5_pair<real_t, SK>(pm, hs, hn, hw, he, s0, s1);
        const real_t ed = central ? 2 * e : e;
        *reinterpret_cast<r2_t *>(qout + k) = r2_t{sub_exact(a0, s0) / ed, sub_exact(a1, s1) / ed};
        return;
    }
    if (base_out) {
        real_t b0, b1;
        stencil5_pair<real_t, SK>(xv, hs, hn, hw, he, b0, b1);
        *reinterpret_cast<r2_t *>(base_out + k) = r2_t{b0, b1};
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (m == 1 && !central) break;
        const bool minus = central && m == 0;
        real_t p[8], o0, o1;
#pragma unroll
        for (int q = 0; q < 8; ++q) p[q] = minus ? xv[q] - ev[q] : xv[q] + ev[q];
        stencil5_pair<real_t, SK>(p, hs, hn, hw, he, o0, o1);
        *reinterpret_cast<r2_t *>(fx + (int64_t)m * fs + k) = r2_t{o0, o1};
    }
}

static bool has_lazy_jvp(const BuiltinF *b)
{
    if (b->family == FD_F_TRIDIAG || b->family == FD_F_TRIDIAG_NL) return true;
    return (b->family == FD_F_LAP5 || b->family == FD_F_CLAMP5 || b->family == FD_F_LAP5_NL) && (b->prm[0] % 2 == 0);  // pairs must not straddle grid rows
}

static int builtin_launch_lazy_jvp(void *fctx, void *fx, const fd_lazy_jvp_points *lp, int64_t fx_stride, void *stream)
{
    BuiltinF *b = (BuiltinF *)fctx;
    if (!b || b->magic != 0xFD0F00D5u || !lp) return 1;
    if (!has_lazy_jvp(b)) return FD_LAZY_DECLINED;
    if (((((uintptr_t)fx) | ((uintptr_t)lp->base_out) | ((uintptr_t)lp->quotient_out)) & kPairMask) != 0 || (fx_stride & 1)) return FD_LAZY_DECLINED;
    b->launches.fetch_add(1);
    b->points.fetch_add((lp->central ? 2 : 1) + ((lp->base_out || (lp->quotient_out && !lp->central)) ? 1 : 0));
    const hipStream_t s = (hipStream_t)stream;
    real_t *fxp = (real_t *)fx, *base = (real_t *)lp->base_out, *qout = (real_t *)lp->quotient_out;
    const real_t *x = (const real_t *)lp->x, *v = (const real_t *)lp->v, *eps = (const real_t *)lp->eps;
    if (b->family == FD_F_TRIDIAG || b->family == FD_F_TRIDIAG_NL) {
        const int64_t n = b->prm[0];
        const unsigned g = (unsigned)(((n + 1) / 2 + kBlock - 1) / kBlock);
        if (b->family == FD_F_TRIDIAG_NL)
            hipLaunchKernelGGL(k_f_tridiag_lazy_jvp<true>, dim3(g), dim3(kBlock), 0, s, fxp, fx_stride, base, x, v, eps, lp->central, n, qout);
        else
            hipLaunchKernelGGL(k_f_tridiag_lazy_jvp<false>, dim3(g), dim3(kBlock), 0, s, fxp, fx_stride, base, x, v, eps, lp->central, n, qout);
    } else {
        const int64_t n = b->prm[0] * b->prm[1];
        const unsigned g = (unsigned)(8 * xcd_chunks((n + 2 * kBlock - 1) / (2 * kBlock)));
#define FD_ST_JVP(SKK) hipLaunchKernelGGL(k_f_stencil5_lazy_jvp<SKK>, dim3(g), dim3(kBlock), 0, s, fxp, fx_stride, base, x, v, eps, \
                               lp->central, b->prm[0], b->prm[1], qout)
        if (b->family == FD_F_CLAMP5) FD_ST_JVP(1); else if (b->family == FD_F_LAP5_NL) FD_ST_JVP(2); else FD_ST_JVP(0);
#undef FD_ST_JVP
    }
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

static bool has_lazy(const BuiltinF *b)
{
    if (b->family == FD_F_TRIDIAG || b->family == FD_F_TRIDIAG_NL || b->family == FD_F_LAP7 || b->family == FD_F_SPARSE) return true;
    if (b->family == FD_F_BLOCKCOUPLED) return b->prm[1] <= 64;   // one wave per block
    return (b->family == FD_F_LAP5 || b->family == FD_F_CLAMP5 || b->family == FD_F_LAP5_NL) && (b->prm[0] % 2 == 0);  // pairs must not straddle grid rows
}

static int builtin_launch_lazy(void *fctx, void *fx, const fd_lazy_points *lp, int64_t fx_stride, int64_t row_begin,
                               int64_t row_end, void *stream)
{
    BuiltinF *b = (BuiltinF *)fctx;
    if (!b || b->magic != 0xFD0F00D5u || !lp) return 1;
    if (!has_lazy(b)) return 6;
    if (b->family == FD_F_LAP7 || b->family == FD_F_SPARSE) {      // functor families: the column-by-column store, or nothing
        const int rc = lp->color_bytes == 1 ? functor_family_lazy<uint8_t>(b, lp, (hipStream_t)stream) : functor_family_lazy<int32_t>(b, lp, (hipStream_t)stream);
        if (rc == 0) {
            b->launches.fetch_add(1);
            // (f(x) of a forward difference: one plain launch, counted there -- or, FD_LAZY_CAP_STORE_CSC_BASE without f_in, formed
            //  inside this launch: counted with the first colour chunk)
            const bool own_base = lp->pts == 1 && !lp->is_complex && lp->store && !((const fd_csc_store *)lp->store)->fx_base && lp->c_lo == 0;
            b->points.fetch_add((int64_t)lp->ncolors * lp->pts + (own_base ? 1 : 0));
        }
        return rc;
    }
    // 16-B vector accesses: bases are hipMalloc/torch allocations, fx_stride is a multiple of 32 elements
    if (((((uintptr_t)fx) | ((uintptr_t)lp->base_out)) & kPairMask) != 0 || (fx_stride & 1)) return 7;
    if (lp->diff && (b->family == FD_F_BLOCKCOUPLED || lp->is_complex || lp->base_out)) return 8;   // (not registered with FD_LAZY_CAP_DIFF)
    if (lp->store && !((lp->diff && (b->family == FD_F_TRIDIAG || b->family == FD_F_TRIDIAG_NL || b->family == FD_F_LAP5 || b->family == FD_F_LAP5_NL)) ||
                       (b->family == FD_F_BLOCKCOUPLED && lp->is_complex)))
        return 9;   // (FD_LAZY_CAP_STORE: the tridiagonal and Laplacian families; the block-coupled family in the complex step)
    const int64_t npts = (int64_t)lp->ncolors * lp->pts + ((lp->base_out || lp->diff == 2) ? 1 : 0);
    // the block-coupled kernel keeps one sigma per (block, point) in LDS: decline batches that would not fit
    if (b->family == FD_F_BLOCKCOUPLED && !lp->store && bc_lds_bytes(lp->ncolors, lp->pts, lp->is_complex != 0) > (size_t)56 * 1024)
        return FD_LAZY_DECLINED;
    const bool count_points = lp->nparts <= 1 || lp->part == 0;   // row strips: the parts together are ONE evaluation per point
    b->launches.fetch_add(1);
    if (count_points) b->points.fetch_add(npts);
    const int64_t r0 = std::max<int64_t>(row_begin, 0), r1 = std::min<int64_t>(row_end, b->M);
    if (r1 <= r0 || lp->ncolors <= 0) return 0;
    const hipStream_t s = (hipStream_t)stream;
    int rc;
    if (b->family == FD_F_BLOCKCOUPLED)
        rc = lp->color_bytes == 1 ? lazy_blockcoupled_launch<uint8_t>(b, fx, lp, fx_stride, r0, r1, s)
                                  : lazy_blockcoupled_launch<int32_t>(b, fx, lp, fx_stride, r0, r1, s);
    else if (b->family == FD_F_LAP5 || b->family == FD_F_CLAMP5 || b->family == FD_F_LAP5_NL)
        rc = lp->color_bytes == 1 ? lazy_stencil5_launch<uint8_t>(b, fx, lp, fx_stride, r0, r1, s)
                                  : lazy_stencil5_launch<int32_t>(b, fx, lp, fx_stride, r0, r1, s);
    else
        rc = lp->color_bytes == 1 ? lazy_tridiag_launch<uint8_t>(b, fx, lp, fx_stride, r0, r1, s)
                                  : lazy_tridiag_launch<int32_t>(b, fx, lp, fx_stride, r0, r1, s);
    if (rc == FD_LAZY_DECLINED) {   // nothing was enqueued: the library materialises / re-asks, and counts then
        b->launches.fetch_sub(1);
        if (count_points) b->points.fetch_sub(npts);
    }
    return rc;
}

}  // namespace fdjac

using namespace fdjac;

extern "C" {

int fd_builtin_f_create(fd_ctx *ctx, int family, const int64_t *params, int nparams, fd_f_launch *fn_out,
                        void **fctx_out)
{
    FD_REQUIRE(ctx && params && fn_out && fctx_out, FD_ERR_ARG, "NULL argument");
    BuiltinF *b = new (std::nothrow) BuiltinF();
    FD_REQUIRE(b != nullptr, FD_ERR_NOMEM, "out of host memory");
    b->ctx = ctx;
    b->family = family;
    int need = 1;
    switch (family) {
    case FD_F_TRIDIAG:
    case FD_F_TRIDIAG_NL: need = 1; break;
    case FD_F_LAP5:
    case FD_F_LAP5_NL:
    case FD_F_CLAMP5:
    case FD_F_BLOCKCOUPLED: need = 2; break;
    case FD_F_NONSQUARE: need = 1; break;
    case FD_F_LAP7: need = 3; break;
    default:
        delete b;
        set_error(family == FD_F_SPARSE ? "FD_F_SPARSE is created by fd_builtin_f_create_sparse" : "unknown built-in f family %d", family);
        return FD_ERR_ARG;
    }
    if (nparams < need) {
        delete b;
        set_error("family %d needs %d parameters", family, need);
        return FD_ERR_ARG;
    }
    for (int i = 0; i < need; ++i) b->prm[i] = params[i];
    for (int i = 0; i < need; ++i)
        if (b->prm[i] < 1) {
            delete b;
            set_error("parameters must be >= 1");
            return FD_ERR_ARG;
        }
    switch (family) {
    case FD_F_TRIDIAG:
    case FD_F_TRIDIAG_NL: b->M = b->N = b->prm[0]; break;
    case FD_F_NONSQUARE: b->M = b->prm[0]; b->N = 2 * b->prm[0]; break;
    case FD_F_LAP7:
        b->M = b->N = b->prm[0] * b->prm[1] * b->prm[2];
        if (b->prm[0] * b->prm[1] >= ((int64_t)1 << 31) || b->M >= ((int64_t)1 << 31)) {      // (32-bit grid arithmetic in the kernels)
            delete b;
            set_error("grid too large for the 7-point family");
            return FD_ERR_ARG;
        }
        break;
    default: b->M = b->N = b->prm[0] * b->prm[1]; break;
    }
    *fn_out = builtin_launch;
    *fctx_out = b;
    return FD_OK;
}

// FD_F_SPARSE: the pattern arrives as the CSC pattern of the Jacobian; the launcher keeps its transpose (rows, ascending columns)
int fd_builtin_f_create_sparse(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr, const void *rowval, int idx_bytes, int idx_base,
                               fd_f_launch *fn_out, void **fctx_out)
{
    FD_REQUIRE(ctx && colptr && rowval && fn_out && fctx_out, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(idx_bytes == 4 || idx_bytes == 8, FD_ERR_ARG, "idx_bytes must be 4 or 8");
    FD_REQUIRE(idx_base == 0 || idx_base == 1, FD_ERR_ARG, "idx_base must be 0 or 1");
    FD_REQUIRE(M >= 1 && N >= 1 && M < ((int64_t)1 << 31) && N < ((int64_t)1 << 31), FD_ERR_ARG, "bad shape");
    auto ld = [&](const void *p, int64_t i) { return idx_bytes == 8 ? ((const int64_t *)p)[i] : (int64_t)((const int32_t *)p)[i]; };
    const int64_t nnz = ld(colptr, N) - idx_base;
    FD_REQUIRE(nnz >= 0 && nnz < ((int64_t)1 << 31), FD_ERR_SHAPE, "colptr is not monotone / too many entries");
    std::vector<int32_t> srow((size_t)M + 1, 0), scol((size_t)std::max<int64_t>(nnz, 1)), sdest((size_t)std::max<int64_t>(nnz, 1)), hcp((size_t)N + 1);
    for (int64_t j = 0; j < N; ++j) {
        const int64_t a = ld(colptr, j) - idx_base, b2 = ld(colptr, j + 1) - idx_base;
        FD_REQUIRE(a >= 0 && a <= b2 && b2 <= nnz, FD_ERR_SHAPE, "colptr is not monotone at column %lld", (long long)j);
        hcp[(size_t)j] = (int32_t)a;
        hcp[(size_t)j + 1] = (int32_t)b2;
        for (int64_t q = a; q < b2; ++q) {
            const int64_t r = ld(rowval, q) - idx_base;
            FD_REQUIRE(r >= 0 && r < M, FD_ERR_SHAPE, "rowval[%lld] outside the matrix", (long long)q);
            ++srow[(size_t)r + 1];
        }
    }
    for (int64_t r = 0; r < M; ++r) srow[(size_t)r + 1] += srow[(size_t)r];
    {
        std::vector<int32_t> cur(srow.begin(), srow.end() - 1);
        for (int64_t j = 0; j < N; ++j)       // columns ascending: every row's entries end up in ascending column order
            for (int64_t q = ld(colptr, j) - idx_base; q < ld(colptr, j + 1) - idx_base; ++q) {
                const size_t at = (size_t)cur[(size_t)(ld(rowval, q) - idx_base)]++;
                scol[at] = (int32_t)j;
                sdest[at] = (int32_t)q;
            }
    }
    BuiltinF *b = new (std::nothrow) BuiltinF();
    FD_REQUIRE(b != nullptr, FD_ERR_NOMEM, "out of host memory");
    b->ctx = ctx;
    b->family = FD_F_SPARSE;
    b->M = M;
    b->N = N;
    b->prm[0] = M; b->prm[1] = N; b->prm[2] = nnz;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_srow, sizeof(int32_t) * srow.size());
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_scol, sizeof(int32_t) * scol.size());
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_sdest, sizeof(int32_t) * sdest.size());
    if (e == hipSuccess) e = hipMemcpy(b->d_sdest, sdest.data(), sizeof(int32_t) * sdest.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_srow, srow.data(), sizeof(int32_t) * srow.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_scol, scol.data(), sizeof(int32_t) * scol.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("uploading the pattern of the sparse family failed: %s", hipGetErrorString(e));
        if (b->d_srow) (void)hipFree(b->d_srow);
        if (b->d_scol) (void)hipFree(b->d_scol);
        if (b->d_sdest) (void)hipFree(b->d_sdest);
        delete b;
        return FD_ERR_HIP;
    }
    b->h_colptr.swap(hcp);
    *fn_out = builtin_launch;
    *fctx_out = b;
    return FD_OK;
}

int fd_builtin_f_destroy(void *fctx)
{
    BuiltinF *b = (BuiltinF *)fctx;
    if (!b) return FD_OK;
    FD_REQUIRE(b->magic == 0xFD0F00D5u, FD_ERR_ARG, "not a built-in f context");
    if (b->d_sig) (void)hipFree(b->d_sig);
    if (b->d_srow) (void)hipFree(b->d_srow);
    if (b->d_scol) (void)hipFree(b->d_scol);
    if (b->d_sdest) (void)hipFree(b->d_sdest);
    for (auto &kv : b->rows_memo) {
        if (kv.second.pending) (void)hipEventSynchronize(kv.second.ev);      // (a copy into the pinned word may still be in flight)
        if (kv.second.ev) (void)hipEventDestroy(kv.second.ev);
        if (kv.second.h_note) (void)hipHostFree(kv.second.h_note);
    }
    b->magic = 0;
    delete b;
    return FD_OK;
}

int fd_builtin_f_lazy(void *fctx, fd_f_launch_lazy *fn_out)
{
    BuiltinF *b = (BuiltinF *)fctx;
    FD_REQUIRE(b && b->magic == 0xFD0F00D5u && fn_out, FD_ERR_ARG, "not a built-in f context");
    if (!has_lazy(b)) {
        set_error("family %d has no lazy-point launcher (5-point stencils need an even nx)", b->family);
        return FD_ERR_UNSUPPORTED;
    }
    *fn_out = builtin_launch_lazy;
    return FD_OK;
}

int fd_builtin_f_lazy_jvp(void *fctx, fd_f_launch_lazy_jvp *fn_out)
{
    BuiltinF *b = (BuiltinF *)fctx;
    FD_REQUIRE(b && b->magic == 0xFD0F00D5u && fn_out, FD_ERR_ARG, "not a built-in f context");
    if (!has_lazy_jvp(b)) {
        set_error("family %d has no lazy JVP launcher (5-point stencils need an even nx)", b->family);
        return FD_ERR_UNSUPPORTED;
    }
    *fn_out = builtin_launch_lazy_jvp;
    return FD_OK;
}

int fd_builtin_f_lazy_jvp_caps(void *fctx, int *caps_out)
{
    BuiltinF *b = (BuiltinF *)fctx;
    FD_REQUIRE(b && b->magic == 0xFD0F00D5u && caps_out, FD_ERR_ARG, "bad argument");
    *caps_out = has_lazy_jvp(b) ? FD_LAZY_JVP_CAP_QUOTIENT : 0;
    return FD_OK;
}

int fd_builtin_f_lazy_caps(void *fctx, int *caps_out)
{
    BuiltinF *b = (BuiltinF *)fctx;
    FD_REQUIRE(b && b->magic == 0xFD0F00D5u && caps_out, FD_ERR_ARG, "not a built-in f context");
    // the tridiagonal and 5-point kernels write exactly the (pair-rounded) row window they are handed; the block-coupled
    // kernel writes whole blocks, so it does not claim FD_LAZY_CAP_ROW_WINDOW
    if (b->family == FD_F_LAP7 || b->family == FD_F_SPARSE) {      // functor families: the column-by-column store only
        // (a 7-point row costs less than the gather of its f(x), and the sparse family's row-wise store has f(x) of its rows anyway: both
        //  evaluate the unperturbed rows inside the storing launch)
        *caps_out = FD_LAZY_CAP_STORE_CSC | FD_LAZY_CAP_STORE_CSC_COMPLEX | FD_LAZY_CAP_STORE_CSC_BASE;
        return FD_OK;
    }
    *caps_out = has_lazy(b) ? (FD_LAZY_CAP_IMAG_ONLY | (b->family == FD_F_BLOCKCOUPLED ? 0 : (FD_LAZY_CAP_ROW_WINDOW | FD_LAZY_CAP_DIFF)) |
                               ((b->family == FD_F_TRIDIAG || b->family == FD_F_TRIDIAG_NL || b->family == FD_F_LAP5 || b->family == FD_F_LAP5_NL ||
                                 b->family == FD_F_BLOCKCOUPLED) ? FD_LAZY_CAP_STORE : 0) |
                               ((b->family == FD_F_TRIDIAG || b->family == FD_F_TRIDIAG_NL) ? FD_LAZY_CAP_FUSED_EPS : 0)) : 0;
    return FD_OK;
}

int fd_builtin_f_counts(void *fctx, int64_t *launches, int64_t *points)
{
    BuiltinF *b = (BuiltinF *)fctx;
    FD_REQUIRE(b && b->magic == 0xFD0F00D5u, FD_ERR_ARG, "not a built-in f context");
    if (launches) *launches = b->launches.load();
    if (points) *points = b->points.load();
    return FD_OK;
}

int fd_builtin_f_info(void *fctx, int key, int64_t *value)
{
    BuiltinF *b = (BuiltinF *)fctx;
    FD_REQUIRE(b && b->magic == 0xFD0F00D5u && value, FD_ERR_ARG, "not a built-in f context");
    FD_REQUIRE(key == FD_F_INFO_ROW_STORES, FD_ERR_ARG, "unknown key %d", key);
    *value = b->row_stores.load();
    return FD_OK;
}

}  // extern "C"
