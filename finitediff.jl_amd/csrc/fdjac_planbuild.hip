// Plan construction ON THE DEVICE: common-pattern CSC plans (fd_plan_create_csc / fd_plan_create_csc_device), BandedMatrix
// plans (fd_plan_create_banded) and the colours of Tridiagonal plans (fd_plan_create_tridiagonal).
//
// What a plan compiles is the per-call pattern work of the reference -- the O(nnz) pattern comparison, the O(N) colour
// scan per colour and the colptr walk of `_colorediteration!` (src/jacobians.jl:524-535,547;
// ext/FiniteDiffSparseArraysExt.jl:38-47,51-52; ext/FiniteDiffBandedMatricesExt.jl:13-27) -- into tile descriptors and
// 16-bit entry codes.  The host builders (fdjac_api.hip: csc_common / try_window_plan / try_window2d_plan) do that with
// serial loops over nnz (0.3 - 0.7 s at nnz = 3..5 * 10^7).  Here the same arrays are produced by kernels:
//   k_pb_colmax / k_pb_colors   colorvec (Int32 / Int64, 1-based) -> 0-based uint8 colours, C = maximum(colorvec),
//                               "some column has no colour", "colours are cyclic" (wave ballots)
//   k_pb_tiles                  1-D tiles: one workgroup per tile of stored entries, straight from colptr / rowval (or, BAND,
//                               from the implicit indices of a band's column-major storage): the tile's slice of colptr in
//                               LDS, every entry finds its column by binary search, down-converts its row (the index
//                               down-conversion SURVEY section 7 asks for) and takes its column's colour; row / colour
//                               extent by wave reductions -> tile descriptor; the 16-bit entry codes
//   k_pb_periodic               which tiles repeat their codes with the plan-wide period (ballot over the tile)
//   k_pb_band_check             uniform-band test (affine colptr, consecutive rows) -> the tiles whose descriptors the
//                               row-window kernel computes instead of loading
//   k_pb_sample_expand / k_pb_coherence   the gather-coherence estimate that decides "scattered" (bitonic sort of each
//                               sampled tile in LDS, distinct 128-B lines per wave gather by lane-to-lane comparison)
//   k_pb2_sample_cols / k_pb2_check / k_pb2_count / k_pb2_tiles   2-D (strided) tiles of 2-D stencil patterns: the stride, its
//                               validation, tile placement (host prefix sum over 8 B per tile), row windows from an LDS
//                               bitmap of the tile's rows, descriptors and codes
// No intermediate per-entry arrays, no same-address atomics in the hot kernels (grid-wide statistics are computed by the
// host from the 48-byte tile descriptors).
// The pattern may already live on the device (fd_plan_create_csc_device: nothing crosses PCIe) or is uploaded raw.
// Patterns the device builder does not handle (more than 8 colours; tiles that need clustered / sorted row windows; 2-D grids
// narrower than 64; forced kernel variants) make it step aside -- the host builder then runs as before.  The host builders
// are also the CHECKERS: they decide in the same order (one-window 1-D tiles, 2-D tiles, clustered windows), and the tests build
// every plan both ways (FDJAC_PLAN_DEVICE=0/1) and compare the plan arrays bit for bit (fd_plan_checksum), fixed and
// randomised patterns alike.
#include <time.h>

#include <algorithm>
#include <limits>
#include <vector>

#include "fdjac_internal.h"

namespace fdjac {

struct PbStats {
    unsigned long long max_color;     // maximum(colorvec)
    long long first_color;            // colorvec[0]
    unsigned int flags;               // PB_* below
    int row_min, row_max;             // over the local stored entries
    int max_slots, max_ncol;          // over the tiles
    unsigned long long elems;         // sum over tiles of 2 * pairs * ncol
    unsigned int regular;             // tiles with periodic codes
    unsigned int pad;
};
enum {
    PB_NONE = 1,          // some column has no colour
    PB_NOT_CYCLIC = 2,    // colorvec is not (j + shift) mod C
    PB_BAD_ROW = 4,       // rowval outside 1..M
    PB_BAD_COLPTR = 8,    // colptr not monotone / outside the slice
    PB_COLOR_BIG = 16,    // a colour does not fit the device representation
    PB_NEED_SORT = 32,    // a tile needs several row windows (scattered pattern): host builder
    PB_TOO_MANY_COL = 64, // more than kWinMaxCol colours in one tile
    PB_TOO_WIDE = 128     // more than 2048 window rows in one tile
};

__device__ __forceinline__ int64_t pb_load(const void *p, int bytes, int64_t i)
{
    return bytes == 8 ? ((const int64_t *)p)[i] : (int64_t)((const int32_t *)p)[i];
}

__global__ void __launch_bounds__(kBlock) k_pb_colmax(const void *__restrict__ colorvec, int cb, int64_t N, PbStats *st)
{
    long long m = 0;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < N; j += (int64_t)gridDim.x * kBlock) {
        const long long c = pb_load(colorvec, cb, j);
        m = c > m ? c : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const long long o = __shfl_down(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(&st->max_color, (unsigned long long)m);
    if (blockIdx.x == 0 && threadIdx.x == 0) st->first_color = N > 0 ? pb_load(colorvec, cb, 0) : 0;
}

// 0-based uint8 colours (0xFF = no colour); cyclic test against (j + shift) mod C
__global__ void __launch_bounds__(kBlock) k_pb_colors(const void *__restrict__ colorvec, int cb, int64_t N, int C, int shift,
                                                      uint8_t *__restrict__ color8, PbStats *st)
{
    unsigned flags = 0;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < N; j += (int64_t)gridDim.x * kBlock) {
        const long long c = pb_load(colorvec, cb, j);
        if (c > 253) flags |= PB_COLOR_BIG;
        const int c0 = c >= 1 ? (int)(c - 1) : -1;
        color8[j] = c0 < 0 ? (uint8_t)0xFF : (uint8_t)c0;
        if (c0 < 0) flags |= PB_NONE;
        if (C <= 0 || c0 != (int)((j + shift) % C)) flags |= PB_NOT_CYCLIC;
    }
    // one atomic per wave: OR of the lanes' flags by ballots
    unsigned wf = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b)
        if (__builtin_amdgcn_ballot_w64((flags >> b) & 1)) wf |= 1u << b;
    if ((threadIdx.x & 63) == 0 && wf) atomicOr(&st->flags, wf);
}

// colptr must be monotone and stay inside the local entry range [e0, e1): the tile kernel's searches rely on it
__global__ void __launch_bounds__(kBlock) k_pb_check_colptr(const void *__restrict__ colptr, int ib, int base, int64_t col0, int64_t col1,
                                                            int64_t e0, int64_t e1, PbStats *st)
{
    bool bad = false;
    for (int64_t j = col0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < col1; j += (int64_t)gridDim.x * kBlock) {
        const int64_t a = pb_load(colptr, ib, j) - base, b = pb_load(colptr, ib, j + 1) - base;
        bad = bad || !(a <= b && a >= e0 && b <= e1);
    }
    if (__builtin_amdgcn_ballot_w64(bad) && (threadIdx.x & 63) == 0) atomicOr(&st->flags, (unsigned)PB_BAD_COLPTR);
}

// One workgroup per tile of T stored entries, straight from the caller's colptr / rowval (any index width / base):
//   * the tile's column range by two binary searches in colptr, its slice of colptr staged in LDS;
//   * every entry finds its column by a binary search in that slice, down-converts its row, takes its column's colour;
//   * row / colour extent of the coloured entries by wave reductions -> the tile descriptor;
//   * CODES: the 16-bit entry codes.
// Reproduces try_window_plan's build_windows for tiles whose rows form ONE window; other tiles raise PB_NEED_SORT (the
// host builder clusters / sorts them).  Grid-wide statistics are computed by the host from the descriptors.
constexpr int kPbMaxCols = 4096;     // columns (+1) of one tile staged in LDS; tiles spanning more (empty columns): host builder
// BAND: the entries are the slots of a band's column-major storage (fd_plan_create_banded): slot k of column j <-> row
// j - bu + k, bw slots per column, no index arrays; slots outside the matrix and columns without colour are "uncoloured"
// entries (written as 0), exactly as the host builder lists them.  tstride > 1: a sample of the tiles (descriptors only).
template <bool CODES, int T, bool BAND>
__global__ void __launch_bounds__(kBlock) k_pb_tiles(const void *__restrict__ colptr, const void *__restrict__ rowval, int ib, int base,
                                                     int64_t col0, int64_t col1, int64_t e0, int64_t nloc, int64_t M,
                                                     const uint8_t *__restrict__ color8, int64_t ntiles, int4 *__restrict__ wt,
                                                     uint16_t *__restrict__ code, PbStats *st, int64_t bw, int64_t bu, int64_t tstride)
{
    constexpr int E = T / kBlock;                     // entries per thread
    __shared__ int s_cp[kPbMaxCols + 1];              // colptr of the tile's columns, relative to the tile's first entry
    __shared__ int64_t s_j[2];
    __shared__ int s_red[kBlock / 64][7];
    __shared__ int s_tile[2];
    const int64_t t = (int64_t)blockIdx.x * tstride;
    if (t >= ntiles) return;
    const int64_t b0 = t * (int64_t)T;                                    // first local entry of the tile
    const int64_t last = (b0 + T < nloc ? b0 + T : nloc) - 1;            // last REAL local entry (b0 <= last: tiles cover [0, padded))
    if (!BAND && threadIdx.x < 2) {
        // column of local entry q: the largest j with colptr[j] - base - e0 <= q
        const int64_t q = threadIdx.x == 0 ? b0 : last;
        int64_t lo = col0, hi = col1;                                     // invariant: cp(lo) <= q < cp(hi)  (cp(col1) = nloc > q)
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (pb_load(colptr, ib, mid) - base - e0 <= q) lo = mid; else hi = mid;
        }
        s_j[threadIdx.x] = lo;
    }
    __syncthreads();
    const int64_t jlo = BAND ? 0 : s_j[0], jhi = BAND ? 0 : (b0 <= last ? s_j[1] : s_j[0]);
    const int ncols = (int)(jhi - jlo + 1);
    const bool fits = BAND || jhi - jlo + 1 <= kPbMaxCols;
    if (!BAND && fits)
        for (int k = threadIdx.x; k <= ncols; k += kBlock) {
            int64_t v = pb_load(colptr, ib, jlo + k) - base - e0 - b0;
            v = v < -1 ? -1 : (v > T ? T : v);
            s_cp[k] = (int)v;
        }
    __syncthreads();
    int row[E], col[E];                               // col: 0..253 colour, 0xFF none, 0xFE padding
    int rmin = 0x7fffffff, rmax = -1, cmin = 0x7fffffff, cmax = -1, cnt = 0, amin = 0x7fffffff, amax = -1;
    bool badrow = false;
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const int q = u * kBlock + threadIdx.x;
        row[u] = 0; col[u] = 0xFE;
        if (BAND) {
            if (b0 + q < nloc) {
                const int64_t jj = (b0 + q) / bw, k = (b0 + q) - jj * bw, j = col0 + jj, r = j - bu + k;
                const int cb = color8[j];
                const bool in = r >= 0 && r < M && cb != 0xFF;
                row[u] = in ? (int)r : 0;
                col[u] = in ? cb : 0xFF;
                if (in) {
                    rmin = row[u] < rmin ? row[u] : rmin; rmax = row[u] > rmax ? row[u] : rmax;
                    cmin = cb < cmin ? cb : cmin; cmax = cb > cmax ? cb : cmax;
                    ++cnt;
                }
            }
        } else if (fits && b0 + q < nloc) {
            int lo = 0, hi = ncols;                   // s_cp[lo] <= q < s_cp[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_cp[mid] <= q) lo = mid; else hi = mid;
            }
            const int64_t r = pb_load(rowval, ib, e0 + b0 + q) - base;
            if (r < 0 || r >= M) { badrow = true; continue; }
            row[u] = (int)r;
            col[u] = color8[jlo + lo];
            amin = row[u] < amin ? row[u] : amin; amax = row[u] > amax ? row[u] : amax;
            if (col[u] < 0xFE) {
                rmin = row[u] < rmin ? row[u] : rmin; rmax = row[u] > rmax ? row[u] : rmax;
                cmin = col[u] < cmin ? col[u] : cmin; cmax = col[u] > cmax ? col[u] : cmax;
                ++cnt;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int a = __shfl_down(rmin, off, 64), b = __shfl_down(rmax, off, 64), c = __shfl_down(cmin, off, 64),
                  d = __shfl_down(cmax, off, 64), e = __shfl_down(cnt, off, 64), f = __shfl_down(amin, off, 64),
                  g = __shfl_down(amax, off, 64);
        rmin = a < rmin ? a : rmin; rmax = b > rmax ? b : rmax;
        cmin = c < cmin ? c : cmin; cmax = d > cmax ? d : cmax; cnt += e;
        amin = f < amin ? f : amin; amax = g > amax ? g : amax;
    }
    if ((threadIdx.x & 63) == 0) {
        int *r = s_red[threadIdx.x >> 6];
        r[0] = rmin; r[1] = rmax; r[2] = cmin; r[3] = cmax; r[4] = cnt; r[5] = amin; r[6] = amax;
    }
    const bool anybad = __builtin_amdgcn_ballot_w64(badrow) != 0;
    if (anybad && (threadIdx.x & 63) == 0) atomicOr(&st->flags, (unsigned)PB_BAD_ROW);
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            rmin = s_red[w][0] < rmin ? s_red[w][0] : rmin; rmax = s_red[w][1] > rmax ? s_red[w][1] : rmax;
            cmin = s_red[w][2] < cmin ? s_red[w][2] : cmin; cmax = s_red[w][3] > cmax ? s_red[w][3] : cmax;
            cnt += s_red[w][4];
            amin = s_red[w][5] < amin ? s_red[w][5] : amin; amax = s_red[w][6] > amax ? s_red[w][6] : amax;
        }
        int wr0 = 0, pairs = 0, nwin = 0, ncol = 0;
        unsigned flags = fits ? 0u : (unsigned)PB_NEED_SORT;
        if (rmax >= 0) {
            ncol = cmax - cmin + 1;
            if (ncol > kWinMaxCol) flags |= PB_TOO_MANY_COL;
            if (rmax - rmin < 2 * kWinGap || (double)(rmax - rmin + 2) * ncol <= 1.25 * (double)cnt) {
                wr0 = rmin & ~1;
                pairs = (rmax - wr0) / 2 + 1;
                nwin = 1;
            } else {
                flags |= PB_NEED_SORT;
            }
            if (2 * pairs > 2048) flags |= PB_TOO_WIDE;
        }
        wt[3 * t] = int4{rmax >= 0 ? cmin : 0, ncol, pairs, nwin};
        wt[3 * t + 1] = int4{wr0, pairs, 0, pairs};
        wt[3 * t + 2] = int4{0, pairs, 0, pairs};
        s_tile[0] = wr0; s_tile[1] = rmax >= 0 ? cmin : 0;
        if (flags) atomicOr(&st->flags, flags);
        if (amax >= 0) {                              // (plain reads first: almost every tile skips the atomics)
            if (amin < st->row_min) atomicMin(&st->row_min, amin);
            if (amax > st->row_max) atomicMax(&st->row_max, amax);
        }
    }
    if (!CODES) return;
    __syncthreads();
    const int wr0 = s_tile[0], c0 = s_tile[1];
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const int q = u * kBlock + threadIdx.x;
        uint16_t cd;
        if (col[u] == 0xFE) cd = 0x8000;
        else if (col[u] == 0xFF) cd = 0x4000;
        else cd = (uint16_t)((row[u] - wr0) | ((col[u] - c0) << 11));
        code[b0 + q] = cd;
    }
}

// regular[t] = 1 if every code of tile t is a coloured entry and code[q + P] - code[q] == S throughout
__global__ void __launch_bounds__(kBlock) k_pb_periodic(const uint16_t *__restrict__ code, int T, int64_t ntiles, int P, int S,
                                                        uint8_t *__restrict__ regular, PbStats *st)
{
    const int64_t t = blockIdx.x;
    if (t >= ntiles) return;
    const uint16_t *c = code + t * T;
    bool ok = true;
    for (int q = threadIdx.x; q < T; q += kBlock) {
        const int v = c[q];
        ok = ok && v < 0x4000 && (q + P >= T || (int)c[q + P] - v == S);
    }
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    if (__builtin_amdgcn_ballot_w64(!ok) && (threadIdx.x & 63) == 0) atomicOr(&s_bad, 1);
    __syncthreads();
    if (threadIdx.x == 0) regular[t] = s_bad ? 0 : 1;      // (counted by the host: no same-address atomics)
}

__global__ void __launch_bounds__(kBlock) k_pb_set_regular(int4 *__restrict__ wt, const uint8_t *__restrict__ regular, int64_t ntiles)
{
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t < ntiles && regular[t]) wt[3 * t].w |= 0x100;
}

// ---------------------------------------------------------------------------------------------------------------------
// 2-D (strided) tiles on the device: what try_window2d_plan (fdjac_api.hip) does with host loops.
// ---------------------------------------------------------------------------------------------------------------------

// rows / per-entry colours of the sampled tiles (kSortTile entries each) for the gather-coherence estimate
__global__ void __launch_bounds__(kBlock) k_pb_sample_expand(const void *__restrict__ colptr, const void *__restrict__ rowval, int ib, int base,
                                                             int64_t col0, int64_t col1, int64_t e0, int64_t nloc,
                                                             const uint8_t *__restrict__ color8, int64_t step,
                                                             int32_t *__restrict__ rows_out, int32_t *__restrict__ nzc_out)
{
    const int64_t t = (int64_t)blockIdx.x * step;
    for (int k = threadIdx.x; k < kSortTile; k += kBlock) {
        const int64_t q = t * kSortTile + k;
        int32_t r = 0, c = -2;
        if (q < nloc) {
            int64_t lo = col0, hi = col1;                 // cp(lo) <= q < cp(hi)
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if (pb_load(colptr, ib, mid) - base - e0 <= q) lo = mid; else hi = mid;
            }
            r = (int32_t)(pb_load(rowval, ib, e0 + q) - base);
            const int cc = color8[lo];
            c = cc == 0xFF ? -1 : cc;
        }
        rows_out[(size_t)blockIdx.x * kSortTile + k] = r;
        nzc_out[(size_t)blockIdx.x * kSortTile + k] = c;
    }
}

// gather_coherence (fdjac_api.hip) on the device: one workgroup per sampled tile of kSortTile entries.  out[2*t] = sum over
// the tile's 32 wave-level gathers in STORAGE order of the distinct 128-B lines touched, out[2*t+1] = the same in
// (colour, row, position)-sorted order (bitonic sort of the tile in LDS).  Integer counts: the host's averages follow exactly.
__device__ __forceinline__ int pb_distinct64(long long key)
{
    // number of distinct keys among the 64 lanes: a lane counts if no lower lane holds its key
    const int lane = threadIdx.x & 63;
    bool dup = false;
    for (int i = 0; i < 63; ++i) {
        const long long ki = __shfl(key, i, 64);
        dup = dup || (i < lane && ki == key);
    }
    return __popcll(__builtin_amdgcn_ballot_w64(!dup));
}
__global__ void __launch_bounds__(kBlock) k_pb_coherence(const int32_t *__restrict__ rows, const int32_t *__restrict__ nzc, int *__restrict__ out)
{
    __shared__ int s_row[kSortTile], s_col[kSortTile];
    __shared__ unsigned long long s_key[kSortTile];
    __shared__ unsigned short s_idx[kSortTile];
    __shared__ int s_cnt[2];
    const size_t b0 = (size_t)blockIdx.x * kSortTile;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    for (int k = threadIdx.x; k < kSortTile; k += kBlock) {
        const int r = rows[b0 + k], c = nzc[b0 + k];
        s_row[k] = r; s_col[k] = c;
        const unsigned long long cc = c >= 0 ? (unsigned long long)c : (c == -1 ? 0xFFFFFFFEull : 0xFFFFFFFFull);
        s_key[k] = (cc << 32) | (unsigned long long)(unsigned)r;
        s_idx[k] = (unsigned short)k;
    }
    __syncthreads();
    auto line_key = [](int c, int r) { return (long long)(((unsigned long long)(long long)c << 40) | (unsigned long long)(long long)(r >> 4)); };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int direct = 0;
    for (int ii = wave; ii < 2 * (kSortTile / 128); ii += kBlock / 64) {
        const int e = (ii >> 1) * 128 + 2 * lane + (ii & 1);
        direct += pb_distinct64(line_key(s_col[e], s_row[e]));
    }
    // bitonic sort by (key, position)
    for (int k = 2; k <= kSortTile; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < kSortTile; i += kBlock) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long ka = s_key[i], kb = s_key[p];
                    const unsigned short ia = s_idx[i], ib2 = s_idx[p];
                    const bool a_gt_b = ka > kb || (ka == kb && ia > ib2);
                    const bool up = (i & k) == 0;
                    if (a_gt_b == up) { s_key[i] = kb; s_key[p] = ka; s_idx[i] = ib2; s_idx[p] = ia; }
                }
            }
            __syncthreads();
        }
    int sorted = 0;
    for (int ii = wave; ii < 2 * (kSortTile / 128); ii += kBlock / 64) {
        const int e = s_idx[(ii >> 1) * 128 + 64 * (ii & 1) + lane];
        sorted += pb_distinct64(line_key(s_col[e], s_row[e]));
    }
    if (lane == 0) { atomicAdd(&s_cnt[0], direct); atomicAdd(&s_cnt[1], sorted); }
    __syncthreads();
    if (threadIdx.x < 2) out[2 * blockIdx.x + threadIdx.x] = s_cnt[threadIdx.x];
}

// sampled columns for the stride: out[33 * i] = number of entries of column col0 + i * step (or -1: more than 32),
// out[33 * i + 1 ...] = row - column of each
__global__ void __launch_bounds__(kBlock) k_pb2_sample_cols(const void *__restrict__ colptr, const void *__restrict__ rowval, int ib, int base,
                                                            int64_t col0, int64_t ncols, int64_t step, int nsamp, int *__restrict__ out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nsamp) return;
    const int64_t j = col0 + (int64_t)i * step;
    const int64_t a = pb_load(colptr, ib, j) - base, b = pb_load(colptr, ib, j + 1) - base;
    int *o = out + 33 * (size_t)i;
    if (b - a > 32) { o[0] = -1; return; }
    o[0] = (int)(b - a);
    for (int64_t e = a; e < b; ++e) o[1 + (e - a)] = (int)(pb_load(rowval, ib, e) - base - j);
}

struct Pb2Stats {
    unsigned long long bad;        // entries off the stride
    unsigned long long elems;      // sum over tiles of 2 * pairs * ncol
    int ecmax, halo;               // entries per column, reach of an entry around the diagonal / the stride
    int row_min, row_max;
    int max_slots, max_ncol;
    unsigned int flags;            // PB_* (PB_BAD_ROW) | PB2_*
    unsigned int pad;
};
enum { PB2_FAIL = 1u << 16 };      // a tile the 2-D kernel cannot describe (too many colours / windows / rows): host builder

// every column: entries per column, off-stride entries, halo, row range, row validation
__global__ void __launch_bounds__(kBlock) k_pb2_check(const void *__restrict__ colptr, const void *__restrict__ rowval, int ib, int base,
                                                      int64_t col0, int64_t col1, int64_t M, int64_t s, Pb2Stats *st)
{
    int ecmax = 0, halo = 0, rmin = 0x7fffffff, rmax = -1;
    unsigned long long bad = 0;
    bool badrow = false;
    for (int64_t j = col0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < col1; j += (int64_t)gridDim.x * kBlock) {
        const int64_t a = pb_load(colptr, ib, j) - base, b = pb_load(colptr, ib, j + 1) - base;
        const int cnt = (int)(b - a);
        ecmax = cnt > ecmax ? cnt : ecmax;
        for (int64_t e = a; e < b; ++e) {
            const int64_t r = pb_load(rowval, ib, e) - base;
            if (r < 0 || r >= M) { badrow = true; continue; }
            rmin = (int)r < rmin ? (int)r : rmin; rmax = (int)r > rmax ? (int)r : rmax;
            int64_t o = r - j;
            if (o < 0) o = -o;
            if (o <= 8) { halo = (int)o > halo ? (int)o : halo; continue; }
            if (o < s - 4 || o > s + 4) ++bad;
            else { const int h = (int)(o > s ? o - s : s - o); halo = h > halo ? h : halo; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int a = __shfl_down(ecmax, off, 64), b = __shfl_down(halo, off, 64), c = __shfl_down(rmin, off, 64), d = __shfl_down(rmax, off, 64);
        const unsigned long long e = __shfl_down(bad, off, 64);
        ecmax = a > ecmax ? a : ecmax; halo = b > halo ? b : halo; rmin = c < rmin ? c : rmin; rmax = d > rmax ? d : rmax; bad += e;
    }
    const bool anybad = __builtin_amdgcn_ballot_w64(badrow) != 0;
    if ((threadIdx.x & 63) == 0) {
        if (ecmax > st->ecmax) atomicMax(&st->ecmax, ecmax);
        if (halo > st->halo) atomicMax(&st->halo, halo);
        if (rmax >= 0 && rmin < st->row_min) atomicMin(&st->row_min, rmin);
        if (rmax > st->row_max) atomicMax(&st->row_max, rmax);
        if (bad) atomicAdd(&st->bad, bad);
        if (anybad) atomicOr(&st->flags, (unsigned)PB_BAD_ROW);
    }
}

// the column runs of tile (G, I): run r = columns [c0, c1) = local entries [a, b)
struct Pb2Runs {
    int n;
    int64_t c0[kW2MaxRun], a[kW2MaxRun], b[kW2MaxRun];
    int ncols[kW2MaxRun];
};
__device__ __forceinline__ void pb2_runs(Pb2Runs &rn, const void *colptr, int ib, int base, int64_t col0, int64_t col1, int64_t e0,
                                         int64_t s, int L, int R, int64_t g_lo, int64_t g_hi, int64_t G, int64_t I)
{
    rn.n = 0;
    for (int q = 0; q < R; ++q) {
        const int64_t g = g_lo + G * R + q;
        if (g > g_hi) break;
        int64_t c0 = g * s + I * L, c1 = g * s + ((I + 1) * L < s ? (I + 1) * L : s);
        c0 = c0 > col0 ? c0 : col0;
        c1 = c1 < col1 ? c1 : col1;
        if (c1 <= c0) continue;
        const int64_t a = pb_load(colptr, ib, c0) - base - e0, b = pb_load(colptr, ib, c1) - base - e0;
        if (b <= a) continue;
        rn.c0[rn.n] = c0; rn.a[rn.n] = a; rn.b[rn.n] = b; rn.ncols[rn.n] = (int)(c1 - c0);
        ++rn.n;
    }
}

// pass 1: runs and (even-padded) entries per tile -> the host's prefix sums give descriptor slots and code offsets
__global__ void __launch_bounds__(kBlock) k_pb2_count(const void *__restrict__ colptr, int ib, int base, int64_t col0, int64_t col1, int64_t e0,
                                                      int64_t s, int L, int R, int64_t g_lo, int64_t g_hi, int64_t nG, int64_t nI,
                                                      int2 *__restrict__ cnt)
{
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= nG * nI) return;
    Pb2Runs rn;
    pb2_runs(rn, colptr, ib, base, col0, col1, e0, s, L, R, g_lo, g_hi, t / nI, t % nI);
    int nent = 0, raw = 0;
    for (int r = 0; r < rn.n; ++r) {
        const int len = (int)(rn.b[r] - rn.a[r]);
        raw += len;
        nent += len + (len & 1);
    }
    cnt[t] = int2{rn.n ? nent : -1, raw};
}

// pass 2: one workgroup per (non-empty) tile: row windows from a bitmap of the coloured entries' rows (a new window after
// a gap of more than 16 rows, as the host builder's sort does), descriptor, 16-bit entry codes
constexpr int kPb2MaxEnt = 2048 + 2 * kW2MaxRun;
constexpr int kPb2BitWords = 8192;                 // rows spanned by one tile <= 32 * this
__global__ void __launch_bounds__(kBlock) k_pb2_tiles(const void *__restrict__ colptr, const void *__restrict__ rowval, int ib, int base,
                                                      int64_t col0, int64_t col1, int64_t e0, const uint8_t *__restrict__ color8,
                                                      int64_t s, int L, int R, int64_t g_lo, int64_t g_hi, int64_t nG, int64_t nI,
                                                      const int *__restrict__ slot_of, const int64_t *__restrict__ code0_of,
                                                      int *__restrict__ desc, uint16_t *__restrict__ code, Pb2Stats *st)
{
    __shared__ unsigned s_bits[kPb2BitWords];
    __shared__ int s_cp[kW2MaxRun][kPbMaxCols / 32 + 1];     // colptr slices of the runs (relative to the run's first entry): L <= 128 columns
    __shared__ int s_red[kBlock / 64][4];
    __shared__ int s_ext[4];                                  // rmin, rmax, cmin, cmax of the coloured entries
    __shared__ int s_nse[2];                                  // number of window starts / ends found
    __shared__ int s_starts[kW2MaxWin + 1], s_ends[kW2MaxWin + 1];
    __shared__ int s_wr[kW2MaxWin], s_wn[kW2MaxWin], s_cum[kW2MaxWin];
    __shared__ long long s_ra[kW2MaxRun], s_rc0[kW2MaxRun];  // first local entry / first column of every run
    __shared__ int s_rlen[kW2MaxRun], s_rnc[kW2MaxRun], s_roff[kW2MaxRun + 1], s_nrun;   // entries, columns, padded offset in the code list
    const int64_t t = blockIdx.x;
    const int slot = slot_of[t];
    if (slot < 0) return;
    if (threadIdx.x == 0) {
        Pb2Runs rn;
        pb2_runs(rn, colptr, ib, base, col0, col1, e0, s, L, R, g_lo, g_hi, t / nI, t % nI);
        int acc = 0;
        for (int r = 0; r < rn.n; ++r) {
            const int len = (int)(rn.b[r] - rn.a[r]);
            s_ra[r] = rn.a[r]; s_rc0[r] = rn.c0[r]; s_rlen[r] = len; s_rnc[r] = rn.ncols[r]; s_roff[r] = acc;
            acc += len + (len & 1);
        }
        s_roff[rn.n] = acc;
        s_nrun = rn.n;
        s_nse[0] = 0; s_nse[1] = 0;
    }
    __syncthreads();
    const int nrun = s_nrun, nent = s_roff[nrun];
    for (int r = 0; r < nrun; ++r)
        for (int k = threadIdx.x; k <= s_rnc[r]; k += kBlock) s_cp[r][k] = (int)(pb_load(colptr, ib, s_rc0[r] + k) - base - e0 - s_ra[r]);
    __syncthreads();
    // ---- the tile's entries, once, into registers: every load of the tile is in flight together (position pos of the code
    //      list -> run, entry of the run -> column by binary search in the run's colptr slice -> row, colour byte)
    constexpr int E = (kPb2MaxEnt + kBlock - 1) / kBlock;
    int e_row[E], e_cb[E];                               // e_cb: 0..253 colour, 0xFF none, 0x100 not an entry (pad slot / past the end)
#pragma unroll
    for (int uu = 0; uu < E; ++uu) {
        const int pos = uu * kBlock + threadIdx.x;
        e_row[uu] = 0; e_cb[uu] = 0x100;
        if (pos < nent && nent <= kPb2MaxEnt) {
            int r = 0;
            while (r + 1 < nrun && s_roff[r + 1] <= pos) ++r;
            const int q = pos - s_roff[r];
            if (q < s_rlen[r]) {
                int lo = 0, hi = s_rnc[r];                // s_cp[lo] <= q < s_cp[hi]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_cp[r][mid] <= q) lo = mid; else hi = mid;
                }
                e_row[uu] = (int)(pb_load(rowval, ib, e0 + s_ra[r] + q) - base);
                e_cb[uu] = (int)color8[s_rc0[r] + lo];
            }
        }
    }
    // ---- extent of the coloured entries
    int rmin = 0x7fffffff, rmax = -1, cmin = 0x7fffffff, cmax = -1;
    bool fail = false;
    auto for_entries = [&](auto fn) {   // fn(run (unused), position in the tile's code list, row, colour byte)
#pragma unroll
        for (int uu = 0; uu < E; ++uu)
            if (e_cb[uu] != 0x100) fn(0, uu * kBlock + (int)threadIdx.x, e_row[uu], e_cb[uu]);
    };
    for_entries([&](int, int, int row, int cb) {
        if (cb == 0xFF) return;
        rmin = row < rmin ? row : rmin; rmax = row > rmax ? row : rmax;
        cmin = cb < cmin ? cb : cmin; cmax = cb > cmax ? cb : cmax;
    });
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int a = __shfl_down(rmin, off, 64), b = __shfl_down(rmax, off, 64), c = __shfl_down(cmin, off, 64), d = __shfl_down(cmax, off, 64);
        rmin = a < rmin ? a : rmin; rmax = b > rmax ? b : rmax; cmin = c < cmin ? c : cmin; cmax = d > cmax ? d : cmax;
    }
    if ((threadIdx.x & 63) == 0) { int *r = s_red[threadIdx.x >> 6]; r[0] = rmin; r[1] = rmax; r[2] = cmin; r[3] = cmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            rmin = s_red[w][0] < rmin ? s_red[w][0] : rmin; rmax = s_red[w][1] > rmax ? s_red[w][1] : rmax;
            cmin = s_red[w][2] < cmin ? s_red[w][2] : cmin; cmax = s_red[w][3] > cmax ? s_red[w][3] : cmax;
        }
        s_ext[0] = rmin; s_ext[1] = rmax; s_ext[2] = cmin; s_ext[3] = cmax;
    }
    __syncthreads();
    rmin = s_ext[0]; rmax = s_ext[1]; cmin = s_ext[2]; cmax = s_ext[3];
    const bool any = rmax >= 0;
    int nwin = 0, pairs = 0;
    const int ncol = any ? cmax - cmin + 1 : 0;
    if (any) {
        const int base_row = rmin & ~31;
        const int nwords = (rmax - base_row) / 32 + 1;
        if (ncol > kWinMaxCol || nwords > kPb2BitWords) fail = true;
        if (!fail) {
            for (int w = threadIdx.x; w < nwords; w += kBlock) s_bits[w] = 0;
            __syncthreads();
            for_entries([&](int, int, int row, int cb) {
                if (cb != 0xFF) atomicOr(&s_bits[(row - base_row) >> 5], 1u << ((row - base_row) & 31));
            });
            __syncthreads();
            // window starts: a set bit with no set bit in the 16 rows before it; ends: none in the 16 rows after it
            for (int w = threadIdx.x; w < nwords; w += kBlock) {
                const unsigned cur = s_bits[w];
                if (!cur) continue;
                const unsigned prev = w > 0 ? s_bits[w - 1] : 0u, next = w + 1 < nwords ? s_bits[w + 1] : 0u;
                unsigned long long x = ((unsigned long long)cur << 16) | (unsigned long long)(prev >> 16);      // bit i+16 = row bit i
                unsigned long long m = x; m |= m << 1; m |= m << 2; m |= m << 4; m |= m << 8;                    // shifts 0..15
                unsigned starts = (unsigned)(((x & ~(m << 1)) >> 16) & 0xFFFFFFFFull);
                unsigned long long y = (unsigned long long)cur | ((unsigned long long)(next & 0xFFFFu) << 32);   // bit i = row bit i
                unsigned long long n = y; n |= n >> 1; n |= n >> 2; n |= n >> 4; n |= n >> 8;
                unsigned ends = (unsigned)((y & ~(n >> 1)) & 0xFFFFFFFFull);
                while (starts) {
                    const int b = __ffs(starts) - 1; starts &= starts - 1;
                    const int k = atomicAdd(&s_nse[0], 1);
                    if (k < kW2MaxWin + 1) s_starts[k] = base_row + 32 * w + b;
                }
                while (ends) {
                    const int b = __ffs(ends) - 1; ends &= ends - 1;
                    const int k = atomicAdd(&s_nse[1], 1);
                    if (k < kW2MaxWin + 1) s_ends[k] = base_row + 32 * w + b;
                }
            }
            __syncthreads();
            nwin = s_nse[0];
            if (nwin > kW2MaxWin || s_nse[1] != nwin) fail = true;
        }
        if (!fail && threadIdx.x == 0) {
            for (int i = 1; i < nwin; ++i) {              // (insertion sorts of at most 12 values)
                int v = s_starts[i], k = i;
                while (k > 0 && s_starts[k - 1] > v) { s_starts[k] = s_starts[k - 1]; --k; }
                s_starts[k] = v;
                v = s_ends[i]; k = i;
                while (k > 0 && s_ends[k - 1] > v) { s_ends[k] = s_ends[k - 1]; --k; }
                s_ends[k] = v;
            }
            int cum = 0;
            for (int k = 0; k < nwin; ++k) {
                s_wr[k] = s_starts[k] & ~1;
                s_wn[k] = (s_ends[k] - s_wr[k]) / 2 + 1;
                cum += s_wn[k];
                s_cum[k] = cum;
            }
        }
        __syncthreads();
        if (!fail) pairs = s_cum[nwin - 1];
        if (2 * pairs > 2048) fail = true;
    }
    if (nent > kPb2MaxEnt) fail = true;
    if (fail) {
        if (threadIdx.x == 0) atomicOr(&st->flags, (unsigned)PB2_FAIL);
        return;
    }
    // ---- descriptor
    const int64_t code0 = code0_of[t];
    if (threadIdx.x == 0) {
        int *d = desc + (size_t)kW2Desc * (size_t)slot;
        for (int k = 0; k < kW2Desc; ++k) d[k] = 0;
        d[0] = any ? cmin : 0; d[1] = ncol; d[2] = pairs; d[3] = nwin; d[4] = nrun; d[5] = nent;
        d[6] = (int)(unsigned)(code0 & 0xFFFFFFFFll); d[7] = (int)(code0 >> 32);
        for (int k = 0; k < nwin; ++k) { d[8 + 2 * k] = s_wr[k]; d[9 + 2 * k] = s_cum[k]; }
        for (int r = 0; r < nrun; ++r) {
            d[32 + 3 * r] = (int)(unsigned)(s_ra[r] & 0xFFFFFFFFll); d[33 + 3 * r] = (int)(s_ra[r] >> 32);
            d[34 + 3 * r] = s_roff[r + 1];
        }
        if (2 * pairs > st->max_slots) atomicMax(&st->max_slots, 2 * pairs);
        if (ncol > st->max_ncol) atomicMax(&st->max_ncol, ncol);
        atomicAdd(&st->elems, (unsigned long long)(2 * pairs) * (unsigned long long)ncol);
    }
    // ---- codes
    for_entries([&](int, int pos, int row, int cb) {
        uint16_t c = 0x4000;
        if (cb != 0xFF) {
            int k = 0;
            while (!(row >= s_wr[k] && row < s_wr[k] + 2 * s_wn[k])) ++k;
            const int sl = 2 * (k ? s_cum[k - 1] : 0) + (row - s_wr[k]);
            c = (uint16_t)(sl | ((cb - cmin) << 11));
        }
        code[code0 + pos] = c;
    });
    if ((int)threadIdx.x < nrun && (s_rlen[threadIdx.x] & 1))          // the pad slot of odd runs
        code[code0 + s_roff[threadIdx.x] + s_rlen[threadIdx.x]] = 0x8000;
}

// uniform-band test (try_band_plan_csc on the device): columns that do not hold the middle column's w consecutive rows
// j - u .. or break the affine colptr; the nearest violators on either side of the middle column bound the band
struct PbBandStat { long long lo, hi; int mid_bad; int pad; };
__global__ void __launch_bounds__(kBlock) k_pb_band_check(const void *__restrict__ colptr, const void *__restrict__ rowval, int ib, int base,
                                                          int64_t col0, int64_t col1, int64_t jm, int64_t cpm, int64_t w, int64_t u,
                                                          PbBandStat *st)
{
    long long lo = -1, hi = 0x7fffffffffffffffll;
    bool mid = false;
    for (int64_t j = col0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < col1; j += (int64_t)gridDim.x * kBlock) {
        const int64_t a = pb_load(colptr, ib, j) - base, b = pb_load(colptr, ib, j + 1) - base;
        bool bad = (b - a != w) || (a != cpm + w * (j - jm));
        if (!bad)
            for (int64_t k = 0; k < w; ++k) bad = bad || (pb_load(rowval, ib, a + k) - base != j - u + k);
        if (bad) {
            if (j < jm) lo = j > lo ? j : lo;
            else if (j > jm) hi = j < hi ? j : hi;
            else mid = true;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const long long a = __shfl_down(lo, off, 64), b = __shfl_down(hi, off, 64);
        lo = a > lo ? a : lo; hi = b < hi ? b : hi;
    }
    const bool anymid = __builtin_amdgcn_ballot_w64(mid) != 0;
    if ((threadIdx.x & 63) == 0) {
        if (lo > st->lo) atomicMax(&st->lo, lo);
        if (hi < st->hi) atomicMin(&st->hi, hi);
        if (anymid) st->mid_bad = 1;
    }
}

// experimental store capability: is every local column exactly the band column include/fdjac_device.h describes?
__global__ void __launch_bounds__(kBlock) k_pb_band_exact(const void *__restrict__ colptr, const void *__restrict__ rowval, int ib, int base,
                                                          int64_t col0, int64_t col1, fd_band_store d, int *bad_out)
{
    bool bad = false;
    for (int64_t j = col0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < col1; j += (int64_t)gridDim.x * kBlock) {
        const int64_t a = pb_load(colptr, ib, j) - base, b = pb_load(colptr, ib, j + 1) - base;
        const int64_t first = j - d.u > 0 ? j - d.u : 0, last = d.M - 1 < j + d.l ? d.M - 1 : j + d.l;
        bad = bad || a != fd_band_colptr(&d, j) || b != fd_band_colptr(&d, j + 1) || b - a != last - first + 1 || last < first;
        if (!bad)
            for (int64_t k = 0; k < b - a; ++k) bad = bad || (pb_load(rowval, ib, a + k) - base != first + k);
    }
    if (__builtin_amdgcn_ballot_w64(bad) && (threadIdx.x & 63) == 0) atomicOr(bad_out, 1);
}

// the store capability of the 5-point stencil (fd_stencil5_store): every local column holds exactly the stencil's rows at the
// closed-form position, and the columns of every row differ in colour (try_store_plan_stencil5 on the device)
__global__ void __launch_bounds__(kBlock) k_pb_stencil5_exact(const void *__restrict__ colptr, const void *__restrict__ rowval, int ib, int base,
                                                              int64_t col0, int64_t col1, fd_stencil5_store d, const uint8_t *__restrict__ color8,
                                                              int *bad_out)
{
    bool bad = false;
    const int64_t nx = d.nx, N = d.nx * d.ny;
    for (int64_t k = col0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; k < col1; k += (int64_t)gridDim.x * kBlock) {
        const int64_t j = k / nx, i = k - j * nx;
        const int64_t a = pb_load(colptr, ib, k) - base, b = pb_load(colptr, ib, k + 1) - base;
        int64_t want[5];
        int n = 0;
        if (j > 0) want[n++] = k - nx;
        if (i > 0) want[n++] = k - 1;
        want[n++] = k;
        if (i < nx - 1) want[n++] = k + 1;
        if (j < d.ny - 1) want[n++] = k + nx;
        bad = bad || a != fd_stencil5_colptr(&d, k) || b - a != n;
        if (!bad)
            for (int q = 0; q < n; ++q) bad = bad || (pb_load(rowval, ib, a + q) - base != want[q]);
    }
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < N; r += (int64_t)gridDim.x * kBlock) {
        const int64_t j = r / nx, i = r - j * nx;
        int c[5];
        int n = 0;
        c[n++] = color8[r];
        if (i > 0) c[n++] = color8[r - 1];
        if (i < nx - 1) c[n++] = color8[r + 1];
        if (j > 0) c[n++] = color8[r - nx];
        if (j < d.ny - 1) c[n++] = color8[r + nx];
        for (int a = 0; a < n; ++a) {
            bad = bad || c[a] == 0xFF;
            for (int b = a + 1; b < n; ++b) bad = bad || c[a] == c[b];
        }
    }
    if (__builtin_amdgcn_ballot_w64(bad) && (threadIdx.x & 63) == 0) atomicOr(bad_out, 1);
}

// outcome of the device builder
enum { PBR_DONE = 0, PBR_DECLINED = 1 };


// (this file is included by fdjac_api.hip after window_lds_bytes / alloc_scratch are defined)

struct PbTemps {
    void *ptrs[10];
    int n = 0;
    template <typename T> T *add(T *p) { ptrs[n++] = (void *)p; return p; }
    ~PbTemps() { for (int i = 0; i < n; ++i) if (ptrs[i]) (void)hipFree(ptrs[i]); }
};

// device counterpart of try_store_plan_stencil5 (same decisions: the grid width from the middle column, then the exact test)
static void device_store_stencil5(fd_plan *p, const void *d_colptr, const void *d_rowval, int ib, int base, int64_t e0, int64_t C,
                                  const uint8_t *d_color8, bool has_none)
{
    p->store5_ok = false;
    hipStream_t s = p->ctx->stream;
    if (!p->store_allowed || p->store_ok || p->M != p->N || p->col1 - p->col0 < 16 || C < 5 || C > 254 || has_none) return;
    int64_t jm = (p->col0 + p->col1) / 2;
    char raw[2][32];
    if (jm + 3 >= p->col1) return;
    if (hipMemcpyAsync(raw[0], (const char *)d_colptr + (size_t)ib * (size_t)jm, 4 * (size_t)ib, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) return;
    int t = 0;                                   // (the middle column may be the first or the last of its grid row)
    while (t < 2 && load_idx(raw[0], ib, t + 1) - load_idx(raw[0], ib, t) != 5) ++t;
    jm += t;
    const int64_t a = load_idx(raw[0], ib, t) - base, b = load_idx(raw[0], ib, t + 1) - base;
    if (b - a != 5 || a < e0) return;
    if (hipMemcpyAsync(raw[1], (const char *)d_rowval + (size_t)ib * (size_t)(a + 4), (size_t)ib, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) return;
    const int64_t nx = (load_idx(raw[1], ib, 0) - base) - jm;
    if (nx < 4 || (nx & 1) || p->N % nx != 0 || p->N / nx < 3) return;
    fd_stencil5_store d;
    memset(&d, 0, sizeof d);
    d.nx = nx; d.ny = p->N / nx;
    int *d_bad = nullptr, hbad = 0;
    if (hipMalloc((void **)&d_bad, sizeof(int)) != hipSuccess) return;
    if (hipMemcpyAsync(d_bad, &hbad, sizeof hbad, hipMemcpyHostToDevice, s) == hipSuccess) {
        hipLaunchKernelGGL(k_pb_stencil5_exact, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((p->N + kBlock - 1) / kBlock, (int64_t)p->ctx->num_cus * 16))),
                           dim3(kBlock), 0, s, d_colptr, d_rowval, ib, base, p->col0, p->col1, d, d_color8, d_bad);
        hbad = 1;
        if (hipMemcpyAsync(&hbad, d_bad, sizeof hbad, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess && !hbad) {
            p->store5_ok = true; p->store5_nx = nx; p->store5_ny = d.ny;
        }
    }
    (void)hipFree(d_bad);
}

static bool pb_coherence_sample(fd_plan *p, const void *d_colptr, const void *d_rowval, int ib, int base, int64_t e0, int64_t nloc,
                                const uint8_t *d_color8);       // (fdjac_planbuild_lists.hip)

// device counterpart of try_store_plan_csc for plans without row windows (same decisions: width and offset from the middle column,
// then the exact band, corners included)
static void device_store_band(fd_plan *p, const void *d_colptr, const void *d_rowval, int ib, int base, int64_t e0, int64_t e1, int64_t C,
                              int shift, bool cyclic)
{
    p->store_ok = false;
    hipStream_t s = p->ctx->stream;
    if (!p->store_allowed || !cyclic || p->col1 - p->col0 < 4 || p->nnz_local < 1) return;
    const int64_t jm = (p->col0 + p->col1) / 2;
    char raw[2][16];
    if (hipMemcpyAsync(raw[0], (const char *)d_colptr + (size_t)ib * (size_t)jm, 2 * (size_t)ib, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) return;
    const int64_t cp = load_idx(raw[0], ib, 0) - base, w = (load_idx(raw[0], ib, 1) - base) - cp;
    if (w < 1 || w > 64 || C < w || cp < e0 || cp + w > e1) return;
    if (hipMemcpyAsync(raw[1], (const char *)d_rowval + (size_t)ib * (size_t)cp, (size_t)ib, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) return;
    const int64_t u = jm - (load_idx(raw[1], ib, 0) - base);
    if (u < 0 || w - 1 - u < 0) return;
    fd_band_store d;
    memset(&d, 0, sizeof d);
    d.M = p->M; d.N = p->N; d.l = (int)(w - 1 - u); d.u = (int)u; d.C = (int)C; d.shift = shift;
    int *d_bad = nullptr, hbad = 0;
    if (hipMalloc((void **)&d_bad, sizeof(int)) != hipSuccess) return;
    if (hipMemcpyAsync(d_bad, &hbad, sizeof hbad, hipMemcpyHostToDevice, s) == hipSuccess) {
        hipLaunchKernelGGL(k_pb_band_exact, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((p->col1 - p->col0 + kBlock - 1) / kBlock, (int64_t)p->ctx->num_cus * 16))),
                           dim3(kBlock), 0, s, d_colptr, d_rowval, ib, base, p->col0, p->col1, d, d_bad);
        hbad = 1;
        if (hipMemcpyAsync(&hbad, d_bad, sizeof hbad, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess && !hbad) {
            p->store_ok = true; p->store_l = d.l; p->store_u = d.u; p->store_C = d.C; p->store_shift = shift;
        }
    }
    (void)hipFree(d_bad);
}

// 2-D (strided) tiles built on the device -- the decisions are try_window2d_plan's / finish_list_plan's (same helper
// functions, same thresholds), the O(nnz) work is done by the kernels above.  PBR_DONE: p->d_w2desc / d_wcode and the
// window fields are set (the caller finishes the plan); PBR_DECLINED: the host builder decides.
static int device_build_2d(fd_plan *p, const void *d_colptr, const void *d_rowval, int ib, int base, int64_t e0, int64_t nloc,
                           const uint8_t *d_color8, PbTimer &tm, int *row0_out, int *row1_out, int *rc_out, bool *host_too)
{
    // *host_too: declined for a reason try_window2d_plan shares (the host builder would not build 2-D tiles either) -- as opposed
    // to a limit of this builder alone (memory, tile counts)
    *host_too = false;
    auto no2d = [&]() { *host_too = true; return (int)PBR_DECLINED; };
    hipStream_t s = p->ctx->stream;
    const char *fw = fdjac::test_switch("FDJAC_WINDOW2D");
    if (fw && *fw && atoi(fw) == 0) return no2d();
    const int64_t ncols = p->col1 - p->col0;
    if (ncols < 1024 || nloc < 8192 || nloc < 4 * kSortTile) return no2d();
    PbTemps tmp;
    auto sync_ok = [&]() { return hipStreamSynchronize(s) == hipSuccess; };
    // ---- is the storage order a scattered gather? (finish_list_plan's estimate on the same sample of tiles)
    if (!pb_coherence_sample(p, d_colptr, d_rowval, ib, base, e0, nloc, d_color8)) return PBR_DECLINED;
    if (!(p->lines_direct > 16.0 && p->lines_direct > 1.5 * p->lines_sorted)) return no2d();
    tm.mark("2-D: coherence sample");
    // ---- the stride: most common far offset over a sample of columns
    int64_t st_s = 0;
    {
        const int64_t step = std::max<int64_t>(1, ncols / 4096);
        const int nsamp = (int)((ncols + step - 1) / step);
        int *d_o = nullptr;
        if (hipMalloc((void **)&d_o, sizeof(int) * 33 * (size_t)nsamp) != hipSuccess) return PBR_DECLINED;
        tmp.add(d_o);
        hipLaunchKernelGGL(k_pb2_sample_cols, dim3((unsigned)((nsamp + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d_colptr, d_rowval, ib, base,
                           p->col0, ncols, step, nsamp, d_o);
        std::vector<int> o(33 * (size_t)nsamp);
        if (hipMemcpyAsync(o.data(), d_o, sizeof(int) * o.size(), hipMemcpyDeviceToHost, s) != hipSuccess || !sync_ok()) return PBR_DECLINED;
        std::vector<int64_t> fars;
        for (int i = 0; i < nsamp; ++i) {
            const int cnt = o[33 * (size_t)i];
            if (cnt < 0) return no2d();                              // a column with more than 32 entries
            for (int e = 0; e < cnt; ++e) {
                const int64_t d = o[33 * (size_t)i + 1 + e];
                if (d > 8 || d < -8) fars.push_back(d < 0 ? -d : d);
            }
        }
        if (fars.empty()) return no2d();
        std::sort(fars.begin(), fars.end());
        st_s = fars[fars.size() / 2];
        if (st_s < 64 || ncols < 4 * st_s) return no2d();
    }
    Pb2Stats *d_st = nullptr;
    if (hipMalloc((void **)&d_st, sizeof(Pb2Stats)) != hipSuccess) return PBR_DECLINED;
    tmp.add(d_st);
    Pb2Stats h;
    memset(&h, 0, sizeof h);
    h.row_min = 0x7fffffff; h.row_max = -1;
    if (hipMemcpyAsync(d_st, &h, sizeof h, hipMemcpyHostToDevice, s) != hipSuccess) return PBR_DECLINED;
    hipLaunchKernelGGL(k_pb2_check, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((ncols + kBlock - 1) / kBlock, (int64_t)p->ctx->num_cus * 16))),
                       dim3(kBlock), 0, s, d_colptr, d_rowval, ib, base, p->col0, p->col1, p->M, st_s, d_st);
    if (hipMemcpyAsync(&h, d_st, sizeof h, hipMemcpyDeviceToHost, s) != hipSuccess || !sync_ok()) return PBR_DECLINED;
    if (h.flags & PB_BAD_ROW) {
        set_error("colptr / rowval are inconsistent (an entry outside 1..%lld)", (long long)p->M);
        *rc_out = FD_ERR_SHAPE;
        return PBR_DONE;
    }
    const int ecmax = h.ecmax, halo = h.halo;
    if ((int64_t)h.bad * 1000 > nloc || ecmax < 1 || ecmax > 32) return no2d();
    tm.mark("2-D: stride + check");
    int L, R;
    if (!w2_shape(p, ecmax, halo, &L, &R)) return no2d();
    if (L > 128) return PBR_DECLINED;
    const int64_t g_lo = p->col0 / st_s, g_hi = (p->col1 - 1) / st_s;
    const int64_t nG = (g_hi - g_lo + R) / R, nI = (st_s + L - 1) / L;
    const int64_t ntl = nG * nI;
    if (ntl <= 0 || ntl >= ((int64_t)1 << 30)) return PBR_DECLINED;
    // ---- pass 1: entries per tile; prefix sums on the host (8 B per tile)
    int2 *d_cnt = nullptr;
    if (hipMalloc((void **)&d_cnt, sizeof(int2) * (size_t)ntl) != hipSuccess) return PBR_DECLINED;
    tmp.add(d_cnt);
    hipLaunchKernelGGL(k_pb2_count, dim3((unsigned)((ntl + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d_colptr, ib, base, p->col0, p->col1, e0,
                       st_s, L, R, g_lo, g_hi, nG, nI, d_cnt);
    std::vector<int2> cnt((size_t)ntl);
    if (hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int2) * cnt.size(), hipMemcpyDeviceToHost, s) != hipSuccess || !sync_ok()) return PBR_DECLINED;
    std::vector<int> slot_of((size_t)ntl);
    std::vector<int64_t> code0_of((size_t)ntl);
    int64_t ntiles = 0, ncode = 0, covered = 0;
    for (int64_t t = 0; t < ntl; ++t) {
        code0_of[(size_t)t] = ncode;
        if (cnt[(size_t)t].x < 0) { slot_of[(size_t)t] = -1; continue; }
        if (cnt[(size_t)t].x > kPb2MaxEnt) return PBR_DECLINED;
        slot_of[(size_t)t] = (int)ntiles++;
        ncode += cnt[(size_t)t].x;
        covered += cnt[(size_t)t].y;
    }
    if (covered != nloc || ntiles == 0) return PBR_DECLINED;         // every stored entry must belong to exactly one run
    // ---- pass 2: descriptors and codes
    int *d_slot = nullptr, *d_desc = nullptr;
    int64_t *d_code0 = nullptr;
    uint16_t *d_code = nullptr;
    if (hipMalloc((void **)&d_slot, sizeof(int) * (size_t)ntl) != hipSuccess) return PBR_DECLINED;
    tmp.add(d_slot);
    if (hipMalloc((void **)&d_code0, sizeof(int64_t) * (size_t)ntl) != hipSuccess) return PBR_DECLINED;
    tmp.add(d_code0);
    if (hipMalloc((void **)&d_desc, sizeof(int) * (size_t)kW2Desc * (size_t)ntiles) != hipSuccess) return PBR_DECLINED;
    if (hipMalloc((void **)&d_code, sizeof(uint16_t) * (size_t)(ncode + 2)) != hipSuccess) { (void)hipFree(d_desc); return PBR_DECLINED; }
    static const uint16_t tail[2] = {0x8000, 0x8000};                 // the last pair load may touch one code past the end
    bool ok = hipMemcpyAsync(d_slot, slot_of.data(), sizeof(int) * (size_t)ntl, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(d_code0, code0_of.data(), sizeof(int64_t) * (size_t)ntl, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(d_code + ncode, tail, sizeof tail, hipMemcpyHostToDevice, s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_pb2_tiles, dim3((unsigned)ntl), dim3(kBlock), 0, s, d_colptr, d_rowval, ib, base, p->col0, p->col1, e0, d_color8,
                           st_s, L, R, g_lo, g_hi, nG, nI, d_slot, d_code0, d_desc, d_code, d_st);
        ok = hipMemcpyAsync(&h, d_st, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && sync_ok() && hipGetLastError() == hipSuccess;
    }
    tm.mark("2-D: tiles");
    const double overread = (double)h.elems / (double)std::max<int64_t>(nloc, 1);
    if (!ok || (h.flags & PB2_FAIL) || h.max_slots == 0 || window_lds_bytes(p->fdtype, h.max_slots, h.max_ncol) > (size_t)kWinMaxLds ||
        overread > 2.2) {
        (void)hipFree(d_desc); (void)hipFree(d_code);
        return PBR_DECLINED;
    }
    p->window = true;
    p->window2d = true;
    p->w2_ntiles = ntiles;
    p->w2_codes = ncode + 2;
    p->win_tile = 0;
    p->win_pairs = h.max_slots / 2;
    p->win_ncol = h.max_ncol;
    p->win_overread = overread;
    p->d_w2desc = d_desc;
    p->d_wcode = d_code;
    *row0_out = h.row_max >= 0 ? h.row_min : 0;
    *row1_out = h.row_max >= 0 ? h.row_max + 1 : 0;
    return PBR_DONE;
}

// colptr / rowval / colorvec: DEVICE arrays (raw, caller's index width and base).  On PBR_DONE the plan is complete up
// to alloc_scratch (called here); on PBR_DECLINED nothing was changed that the host builder does not overwrite.
// *rc_out carries an error status (FD_ERR_SHAPE etc.) when the pattern is invalid.
// Only the colours (Tridiagonal J: its three diagonals need no entry list): the caller's colorvec (host array, Int32 /
// Int64, 1-based) -> 0-based bytes on the device, C, the cyclic test.  PBR_DECLINED: the host loops (ingest_colors).
static int device_colors_only(fd_plan *p, const void *colorvec, int color_bytes)
{
    hipStream_t s = p->ctx->stream;
    const int64_t N = p->N;
    if (N < 1 || N >= ((int64_t)1 << 31)) return PBR_DECLINED;
    PbTemps tmp;
    void *d_cv = nullptr;
    PbStats *d_st = nullptr;
    if (hipMalloc(&d_cv, (size_t)color_bytes * (size_t)N) != hipSuccess) return PBR_DECLINED;
    tmp.add(d_cv);
    if (hipMalloc((void **)&d_st, sizeof(PbStats)) != hipSuccess) return PBR_DECLINED;
    tmp.add(d_st);
    PbStats h;
    memset(&h, 0, sizeof h);
    if (hipMemcpyAsync(d_cv, colorvec, (size_t)color_bytes * (size_t)N, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_st, &h, sizeof h, hipMemcpyHostToDevice, s) != hipSuccess) return PBR_DECLINED;
    const int gN = (int)std::min<int64_t>((N + kBlock - 1) / kBlock, (int64_t)p->ctx->num_cus * 16);
    hipLaunchKernelGGL(k_pb_colmax, dim3(gN), dim3(kBlock), 0, s, d_cv, color_bytes, N, d_st);
    if (hipMemcpyAsync(&h, d_st, sizeof h, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return PBR_DECLINED;
    const int64_t C = (int64_t)h.max_color;
    if (C < 1 || C > kRegColors) return PBR_DECLINED;
    const int shift = h.first_color >= 1 ? (int)(h.first_color - 1) : 0;
    uint8_t *d_color8 = nullptr;
    if (hipMalloc((void **)&d_color8, (size_t)N) != hipSuccess) return PBR_DECLINED;
    hipLaunchKernelGGL(k_pb_colors, dim3(gN), dim3(kBlock), 0, s, d_cv, color_bytes, N, (int)C, shift, d_color8, d_st);
    if (hipMemcpyAsync(&h, d_st, sizeof h, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess ||
        (h.flags & PB_COLOR_BIG)) {
        (void)hipFree(d_color8);
        return PBR_DECLINED;
    }
    p->C = C;
    p->color8 = true;
    p->d_color = d_color8;
    const char *fc = fdjac::test_switch("FDJAC_EPS_CYCLIC");
    const bool cyc = !(h.flags & (PB_NOT_CYCLIC | PB_NONE)) && !(fc && *fc && atoi(fc) == 0) && p->fdtype != FD_COMPLEX;
    p->cyc_C = cyc ? (int)C : 0;
    p->cyc_shift = cyc ? shift : 0;
    p->built_on_device = true;
    return PBR_DONE;
}

static void pb_launch_tiles(bool codes, int T, bool band, int64_t grid, int64_t tstride, hipStream_t s, const void *d_colptr, const void *d_rowval,
                            int ib, int base, const fd_plan *p, int64_t e0, int64_t nloc, const uint8_t *d_color8, int64_t ntiles, int4 *d_wt,
                            uint16_t *d_code, PbStats *d_st, int64_t bw, int64_t bu)
{
#define FD_PB_TILES(CC, TT, BB)                                                                                                       \
    hipLaunchKernelGGL((k_pb_tiles<CC, TT, BB>), dim3((unsigned)grid), dim3(kBlock), 0, s, d_colptr, d_rowval, ib, base, p->col0, p->col1, e0, \
                       nloc, p->M, d_color8, ntiles, d_wt, d_code, d_st, bw, bu, tstride)
    if (!codes) { if (T == 2048) FD_PB_TILES(false, 2048, false); else if (T == 1024) FD_PB_TILES(false, 1024, false); else FD_PB_TILES(false, 512, false); }
    else if (band) { if (T == 2048) FD_PB_TILES(true, 2048, true); else if (T == 1024) FD_PB_TILES(true, 1024, true); else FD_PB_TILES(true, 512, true); }
    else { if (T == 2048) FD_PB_TILES(true, 2048, false); else if (T == 1024) FD_PB_TILES(true, 1024, false); else FD_PB_TILES(true, 512, false); }
#undef FD_PB_TILES
}

static int device_build_lists(fd_plan *p, const void *d_colptr, const void *d_rowval, int ib, int base, int64_t e0, int64_t nloc,
                              const uint8_t *d_color8, int64_t C, PbTimer &tm, int *row0_out, int *row1_out, int *rc_out);

// alloc_scratch for a device-built plan: the cyclic test was made by the kernels; only the many-colour reduction needs the colours
static int device_alloc_scratch(fd_plan *p, const uint8_t *d_color8)
{
    std::vector<int32_t> col0;
    if (p->C > kRegColors && p->fdtype != FD_COMPLEX) {
        std::vector<uint8_t> c8((size_t)p->N);
        FD_HIP_CHECK(hipMemcpy(c8.data(), d_color8, (size_t)p->N, hipMemcpyDeviceToHost));
        col0.resize((size_t)p->N);
        for (int64_t j = 0; j < p->N; ++j) col0[(size_t)j] = c8[(size_t)j] == 0xFF ? -1 : (int32_t)c8[(size_t)j];
    }
    return alloc_scratch(p, col0);
}

// band != nullptr: the "pattern" is a band's column-major storage (fd_plan_create_banded; d_colptr / d_rowval unused,
// entries [0, e1) = the slots of the local columns).
struct PbBand { int64_t w, u; };
static int device_build_csc(fd_plan *p, const void *d_colptr, const void *d_rowval, int idx_bytes, int idx_base,
                            const void *d_colorvec, int color_bytes, int64_t e0, int64_t e1, int *rc_out, const PbBand *band = nullptr)
{
    *rc_out = FD_OK;
    hipStream_t s = p->ctx->stream;
    const int64_t N = p->N, nloc = e1 - e0;
    const char *fw = fdjac::test_switch("FDJAC_WINDOW"), *fso = fdjac::test_switch("FDJAC_SORTED");
    if ((fw && *fw && atoi(fw) == 0) || (fso && *fso && atoi(fso) == 1)) return PBR_DECLINED;   // forced gather kernels
    if (nloc <= 0 || N >= ((int64_t)1 << 31)) return PBR_DECLINED;
    PbTemps tmp;
    PbTimer tm(s);
    PbStats *d_st = nullptr;
    if (hipMalloc((void **)&d_st, sizeof(PbStats)) != hipSuccess) return PBR_DECLINED;
    tmp.add(d_st);
    PbStats h;
    memset(&h, 0, sizeof h);
    h.row_min = 0x7fffffff; h.row_max = -1;
    if (hipMemcpyAsync(d_st, &h, sizeof h, hipMemcpyHostToDevice, s) != hipSuccess) return PBR_DECLINED;
    const int gN = (int)std::min<int64_t>((N + kBlock - 1) / kBlock, (int64_t)p->ctx->num_cus * 16);
    hipLaunchKernelGGL(k_pb_colmax, dim3(gN), dim3(kBlock), 0, s, d_colorvec, color_bytes, N, d_st);
    if (hipMemcpyAsync(&h, d_st, sizeof h, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return PBR_DECLINED;
    const int64_t C = (int64_t)h.max_color;
    // (more than kRegColors colours: the step-size reduction walks per-colour column lists -- built by alloc_scratch's counting sort from
    //  the colours copied back, N bytes; a BandedMatrix with that many colours stays with the host builder)
    if (C < 1 || C > 253 || (C > kRegColors && band)) return PBR_DECLINED;
    tm.mark("colour maximum");
    const int shift = h.first_color >= 1 ? (int)(h.first_color - 1) : 0;
    uint8_t *d_color8 = nullptr;
    if (hipMalloc((void **)&d_color8, (size_t)N) != hipSuccess) return PBR_DECLINED;
    hipLaunchKernelGGL(k_pb_colors, dim3(gN), dim3(kBlock), 0, s, d_colorvec, color_bytes, N, (int)C, shift, d_color8, d_st);
    tm.mark("colours (alloc + kernel)");
    if (!band)
    hipLaunchKernelGGL(k_pb_check_colptr, dim3(std::max(1, (int)std::min<int64_t>((p->col1 - p->col0 + kBlock - 1) / kBlock, (int64_t)p->ctx->num_cus * 16))),
                       dim3(kBlock), 0, s, d_colptr, idx_bytes, idx_base, p->col0, p->col1, e0, e1, d_st);
    tm.mark("colptr check");
    const size_t padded = (size_t)((std::max<int64_t>(nloc, 1) + kListPad - 1) / kListPad * kListPad);
    // tile size: the host builder's rule (try_window_plan).  Every candidate is ONE fused pass (expand + extent + codes);
    // the statistics the rule needs are computed on the host from the descriptors (48 B per tile).
    const char *ft = fdjac::test_switch("FDJAC_WIN_TILE");
    const int force_t = (ft && *ft) ? atoi(ft) : 0;
    const int force_w = (fw && *fw) ? atoi(fw) : -1;
    const bool prefer_small = sizeof(real_t) >= 8;   // (the host builder's rule, try_window_plan)
    int4 *d_wt = nullptr;
    uint16_t *d_code = nullptr;
    if (hipMalloc((void **)&d_wt, sizeof(int4) * 3 * (padded / 512)) != hipSuccess) { (void)hipFree(d_color8); return PBR_DECLINED; }
    if (hipMalloc((void **)&d_code, sizeof(uint16_t) * padded) != hipSuccess) { (void)hipFree(d_color8); (void)hipFree(d_wt); return PBR_DECLINED; }
    int bestT = 0;
    struct HostStats { int max_slots = 0, max_ncol = 0; double elems = 0; };
    HostStats best;
    std::vector<int4> wt;
    PbStats fin;
    memset(&fin, 0, sizeof fin);
    bool declined = false, bad = false;
    for (int T : {2048, 1024, 512}) {
        if (force_t && T != force_t) continue;
        if (!force_t && T == 2048 && prefer_small) continue;
        const int64_t ntiles = (int64_t)(padded / (size_t)T);
        {   // clear the per-pass flags, keep colour / validation results
            PbStats cur;
            if (hipMemcpyAsync(&cur, d_st, sizeof cur, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { declined = true; break; }
            if (cur.flags & PB_BAD_COLPTR) { bad = true; break; }
            if (cur.flags & PB_COLOR_BIG) { declined = true; break; }
            cur.flags &= ~(unsigned)(PB_NEED_SORT | PB_TOO_MANY_COL | PB_TOO_WIDE);
            cur.row_min = 0x7fffffff; cur.row_max = -1;
            (void)hipMemcpyAsync(d_st, &cur, sizeof cur, hipMemcpyHostToDevice, s);
        }
        if (!band && !force_t) {
            // a sample of the tiles first (descriptors only): scattered patterns are recognised without a full pass
            const int64_t ts = std::max<int64_t>(1, ntiles / 64);
            pb_launch_tiles(false, T, false, (ntiles + ts - 1) / ts, ts, s, d_colptr, d_rowval, idx_bytes, idx_base, p, e0, nloc, d_color8, ntiles, d_wt, d_code, d_st, 0, 0);
            PbStats smp;
            if (hipMemcpyAsync(&smp, d_st, sizeof smp, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { declined = true; break; }
            if (smp.flags & PB_NEED_SORT) { fin = smp; declined = true; break; }
        }
        pb_launch_tiles(true, T, band != nullptr, ntiles, 1, s, d_colptr, d_rowval, idx_bytes, idx_base, p, e0, nloc, d_color8, ntiles, d_wt, d_code, d_st,
                        band ? band->w : 0, band ? band->u : 0);
        wt.resize((size_t)(3 * ntiles));
        if (hipMemcpyAsync(wt.data(), d_wt, sizeof(int4) * wt.size(), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(&fin, d_st, sizeof fin, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { declined = true; break; }
        if (fin.flags & (PB_BAD_ROW | PB_BAD_COLPTR)) { bad = true; break; }
        if (fin.flags & PB_NEED_SORT) { declined = true; break; }                     // scattered pattern: the host builder's job
        if (fin.flags & (PB_TOO_MANY_COL | PB_TOO_WIDE)) continue;                    // the host builder rejects this T too
        HostStats cur;
        for (int64_t t = 0; t < ntiles; ++t) {
            const int4 th = wt[3 * (size_t)t];
            cur.max_slots = std::max(cur.max_slots, 2 * th.z);
            cur.max_ncol = std::max(cur.max_ncol, th.y);
            cur.elems += 2.0 * th.z * th.y;
        }
        if (cur.max_slots <= 0) continue;
        const size_t lds = window_lds_bytes(p->fdtype, cur.max_slots, cur.max_ncol);
        if (lds > (size_t)kWinMaxLds) continue;
        const double overread = cur.elems / (double)std::max<int64_t>(nloc, 1);
        if (!(overread <= 1.25 || force_w == 1)) continue;
        bestT = T;
        best = cur;
        if (lds <= (size_t)32 * 1024 || T == 1024) break;
        // (a larger tile was acceptable but a smaller one may be better: the loop goes on and, if the smaller one is
        //  rejected, the arrays on the device are those of the LAST pass -- rebuilt below)
    }
    tm.mark("tile pass(es)");
    if (bad) {
        (void)hipFree(d_color8); (void)hipFree(d_wt); (void)hipFree(d_code);
        set_error("colptr / rowval are inconsistent (an entry outside 1..%lld or colptr not monotone)", (long long)p->M);
        *rc_out = FD_ERR_SHAPE;
        return PBR_DONE;
    }
    if (!bad && !band && ((declined && (fin.flags & PB_NEED_SORT)) || (!declined && !bestT))) {
        // not a locally banded pattern (or one whose tiles hold too many colours / rows for a window): 2-D (strided) tiles, if it is a
        // 2-D stencil in natural ordering
        (void)hipFree(d_wt); (void)hipFree(d_code);
        p->C = C;                                      // (w2_shape sizes the LDS tile with the number of colours)
        int r0 = 0, r1 = 0;
        bool host_too = false;
        int res2 = device_build_2d(p, d_colptr, d_rowval, idx_bytes, idx_base, e0, nloc, d_color8, tm, &r0, &r1, rc_out, &host_too);
        // not a 2-D stencil either: index lists (sorted tiles for a scattered storage order), if no window tile size can work
        if (res2 == PBR_DECLINED && host_too && *rc_out == FD_OK)
            res2 = device_build_lists(p, d_colptr, d_rowval, idx_bytes, idx_base, e0, nloc, d_color8, C, tm, &r0, &r1, rc_out);
        if (res2 != PBR_DONE || *rc_out != FD_OK) { (void)hipFree(d_color8); return res2; }
        p->color8 = true;
        p->d_color = d_color8;
        p->has_none = (fin.flags & PB_NONE) != 0;
        p->nnz_local = nloc;
        p->row0 = r0;
        p->row1 = r1;
        const char *fc = fdjac::test_switch("FDJAC_EPS_CYCLIC");
        const bool cyc = !(fin.flags & (PB_NOT_CYCLIC | PB_NONE)) && !(fc && *fc && atoi(fc) == 0) && p->fdtype != FD_COMPLEX &&
                         C <= kRegColors;       // (computed colours are the register reduction's: alloc_scratch)
        p->cyc_C = cyc ? (int)C : 0;
        p->cyc_shift = cyc ? shift : 0;
        // (an exact band whose colours outnumber what a window tile holds ends up here too: its store capability, as in the main path)
        device_store_band(p, d_colptr, d_rowval, idx_bytes, idx_base, e0, e1, C, shift, !(fin.flags & (PB_NOT_CYCLIC | PB_NONE)));
        if (!p->store_ok) device_store_stencil5(p, d_colptr, d_rowval, idx_bytes, idx_base, e0, C, d_color8, p->has_none);
        tm.mark("store tests");
        p->built_on_device = true;
        *rc_out = device_alloc_scratch(p, d_color8);
        tm.mark("scratch allocation");
        return PBR_DONE;
    }
    if (declined || !bestT) { (void)hipFree(d_color8); (void)hipFree(d_wt); (void)hipFree(d_code); return PBR_DECLINED; }
    const int64_t ntiles = (int64_t)(padded / (size_t)bestT);
    if ((int64_t)wt.size() != 3 * ntiles) {   // the accepted tile size is not the one of the last pass: run it again
        pb_launch_tiles(true, bestT, band != nullptr, ntiles, 1, s, d_colptr, d_rowval, idx_bytes, idx_base, p, e0, nloc, d_color8, ntiles, d_wt, d_code, d_st,
                        band ? band->w : 0, band ? band->u : 0);
        wt.resize((size_t)(3 * ntiles));
        if (hipMemcpyAsync(wt.data(), d_wt, sizeof(int4) * wt.size(), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            (void)hipFree(d_color8); (void)hipFree(d_wt); (void)hipFree(d_code);
            return PBR_DECLINED;
        }
    }
    // periodic codes: the period is found on three sample tiles on the host (a few KB), every tile is tested on the device
    int P = 0, S = 0, magic = 0;
    {
        const char *fp = fdjac::test_switch("FDJAC_WIN_PERIODIC");
        if (!(fp && *fp && atoi(fp) == 0) && ntiles >= 3) {
            std::vector<uint16_t> smp((size_t)bestT);
            for (int64_t sample : {ntiles / 2, ntiles / 4, (3 * ntiles) / 4}) {
                if (hipMemcpyAsync(smp.data(), d_code + sample * bestT, sizeof(uint16_t) * (size_t)bestT, hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess) break;
                const uint16_t *c = smp.data();
                for (int cand = 1; cand <= kWinPeriodMax && !P; ++cand) {
                    const int s0 = (int)c[cand] - (int)c[0];
                    bool okp = true;
                    for (int q = 0; q < bestT && okp; ++q)
                        okp = c[q] < 0x4000 && (q + cand >= bestT || (int)c[q + cand] - (int)c[q] == s0);
                    if (okp) { P = cand; S = s0; }
                }
                if (P) break;
            }
        }
        if (P) {
            magic = (int)(((1u << 20) + (unsigned)P - 1) / (unsigned)P);
            for (int q = 0; q < bestT; ++q)
                if ((int)(((int64_t)q * magic) >> 20) != q / P) { P = 0; break; }
        }
        if (P) {
            uint8_t *d_reg = nullptr;
            if (hipMalloc((void **)&d_reg, (size_t)ntiles) == hipSuccess) {
                tmp.add(d_reg);
                hipLaunchKernelGGL(k_pb_periodic, dim3((unsigned)ntiles), dim3(kBlock), 0, s, d_code, bestT, ntiles, P, S, d_reg, d_st);
                std::vector<uint8_t> reg((size_t)ntiles);
                int64_t regular = 0;
                if (hipMemcpyAsync(reg.data(), d_reg, (size_t)ntiles, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess)
                    for (uint8_t v : reg) regular += v;
                if (2 * regular >= ntiles) {
                    hipLaunchKernelGGL(k_pb_set_regular, dim3((unsigned)((ntiles + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d_wt, d_reg, ntiles);
                    for (int64_t t = 0; t < ntiles; ++t) if (reg[(size_t)t]) wt[3 * (size_t)t].w |= 0x100;
                } else {
                    P = 0;
                }
            } else {
                P = 0;
            }
        }
    }
    tm.mark("periodicity");
    p->C = C;
    p->color8 = true;
    p->d_color = d_color8;
    if (!band) p->has_none = (fin.flags & PB_NONE) != 0;        // (the banded kinds write their uncoloured slots through the entry codes)
    p->nnz_local = nloc;
    if (!band) {
        p->row0 = fin.row_max >= 0 ? fin.row_min : 0;
        p->row1 = fin.row_max >= 0 ? (int64_t)fin.row_max + 1 : 0;
    }
    p->window = true;
    p->win_tile = bestT;
    p->win_pairs = best.max_slots / 2;
    p->win_ncol = best.max_ncol;
    p->win_overread = best.elems / (double)std::max<int64_t>(nloc, 1);
    p->win_per_P = P; p->win_per_S = P ? S : 0; p->win_per_magic = P ? magic : 0;
    p->d_wtiles = d_wt;
    p->d_wcode = d_code;
    {
        const char *fc = fdjac::test_switch("FDJAC_EPS_CYCLIC");
        const bool cyc = !(fin.flags & (PB_NOT_CYCLIC | PB_NONE)) && !(fc && *fc && atoi(fc) == 0) &&
                         p->fdtype != FD_COMPLEX && C <= kRegColors;   // (the complex step has no step-size reduction; many colours: its lists)
        p->cyc_C = cyc ? (int)C : 0;
        p->cyc_shift = cyc ? shift : 0;
    }
    // uniform band with cyclic colours: width and offset from the middle column (as try_band_plan_csc / try_store_plan_csc read
    // them on the host), then (a) the tiles whose descriptors the row-window kernel computes (finish_band_plan) and (b) the
    // store capability -- each decided on its own, like the host builder does
    const bool cyclic = !(fin.flags & (PB_NOT_CYCLIC | PB_NONE));
    int64_t mid_w = 0, mid_u = 0, mid_cp = 0;
    const int64_t jm = (p->col0 + p->col1) / 2;
    bool mid_ok = false;
    if (!band && cyclic && p->col1 - p->col0 >= 4 && (p->bd_allowed || p->store_allowed)) {
        char raw[2][16];
        const size_t ib = (size_t)idx_bytes;
        mid_ok = hipMemcpyAsync(raw[0], (const char *)d_colptr + ib * (size_t)jm, 2 * ib, hipMemcpyDeviceToHost, s) == hipSuccess &&
                 hipStreamSynchronize(s) == hipSuccess;
        mid_cp = mid_ok ? load_idx(raw[0], idx_bytes, 0) - idx_base : 0;
        mid_w = mid_ok ? (load_idx(raw[0], idx_bytes, 1) - idx_base) - mid_cp : 0;
        mid_ok = mid_ok && mid_w >= 1 && mid_w <= 64 && mid_cp >= e0 && mid_cp + mid_w <= e1 &&
                 hipMemcpyAsync(raw[1], (const char *)d_rowval + ib * (size_t)mid_cp, ib, hipMemcpyDeviceToHost, s) == hipSuccess &&
                 hipStreamSynchronize(s) == hipSuccess;
        if (mid_ok) mid_u = jm - (load_idx(raw[1], idx_bytes, 0) - idx_base);
    }
    if (p->bd_allowed && cyclic) {
        if (band) {
            finish_band_plan(p, band->w, band->u, 0, p->col0, p->col1, C, shift, wt.data());
        } else if (mid_ok) {
            PbBandStat hs{-1, 0x7fffffffffffffffll, 0, 0}, *d_bs = nullptr;
            bool ok = hipMalloc((void **)&d_bs, sizeof(PbBandStat)) == hipSuccess;
            if (ok) {
                tmp.add(d_bs);
                ok = hipMemcpyAsync(d_bs, &hs, sizeof hs, hipMemcpyHostToDevice, s) == hipSuccess;
                if (ok) {
                    hipLaunchKernelGGL(k_pb_band_check, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((p->col1 - p->col0 + kBlock - 1) / kBlock, (int64_t)p->ctx->num_cus * 16))),
                                       dim3(kBlock), 0, s, d_colptr, d_rowval, idx_bytes, idx_base, p->col0, p->col1, jm, mid_cp, mid_w, mid_u, d_bs);
                    ok = hipMemcpyAsync(&hs, d_bs, sizeof hs, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
                }
                if (ok && !hs.mid_bad) {
                    const int64_t ju0 = std::max<int64_t>(hs.lo + 1, p->col0), ju1 = std::min<int64_t>(hs.hi, p->col1);
                    finish_band_plan(p, mid_w, mid_u, (mid_cp - e0) + mid_w * (ju0 - jm), ju0, ju1, C, shift, wt.data());
                }
            }
        }
    }
    // the store capability (fd_band_store): the exact band, corners included
    p->store_ok = false;
    if (p->store_allowed && !band && mid_ok && C >= mid_w && mid_u >= 0 && mid_w - 1 - mid_u >= 0 && p->nnz_local >= 1) {
        fd_band_store d;
        memset(&d, 0, sizeof d);
        d.M = p->M; d.N = p->N; d.l = (int)(mid_w - 1 - mid_u); d.u = (int)mid_u; d.C = (int)C; d.shift = shift;
        int *d_bad = nullptr, hbad = 0;
        if (hipMalloc((void **)&d_bad, sizeof(int)) == hipSuccess) {
            tmp.add(d_bad);
            if (hipMemcpyAsync(d_bad, &hbad, sizeof hbad, hipMemcpyHostToDevice, s) == hipSuccess) {
                hipLaunchKernelGGL(k_pb_band_exact, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((p->col1 - p->col0 + kBlock - 1) / kBlock, (int64_t)p->ctx->num_cus * 16))),
                                   dim3(kBlock), 0, s, d_colptr, d_rowval, idx_bytes, idx_base, p->col0, p->col1, d, d_bad);
                hbad = 1;
                if (hipMemcpyAsync(&hbad, d_bad, sizeof hbad, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess && !hbad) {
                    p->store_ok = true; p->store_l = d.l; p->store_u = d.u; p->store_C = d.C; p->store_shift = shift;
                }
            }
        }
    }
    if (!band && !p->store_ok) device_store_stencil5(p, d_colptr, d_rowval, idx_bytes, idx_base, e0, C, d_color8, p->has_none);
    tm.mark("band test");
    p->built_on_device = true;
    tm.mark("descriptors to host");
    *rc_out = device_alloc_scratch(p, d_color8);            // (the cyclic test above stands)
    tm.mark("scratch allocation");
    return PBR_DONE;
}

}  // namespace fdjac
