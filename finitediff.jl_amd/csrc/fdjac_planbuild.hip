// Plan construction ON THE DEVICE for the common-pattern CSC path (fd_plan_create_csc / fd_plan_create_csc_device).
//
// What a plan compiles is the per-call pattern work of the reference -- the O(nnz) pattern comparison, the O(N) colour
// scan per colour and the colptr walk of `_colorediteration!` (src/jacobians.jl:524-535,547;
// ext/FiniteDiffSparseArraysExt.jl:38-47,51-52) -- into tile descriptors and 16-bit entry codes.  The host builder
// (fdjac_api.hip: csc_common / try_window_plan) does that with serial loops over nnz (0.3 s at nnz = 3*10^7).  Here the
// same arrays are produced by kernels:
//   k_pb_colmax / k_pb_colors   colorvec (Int32 / Int64, 1-based) -> 0-based uint8 colours, C = maximum(colorvec),
//                               "some column has no colour", "colours are cyclic" (wave ballots)
//   k_pb_tiles                  one workgroup per tile of stored entries, straight from colptr / rowval: the tile's slice
//                               of colptr staged in LDS, every entry finds its column by binary search, down-converts its
//                               row (the index down-conversion SURVEY section 7 asks for) and takes its column's colour;
//                               row / colour extent by wave reductions -> tile descriptor; the 16-bit entry codes
//   k_pb_periodic               which tiles repeat their codes with the plan-wide period (ballot over the tile)
// No intermediate per-entry arrays, no same-address atomics in the hot kernels (grid-wide statistics are computed by the
// host from the 48-byte tile descriptors).
// The pattern may already live on the device (fd_plan_create_csc_device: nothing crosses PCIe) or is uploaded raw.
// Patterns the device builder does not handle (tiles that need several row windows or a sort: scattered stencils; more
// than 8 colours; forced kernel variants) make it step aside -- the host builder then runs as before.  The host builder
// is also the CHECKER: tests build every plan both ways (FDJAC_PLAN_DEVICE=0/1) and compare the plan arrays bit for bit
// (fd_plan_checksum).
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
template <bool CODES, int T>
__global__ void __launch_bounds__(kBlock) k_pb_tiles(const void *__restrict__ colptr, const void *__restrict__ rowval, int ib, int base,
                                                     int64_t col0, int64_t col1, int64_t e0, int64_t nloc, int64_t M,
                                                     const uint8_t *__restrict__ color8, int64_t ntiles, int4 *__restrict__ wt,
                                                     uint16_t *__restrict__ code, PbStats *st)
{
    constexpr int E = T / kBlock;                     // entries per thread
    __shared__ int s_cp[kPbMaxCols + 1];              // colptr of the tile's columns, relative to the tile's first entry
    __shared__ int64_t s_j[2];
    __shared__ int s_red[kBlock / 64][7];
    __shared__ int s_tile[2];
    const int64_t t = blockIdx.x;
    if (t >= ntiles) return;
    const int64_t b0 = t * (int64_t)T;                                    // first local entry of the tile
    const int64_t last = (b0 + T < nloc ? b0 + T : nloc) - 1;            // last REAL local entry (b0 <= last: tiles cover [0, padded))
    if (threadIdx.x < 2) {
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
    const int64_t jlo = s_j[0], jhi = b0 <= last ? s_j[1] : s_j[0];
    const int ncols = (int)(jhi - jlo + 1);
    const bool fits = jhi - jlo + 1 <= kPbMaxCols;
    if (fits)
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
        if (fits && b0 + q < nloc) {
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

// outcome of the device builder
enum { PBR_DONE = 0, PBR_DECLINED = 1 };

// FDJAC_PLAN_TIMING=1: wall-clock of the builder's sections on stderr (each mark synchronises the stream)
struct PbTimer {
    bool on;
    hipStream_t s;
    double t0;
    static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    PbTimer(hipStream_t st) : s(st) { const char *v = getenv("FDJAC_PLAN_TIMING"); on = v && *v && atoi(v) != 0; t0 = on ? now() : 0; }
    void mark(const char *what) { if (!on) return; (void)hipStreamSynchronize(s); const double t = now(); fprintf(stderr, "[fdjac plan] %-28s %8.3f ms\n", what, t - t0); t0 = t; }
};

// (this file is included by fdjac_api.hip after window_lds_bytes / plan_row_strips / alloc_scratch are defined)

struct PbTemps {
    void *ptrs[10];
    int n = 0;
    template <typename T> T *add(T *p) { ptrs[n++] = (void *)p; return p; }
    ~PbTemps() { for (int i = 0; i < n; ++i) if (ptrs[i]) (void)hipFree(ptrs[i]); }
};

// colptr / rowval / colorvec: DEVICE arrays (raw, caller's index width and base).  On PBR_DONE the plan is complete up
// to alloc_scratch (called here); on PBR_DECLINED nothing was changed that the host builder does not overwrite.
// *rc_out carries an error status (FD_ERR_SHAPE etc.) when the pattern is invalid.
static int device_build_csc(fd_plan *p, const void *d_colptr, const void *d_rowval, int idx_bytes, int idx_base,
                            const void *d_colorvec, int color_bytes, int64_t e0, int64_t e1, int *rc_out)
{
    *rc_out = FD_OK;
    hipStream_t s = p->ctx->stream;
    const int64_t N = p->N, nloc = e1 - e0;
    const char *fw = getenv("FDJAC_WINDOW"), *fso = getenv("FDJAC_SORTED");
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
    if (C < 1 || C > kRegColors) return PBR_DECLINED;            // many colours: segmented reduction lists are built on the host
    tm.mark("colour maximum");
    const int shift = h.first_color >= 1 ? (int)(h.first_color - 1) : 0;
    uint8_t *d_color8 = nullptr;
    if (hipMalloc((void **)&d_color8, (size_t)N) != hipSuccess) return PBR_DECLINED;
    hipLaunchKernelGGL(k_pb_colors, dim3(gN), dim3(kBlock), 0, s, d_colorvec, color_bytes, N, (int)C, shift, d_color8, d_st);
    tm.mark("colours (alloc + kernel)");
    hipLaunchKernelGGL(k_pb_check_colptr, dim3(std::max(1, (int)std::min<int64_t>((p->col1 - p->col0 + kBlock - 1) / kBlock, (int64_t)p->ctx->num_cus * 16))),
                       dim3(kBlock), 0, s, d_colptr, idx_bytes, idx_base, p->col0, p->col1, e0, e1, d_st);
    tm.mark("colptr check");
    const size_t padded = (size_t)((std::max<int64_t>(nloc, 1) + kListPad - 1) / kListPad * kListPad);
    // tile size: the host builder's rule (try_window_plan).  Every candidate is ONE fused pass (expand + extent + codes);
    // the statistics the rule needs are computed on the host from the descriptors (48 B per tile).
    const char *ft = getenv("FDJAC_WIN_TILE");
    const int force_t = (ft && *ft) ? atoi(ft) : 0;
    const int force_w = (fw && *fw) ? atoi(fw) : -1;
    const bool prefer_small = true;   // (the host builder's rule, try_window_plan)
    int4 *d_wt = nullptr;
    uint16_t *d_code = nullptr;
    if (hipMalloc((void **)&d_wt, sizeof(int4) * 3 * (padded / 512)) != hipSuccess) { (void)hipFree(d_color8); return PBR_DECLINED; }
    if (hipMalloc((void **)&d_code, sizeof(uint16_t) * padded) != hipSuccess) { (void)hipFree(d_color8); (void)hipFree(d_wt); return PBR_DECLINED; }
    int bestT = 0;
    struct HostStats { int max_slots = 0, max_ncol = 0; double elems = 0; };
    HostStats best;
    std::vector<int4> wt;
    PbStats fin;
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
#define FD_PB_TILES(TT)                                                                                                      \
        hipLaunchKernelGGL((k_pb_tiles<true, TT>), dim3((unsigned)ntiles), dim3(kBlock), 0, s, d_colptr, d_rowval, idx_bytes, \
                           idx_base, p->col0, p->col1, e0, nloc, p->M, d_color8, ntiles, d_wt, d_code, d_st)
        if (T == 2048) FD_PB_TILES(2048); else if (T == 1024) FD_PB_TILES(1024); else FD_PB_TILES(512);
#undef FD_PB_TILES
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
        const size_t lds = window_lds_bytes(p->dma, p->fdtype, cur.max_slots, cur.max_ncol);
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
    if (declined || !bestT) { (void)hipFree(d_color8); (void)hipFree(d_wt); (void)hipFree(d_code); return PBR_DECLINED; }
    const int64_t ntiles = (int64_t)(padded / (size_t)bestT);
    if ((int64_t)wt.size() != 3 * ntiles) {   // the accepted tile size is not the one of the last pass: run it again
#define FD_PB_TILES(TT)                                                                                                      \
        hipLaunchKernelGGL((k_pb_tiles<true, TT>), dim3((unsigned)ntiles), dim3(kBlock), 0, s, d_colptr, d_rowval, idx_bytes, \
                           idx_base, p->col0, p->col1, e0, nloc, p->M, d_color8, ntiles, d_wt, d_code, d_st)
        if (bestT == 2048) FD_PB_TILES(2048); else if (bestT == 1024) FD_PB_TILES(1024); else FD_PB_TILES(512);
#undef FD_PB_TILES
        wt.resize((size_t)(3 * ntiles));
        if (hipMemcpyAsync(wt.data(), d_wt, sizeof(int4) * wt.size(), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            (void)hipFree(d_color8); (void)hipFree(d_wt); (void)hipFree(d_code);
            return PBR_DECLINED;
        }
    }
    // periodic codes: the period is found on three sample tiles on the host (a few KB), every tile is tested on the device
    int P = 0, S = 0, magic = 0;
    {
        const char *fp = getenv("FDJAC_WIN_PERIODIC");
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
    // (the descriptors are already on the host: row strips are planned from them)
    p->C = C;
    p->color8 = true;
    p->d_color = d_color8;
    p->has_none = (fin.flags & PB_NONE) != 0;
    p->nnz_local = nloc;
    p->row0 = fin.row_max >= 0 ? fin.row_min : 0;
    p->row1 = fin.row_max >= 0 ? (int64_t)fin.row_max + 1 : 0;
    p->window = true;
    p->win_tile = bestT;
    p->win_pairs = best.max_slots / 2;
    p->win_ncol = best.max_ncol;
    p->win_overread = best.elems / (double)std::max<int64_t>(nloc, 1);
    p->win_per_P = P; p->win_per_S = P ? S : 0; p->win_per_magic = P ? magic : 0;
    p->d_wtiles = d_wt;
    p->d_wcode = d_code;
    plan_row_strips(p, wt, (size_t)ntiles);
    {
        const char *fc = getenv("FDJAC_EPS_CYCLIC");
        const bool cyc = !(fin.flags & (PB_NOT_CYCLIC | PB_NONE)) && !(fc && *fc && atoi(fc) == 0) &&
                         p->fdtype != FD_COMPLEX;   // (the complex step has no step-size reduction)
        p->cyc_C = cyc ? (int)C : 0;
        p->cyc_shift = cyc ? shift : 0;
    }
    p->built_on_device = true;
    tm.mark("descriptors to host");
    *rc_out = alloc_scratch(p, std::vector<int32_t>());     // (empty colour list: the cyclic test above stands)
    tm.mark("scratch allocation");
    return PBR_DONE;
}

}  // namespace fdjac
