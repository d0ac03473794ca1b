// Plan construction ON THE DEVICE for the common-pattern CSC path (fd_plan_create_csc / fd_plan_create_csc_device).
//
// What a plan compiles is the per-call pattern work of the reference -- the O(nnz) pattern comparison, the O(N) colour
// scan per colour and the colptr walk of `_colorediteration!` (src/jacobians.jl:524-535,547;
// ext/FiniteDiffSparseArraysExt.jl:38-47,51-52) -- into tile descriptors and 16-bit entry codes.  The host builder
// (fdjac_api.hip: csc_common / try_window_plan) does that with serial loops over nnz (0.3 s at nnz = 3*10^7).  Here the
// same arrays are produced by kernels:
//   k_pb_colmax / k_pb_colors   colorvec (Int32 / Int64, 1-based) -> 0-based uint8 colours, C = maximum(colorvec),
//                               "some column has no colour", "colours are cyclic" (wave ballots)
//   k_pb_expand                 colptr / rowval -> per stored entry (row, colour of its column), down-converted to
//                               int32 / uint8 (the index down-conversion SURVEY section 7 asks for), range-checked
//   k_pb_tiles                  one workgroup per tile: row / colour extent by wave reductions, the tile descriptor, the
//                               entry codes; grid-wide maxima and sums by atomics
//   k_pb_periodic               which tiles repeat their codes with the plan-wide period (ballot over the tile)
// The pattern may already live on the device (fd_plan_create_csc_device: nothing crosses PCIe) or is uploaded raw.
// Patterns the device builder does not handle (tiles that need several row windows or a sort: scattered stencils; more
// than 8 colours; forced kernel variants) make it step aside -- the host builder then runs as before.  The host builder
// is also the CHECKER: tests build every plan both ways (FDJAC_PLAN_DEVICE=0/1) and compare the plan arrays bit for bit
// (fd_plan_checksum).
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

// per stored entry of the local columns [col0, col1): 0-based int32 row and the uint8 colour of its column
__global__ void __launch_bounds__(kBlock) k_pb_expand(const void *__restrict__ colptr, const void *__restrict__ rowval, int ib,
                                                      int base, int64_t col0, int64_t col1, int64_t e0, int64_t e1, int64_t M,
                                                      const uint8_t *__restrict__ color8, int32_t *__restrict__ rows,
                                                      uint8_t *__restrict__ nzc, PbStats *st)
{
    unsigned flags = 0;
    int rmin = 0x7fffffff, rmax = -1;
    for (int64_t j = col0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < col1; j += (int64_t)gridDim.x * kBlock) {
        const int64_t a = pb_load(colptr, ib, j) - base, b = pb_load(colptr, ib, j + 1) - base;
        if (!(a <= b && a >= e0 && b <= e1)) { flags |= PB_BAD_COLPTR; continue; }
        const uint8_t c = color8[j];
        for (int64_t q = a; q < b; ++q) {
            const int64_t r = pb_load(rowval, ib, q) - base;
            if (r < 0 || r >= M) { flags |= PB_BAD_ROW; continue; }
            rows[q - e0] = (int32_t)r;
            nzc[q - e0] = c;
            rmin = (int)r < rmin ? (int)r : rmin;
            rmax = (int)r > rmax ? (int)r : rmax;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int a = __shfl_down(rmin, off, 64), b = __shfl_down(rmax, off, 64);
        rmin = a < rmin ? a : rmin;
        rmax = b > rmax ? b : rmax;
    }
    if ((threadIdx.x & 63) == 0) {
        if (rmax >= 0) { atomicMin(&st->row_min, rmin); atomicMax(&st->row_max, rmax); }
    }
    if (__builtin_amdgcn_ballot_w64(flags != 0)) {
        unsigned wf = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (__builtin_amdgcn_ballot_w64((flags >> b) & 1)) wf |= 1u << b;
        if ((threadIdx.x & 63) == 0) atomicOr(&st->flags, wf);
    }
}

__global__ void __launch_bounds__(kBlock) k_pb_pad(int32_t *__restrict__ rows, uint8_t *__restrict__ nzc, int64_t n, int64_t padded)
{
    const int64_t i = n + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < padded) { rows[i] = 0; nzc[i] = 0xFE; }
}

// One workgroup per tile of T entries: extent of the coloured entries -> descriptor (+ codes).  Reproduces
// try_window_plan's build_windows for tiles whose rows form ONE window; other tiles raise PB_NEED_SORT.
template <bool CODES>
__global__ void __launch_bounds__(kBlock) k_pb_tiles(const int32_t *__restrict__ rows, const uint8_t *__restrict__ nzc, int T,
                                                     int64_t ntiles, int4 *__restrict__ wt, uint16_t *__restrict__ code,
                                                     PbStats *st)
{
    const int64_t t = blockIdx.x;
    if (t >= ntiles) return;
    const int64_t b0 = t * T;
    int rmin = 0x7fffffff, rmax = -1, cmin = 0x7fffffff, cmax = -1, cnt = 0;
    for (int q = threadIdx.x; q < T; q += kBlock) {
        const int c = nzc[b0 + q];
        if (c >= 0xFE) continue;                     // no colour / padding: loads nothing
        const int r = rows[b0 + q];
        rmin = r < rmin ? r : rmin; rmax = r > rmax ? r : rmax;
        cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax;
        ++cnt;
    }
    __shared__ int s_red[kBlock / 64][5];
    __shared__ int s_tile[5];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int a = __shfl_down(rmin, off, 64), b = __shfl_down(rmax, off, 64), c = __shfl_down(cmin, off, 64),
                  d = __shfl_down(cmax, off, 64), e = __shfl_down(cnt, off, 64);
        rmin = a < rmin ? a : rmin; rmax = b > rmax ? b : rmax;
        cmin = c < cmin ? c : cmin; cmax = d > cmax ? d : cmax; cnt += e;
    }
    if ((threadIdx.x & 63) == 0) {
        int *r = s_red[threadIdx.x >> 6];
        r[0] = rmin; r[1] = rmax; r[2] = cmin; r[3] = cmax; r[4] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            rmin = s_red[w][0] < rmin ? s_red[w][0] : rmin; rmax = s_red[w][1] > rmax ? s_red[w][1] : rmax;
            cmin = s_red[w][2] < cmin ? s_red[w][2] : cmin; cmax = s_red[w][3] > cmax ? s_red[w][3] : cmax;
            cnt += s_red[w][4];
        }
        int wr0 = 0, pairs = 0, nwin = 0, ncol = 0;
        unsigned flags = 0;
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
        atomicMax(&st->max_slots, 2 * pairs);
        atomicMax(&st->max_ncol, ncol);
        atomicAdd(&st->elems, (unsigned long long)(2 * pairs) * (unsigned long long)ncol);
    }
    if (!CODES) return;
    __syncthreads();
    const int wr0 = s_tile[0], c0 = s_tile[1];
    for (int q = threadIdx.x; q < T; q += kBlock) {
        const int c = nzc[b0 + q];
        uint16_t cd;
        if (c == 0xFE) cd = 0x8000;
        else if (c == 0xFF) cd = 0x4000;
        else cd = (uint16_t)((rows[b0 + q] - wr0) | ((c - c0) << 11));
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
    if (threadIdx.x == 0) {
        regular[t] = s_bad ? 0 : 1;
        if (!s_bad) atomicAdd(&st->regular, 1u);
    }
}

__global__ void __launch_bounds__(kBlock) k_pb_set_regular(int4 *__restrict__ wt, const uint8_t *__restrict__ regular, int64_t ntiles)
{
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t < ntiles && regular[t]) wt[3 * t].w |= 0x100;
}

// outcome of the device builder
enum { PBR_DONE = 0, PBR_DECLINED = 1 };

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
    const int shift = h.first_color >= 1 ? (int)(h.first_color - 1) : 0;
    uint8_t *d_color8 = nullptr;
    if (hipMalloc((void **)&d_color8, (size_t)N) != hipSuccess) return PBR_DECLINED;
    hipLaunchKernelGGL(k_pb_colors, dim3(gN), dim3(kBlock), 0, s, d_colorvec, color_bytes, N, (int)C, shift, d_color8, d_st);
    // entries
    const size_t padded = (size_t)((std::max<int64_t>(nloc, 1) + kListPad - 1) / kListPad * kListPad);
    int32_t *d_rows = nullptr;
    uint8_t *d_nzc = nullptr;
    if (hipMalloc((void **)&d_rows, sizeof(int32_t) * padded) != hipSuccess) { (void)hipFree(d_color8); return PBR_DECLINED; }
    tmp.add(d_rows);
    if (hipMalloc((void **)&d_nzc, padded) != hipSuccess) { (void)hipFree(d_color8); return PBR_DECLINED; }
    tmp.add(d_nzc);
    const int64_t ncols = p->col1 - p->col0;
    const int gC = (int)std::min<int64_t>((ncols + kBlock - 1) / kBlock, (int64_t)1 << 20);
    hipLaunchKernelGGL(k_pb_expand, dim3(std::max(gC, 1)), dim3(kBlock), 0, s, d_colptr, d_rowval, idx_bytes, idx_base, p->col0, p->col1,
                       e0, e1, p->M, d_color8, d_rows, d_nzc, d_st);
    if ((int64_t)padded > nloc)
        hipLaunchKernelGGL(k_pb_pad, dim3((unsigned)(((int64_t)padded - nloc + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d_rows, d_nzc,
                           nloc, (int64_t)padded);
    // tile size: the host builder's rule (try_window_plan)
    const char *ft = getenv("FDJAC_WIN_TILE");
    const int force_t = (ft && *ft) ? atoi(ft) : 0;
    const int force_w = (fw && *fw) ? atoi(fw) : -1;
    const bool prefer_small = nloc < (int64_t)2048 * 24 * std::max(p->ctx->num_cus, 1);
    int4 *d_wt = nullptr;
    if (hipMalloc((void **)&d_wt, sizeof(int4) * 3 * (padded / 512)) != hipSuccess) { (void)hipFree(d_color8); return PBR_DECLINED; }
    int bestT = 0;
    PbStats best;
    auto stats_reset = [&](PbStats &dst) {   // keep colour / validation results, clear the per-pass tile statistics
        dst.max_slots = 0; dst.max_ncol = 0; dst.elems = 0; dst.regular = 0;
        dst.flags &= ~(unsigned)(PB_NEED_SORT | PB_TOO_MANY_COL | PB_TOO_WIDE);
    };
    bool declined = false, bad = false;
    for (int T : {2048, 1024, 512}) {
        if (force_t && T != force_t) continue;
        if (!force_t && T == 2048 && prefer_small) continue;
        const int64_t ntiles = (int64_t)(padded / (size_t)T);
        PbStats cur;
        if (hipMemcpyAsync(&cur, d_st, sizeof cur, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { declined = true; break; }
        if (cur.flags & (PB_BAD_ROW | PB_BAD_COLPTR)) { bad = true; break; }
        if (cur.flags & PB_COLOR_BIG) { declined = true; break; }
        stats_reset(cur);
        (void)hipMemcpyAsync(d_st, &cur, sizeof cur, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL((k_pb_tiles<false>), dim3((unsigned)ntiles), dim3(kBlock), 0, s, d_rows, d_nzc, T, ntiles, d_wt, (uint16_t *)nullptr, d_st);
        if (hipMemcpyAsync(&cur, d_st, sizeof cur, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { declined = true; break; }
        if (cur.flags & PB_NEED_SORT) { declined = true; break; }                     // scattered pattern: the host builder's job
        if ((cur.flags & (PB_TOO_MANY_COL | PB_TOO_WIDE)) || cur.max_slots <= 0) continue;   // the host builder rejects this T too
        const size_t lds = window_lds_bytes(p->dma, p->fdtype, cur.max_slots, cur.max_ncol);
        if (lds > (size_t)kWinMaxLds) continue;
        const double overread = (double)cur.elems / (double)std::max<int64_t>(nloc, 1);
        if (!(overread <= 1.25 || force_w == 1)) continue;
        bestT = T;
        best = cur;
        if (lds <= (size_t)32 * 1024 || T == 1024) break;
    }
    if (bad) {
        (void)hipFree(d_color8); (void)hipFree(d_wt);
        set_error("colptr / rowval are inconsistent (an entry outside 1..%lld or colptr not monotone)", (long long)p->M);
        *rc_out = FD_ERR_SHAPE;
        return PBR_DONE;
    }
    if (declined || !bestT) { (void)hipFree(d_color8); (void)hipFree(d_wt); return PBR_DECLINED; }
    // final pass with codes
    const int64_t ntiles = (int64_t)(padded / (size_t)bestT);
    uint16_t *d_code = nullptr;
    if (hipMalloc((void **)&d_code, sizeof(uint16_t) * padded) != hipSuccess) { (void)hipFree(d_color8); (void)hipFree(d_wt); return PBR_DECLINED; }
    {
        PbStats cur = best;
        stats_reset(cur);
        (void)hipMemcpyAsync(d_st, &cur, sizeof cur, hipMemcpyHostToDevice, s);
    }
    hipLaunchKernelGGL((k_pb_tiles<true>), dim3((unsigned)ntiles), dim3(kBlock), 0, s, d_rows, d_nzc, bestT, ntiles, d_wt, d_code, d_st);
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
                PbStats cur;
                if (hipMemcpyAsync(&cur, d_st, sizeof cur, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess &&
                    2 * (int64_t)cur.regular >= ntiles)
                    hipLaunchKernelGGL(k_pb_set_regular, dim3((unsigned)((ntiles + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d_wt, d_reg, ntiles);
                else
                    P = 0;
            } else {
                P = 0;
            }
        }
    }
    // the descriptors come back to the host once (48 B per tile): row strips are planned from them
    std::vector<int4> wt((size_t)(3 * ntiles));
    PbStats fin;
    if (hipMemcpyAsync(wt.data(), d_wt, sizeof(int4) * wt.size(), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(&fin, d_st, sizeof fin, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        (void)hipFree(d_color8); (void)hipFree(d_wt); (void)hipFree(d_code);
        return PBR_DECLINED;
    }
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
    p->win_overread = (double)best.elems / (double)std::max<int64_t>(nloc, 1);
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
    *rc_out = alloc_scratch(p, std::vector<int32_t>());     // (empty colour list: the cyclic test above stands)
    return PBR_DONE;
}

}  // namespace fdjac
