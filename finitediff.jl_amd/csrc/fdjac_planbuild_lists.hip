// Index-list plans built ON THE DEVICE: what finish_list_plan (fdjac_plan_list.hip) compiles for a common-pattern CSC whose tiles cannot
// be described by row windows -- scattered patterns (3-D stencils, random sparsity): the sorted-gather lists, or, when the storage
// order already gathers coherently, the plain lists.  Included by fdjac_planbuild.hip's caller after it (same namespace).
//
//   k_pbl_tiles   one workgroup per tile of kSortTile stored entries, straight from colptr / rowval:
//                 * every entry finds its column (binary search inside the tile's column range), down-converts its row, takes its
//                   column's colour;
//                 * the tile sorted by (colour, row, position) -- bitonic network over 51-bit keys in LDS -- gives the sorted rows,
//                   the sorted colours and each sorted entry's output position (d_rowval, d_nzcolor, d_spos);
//                 * the coloured rows sorted by value -- a second network, stopped after the 512-, 1024- and 2048-entry stages --
//                   answer, for every candidate window tile size, the question the host's try_window_plan asks first: can this tile
//                   be described by at most kWinMaxWin row windows of at most 2048 slots, and how many slots and colours would
//                   its windows hold?  (16 B per tile and size; the host applies try_window_plan's LDS and over-read limits to the
//                   sums.  A pattern where every size is rejected gets index lists -- the only case built here);
//                 * from the fully sorted rows, the runs of rows whose f(x) the forward-difference kernel stages in LDS (d_fxwin):
//                   boundaries where consecutive rows differ by more than 16, ranked by a block-wide scan; more than kFxWin runs:
//                   the kFxWin - 1 widest gaps survive (7 rounds of a block-wide maximum);
//                 * the statistics the host loops keep: row extent, reach max |row - column|, each tile's first column.
// The decisions stay where they were: the coherence estimate (k_pb_coherence), the thresholds, the tile order of far-band patterns
// (far_band_tile_order on 8 B per tile) are the host builder's own code.  The host builder is the checker: tests build every plan
// both ways and compare all list arrays bit for bit (fd_plan_checksum).

namespace fdjac {

struct PblStats {
    unsigned int flags;               // PB_BAD_ROW
    int row_min, row_max;             // over the local stored entries
    unsigned int eligible;            // tiles whose f(x) runs fit the staging area
    unsigned long long reach;         // max |row - column| over the coloured entries
};

template <bool SORT>
__global__ void __launch_bounds__(kBlock) k_pbl_tiles(const void *__restrict__ colptr, const void *__restrict__ rowval, int ib, int base,
                                                      int64_t col0, int64_t col1, int64_t e0, int64_t nloc, int64_t M,
                                                      const uint8_t *__restrict__ color8, int tmask, int32_t *__restrict__ rows_out,
                                                      uint8_t *__restrict__ nzc_out, uint16_t *__restrict__ spos_out,
                                                      int32_t *__restrict__ fxw_out, int64_t *__restrict__ tcol_out,
                                                      int4 *__restrict__ wstat_out, PblStats *st)
{
    __shared__ int s_res[3][4];                       // per window tile size: {some tile impossible, max slots, max colours, sum of slots x colours}
    __shared__ unsigned long long s_key[kSortTile];
    __shared__ int s_row[kSortTile], s_val[kSortTile];
    __shared__ unsigned char s_col[kSortTile];
    __shared__ long long s_j[2];
    __shared__ int s_n[4], s_gaps[4], s_cmin[4], s_cmax[4], s_vmin[4], s_vmax[4];
    __shared__ unsigned long long s_last[4], s_start[4];
    __shared__ int s_wsum[kBlock / 64], s_cnt, s_sel[kFxWin];
    __shared__ unsigned long long s_wmax[kBlock / 64], s_pick;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t t = blockIdx.x, q0 = t * kSortTile;
    constexpr int kNone = 0x7fffffff;

    if (tid < 12) s_res[tid >> 2][tid & 3] = 0;
    // ---- the tile's column range, then every entry's column / row / colour
    if (tid < 2) {
        const int64_t q = tid == 0 ? q0 : std::min<int64_t>(q0 + kSortTile, nloc) - 1;
        int64_t lo = col0, hi = col1;
        if (q >= 0 && q < nloc) {
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if (pb_load(colptr, ib, mid) - base - e0 <= q) lo = mid; else hi = mid;
            }
        }
        s_j[tid] = lo;
    }
    __syncthreads();
    int rmin = kNone, rmax = -1;
    long long reach = 0;
    bool bad = false;
    for (int k = tid; k < kSortTile; k += kBlock) {
        const int64_t q = q0 + k;
        int r = 0, c8 = 0xFE;
        if (q < nloc) {
            int64_t lo = s_j[0], hi = s_j[1] + 1;                 // cp(lo) <= q < cp(hi)
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if (pb_load(colptr, ib, mid) - base - e0 <= q) lo = mid; else hi = mid;
            }
            const int64_t rr = pb_load(rowval, ib, e0 + q) - base;
            bad = bad || rr < 0 || rr >= M;
            r = (int)rr;
            c8 = color8[lo];
            rmin = min(rmin, r); rmax = max(rmax, r);
            if (c8 != 0xFF) reach = max(reach, (long long)(rr > lo ? rr - lo : lo - rr));
            if (k == 0) tcol_out[t] = lo - col0;
        } else if (k == 0) {
            tcol_out[t] = (col1 - col0) - 1;
        }
        s_row[k] = r;
        s_col[k] = (unsigned char)c8;
        s_val[k] = c8 < 0xFE ? r : kNone;
        // (the host's order: colours, then the columns without colour, then the padding -- the stored bytes are 0xFF / 0xFE)
        const unsigned long long cc = c8 < 0xFE ? (unsigned long long)c8 : (c8 == 0xFF ? 0xFEull : 0xFFull);
        s_key[k] = (cc << 43) | ((unsigned long long)(unsigned)r << 11) | (unsigned long long)k;
    }
    for (int o = 32; o > 0; o >>= 1) {
        rmin = min(rmin, __shfl_xor(rmin, o, 64));
        rmax = max(rmax, __shfl_xor(rmax, o, 64));
        reach = max(reach, __shfl_xor(reach, o, 64));
    }
    if (lane == 0) {
        if (rmax >= 0) { atomicMin(&st->row_min, rmin); atomicMax(&st->row_max, rmax); }
        if (reach > 0) atomicMax(&st->reach, (unsigned long long)reach);
    }
    if (__builtin_amdgcn_ballot_w64(bad) && lane == 0) atomicOr(&st->flags, (unsigned)PB_BAD_ROW);
    __syncthreads();

    // ---- (colour, row, position) order
    if (SORT) {
        for (int k = 2; k <= kSortTile; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < kSortTile; i += kBlock) {
                    const int p = i ^ j;
                    if (p > i) {
                        const unsigned long long ka = s_key[i], kb = s_key[p];
                        if ((ka > kb) == ((i & k) == 0)) { s_key[i] = kb; s_key[p] = ka; }
                    }
                }
                __syncthreads();
            }
        for (int k = tid; k < kSortTile; k += kBlock) {
            const int src = (int)(s_key[k] & 2047u);
            rows_out[q0 + k] = s_row[src];
            nzc_out[q0 + k] = s_col[src];
            spos_out[q0 + k] = (uint16_t)src;
        }
    } else {
        for (int k = tid; k < kSortTile; k += kBlock) {
            rows_out[q0 + k] = s_row[k];
            nzc_out[q0 + k] = s_col[k];
        }
    }

    // ---- the coloured rows by value; after the stages of 512 / 1024 / 2048 entries: the window test of that tile size
    for (int k = 2; k <= kSortTile; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < kSortTile; i += kBlock) {
                const int p = i ^ j;
                if (p > i) {
                    const int a = s_val[i], b = s_val[p];
                    if ((a > b) == ((i & k) == 0)) { s_val[i] = b; s_val[p] = a; }
                }
            }
            __syncthreads();
        }
        const int ti = k == 2048 ? 0 : k == 1024 ? 1 : k == 512 ? 2 : -1;
        if (ti < 0 || !((tmask >> ti) & 1)) continue;
        const int T = k, nsub = kSortTile / T;
        if (tid < 4) {
            s_n[tid] = 0; s_gaps[tid] = 0; s_cmin[tid] = kNone; s_cmax[tid] = -1; s_vmin[tid] = kNone; s_vmax[tid] = -1;
            s_last[tid] = 0; s_start[tid] = 0;
        }
        __syncthreads();
        for (int m = tid; m < kSortTile; m += kBlock) {
            const int sub = m / T, im = m - sub * T;
            const bool asc = T == kSortTile || (m & T) == 0;        // (the network leaves the blocks sorted in alternating directions)
            const int idx = asc ? m : sub * T + (T - 1 - im);
            const int v = s_val[idx];
            if (v != kNone) {
                atomicAdd(&s_n[sub], 1);
                atomicMin(&s_vmin[sub], v);
                atomicMax(&s_vmax[sub], v);
                if (im > 0) {
                    const int pv = s_val[asc ? idx - 1 : idx + 1];
                    if (v - pv > kWinGap) {
                        atomicAdd(&s_gaps[sub], 1);
                        atomicAdd(&s_last[sub], (unsigned long long)(pv >> 1));
                        atomicAdd(&s_start[sub], (unsigned long long)(v >> 1));
                    }
                }
            }
            const int c = s_col[m];
            if (c < 0xFE) { atomicMin(&s_cmin[sub], c); atomicMax(&s_cmax[sub], c); }
        }
        __syncthreads();
        if (tid < nsub && s_n[tid] > 0) {
            // try_window_plan's build_windows for this tile: one tight window, else windows split at gaps of more than kWinGap rows
            const int lo = s_vmin[tid], hi = s_vmax[tid], ncol = s_cmax[tid] - s_cmin[tid] + 1;
            bool fail = ncol > kWinMaxCol;
            long long pairs;
            if (hi - lo < 2 * kWinGap || (double)(hi - lo + 2) * ncol <= 1.25 * (double)s_n[tid]) {
                pairs = (hi - (lo & ~1)) / 2 + 1;
            } else {
                const int nwin = 1 + s_gaps[tid];
                fail = fail || nwin > kWinMaxWin;
                pairs = (long long)(s_last[tid] + (unsigned long long)(hi >> 1)) - (long long)(s_start[tid] + (unsigned long long)(lo >> 1)) + nwin;
            }
            fail = fail || 2 * pairs > 2048;
            if (fail) {
                atomicOr(&s_res[ti][0], 1);
            } else {
                atomicMax(&s_res[ti][1], (int)(2 * pairs));
                atomicMax(&s_res[ti][2], ncol);
                atomicAdd(&s_res[ti][3], (int)(2 * pairs) * ncol);
            }
        }
        __syncthreads();
    }
    if (tid < 3) wstat_out[3 * t + tid] = make_int4(s_res[tid][0], s_res[tid][1], s_res[tid][2], s_res[tid][3]);

    // ---- runs of rows for the f(x) staging of the forward-difference kernel (finish_list_plan, "f(x) through LDS")
    if (!SORT || !fxw_out) return;
    int32_t *w = fxw_out + (size_t)t * 2 * kFxWin;
    int *s_bnd = (int *)s_key;                                        // (the keys are written out: their LDS holds the run boundaries)
    int nv = 0;
    {
        int mine = 0;
        for (int m = tid; m < kSortTile; m += kBlock) mine += s_val[m] != kNone;
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
        if (lane == 0) s_wsum[wave] = mine;
        __syncthreads();
        for (int i = 0; i < kBlock / 64; ++i) nv += s_wsum[i];
        __syncthreads();
    }
    if (nv == 0) {
        if (tid < 2 * kFxWin) w[tid] = tid == 0 ? -1 : 0;
        return;
    }
    // boundaries in ascending position: thread tid owns the positions 8 tid .. 8 tid + 7
    constexpr int PER = kSortTile / kBlock;
    int flags = 0, cnt = 0;
    for (int i = 0; i < PER; ++i) {
        const int m = tid * PER + i;
        if (m >= 1 && m < nv && s_val[m] - s_val[m - 1] > 16) { flags |= 1 << i; ++cnt; }
    }
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int rank = incl - cnt, nb = 0;
    for (int i = 0; i < kBlock / 64; ++i) { if (i < wave) rank += s_wsum[i]; nb += s_wsum[i]; }
    for (int i = 0; i < PER; ++i)
        if (flags & (1 << i)) s_bnd[rank++] = tid * PER + i;
    __syncthreads();
    if (nb + 1 > kFxWin) {
        // close the smallest gaps (the first of equal ones first) until kFxWin runs are left: the kFxWin - 1 largest by (gap, position) stay
        unsigned long long below = ~0ull;
        for (int round = 0; round < kFxWin - 1; ++round) {
            unsigned long long best = 0;
            for (int i = tid; i < nb; i += kBlock) {
                const int m = s_bnd[i];
                const unsigned long long key = ((unsigned long long)(unsigned)(s_val[m] - s_val[m - 1]) << 32) | (unsigned)m;
                if (key < below && key > best) best = key;
            }
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long v = __shfl_xor(best, o, 64);
                best = v > best ? v : best;
            }
            if (lane == 0) s_wmax[wave] = best;
            __syncthreads();
            if (tid == 0) {
                unsigned long long b = 0;
                for (int i = 0; i < kBlock / 64; ++i) b = s_wmax[i] > b ? s_wmax[i] : b;
                s_pick = b;
                s_sel[round] = (int)(b & 0xFFFFFFFFu);
            }
            __syncthreads();
            below = s_pick;
        }
        if (tid == 0) {
            for (int a = 1; a < kFxWin - 1; ++a) {                    // (positions ascending)
                const int v = s_sel[a];
                int b = a - 1;
                while (b >= 0 && s_sel[b] > v) { s_sel[b + 1] = s_sel[b]; --b; }
                s_sel[b + 1] = v;
            }
            s_cnt = kFxWin - 1;
        }
    } else if (tid == 0) {
        for (int i = 0; i < nb; ++i) s_sel[i] = s_bnd[i];
        s_cnt = nb;
    }
    __syncthreads();
    if (tid == 0) {
        const int nr = s_cnt + 1;
        long long total = 0;
        int first[kFxWin], len[kFxWin];
        for (int i = 0; i < nr; ++i) {
            const int a = i == 0 ? 0 : s_sel[i - 1], b = i + 1 < nr ? s_sel[i] - 1 : nv - 1;
            first[i] = s_val[a];
            len[i] = s_val[b] - s_val[a] + 1;
            total += len[i];
        }
        const bool ok = total <= kFxRows;
        for (int i = 0; i < kFxWin; ++i) {
            w[2 * i] = ok && i < nr ? first[i] : (i == 0 && !ok ? -1 : 0);
            w[2 * i + 1] = ok && i < nr ? len[i] : 0;
        }
        if (ok) atomicAdd(&st->eligible, 1u);
    }
}

// The gather-coherence estimate that decides "scattered" (finish_list_plan's, on the same sample of tiles): p->lines_direct / lines_sorted.
static bool pb_coherence_sample(fd_plan *p, const void *d_colptr, const void *d_rowval, int ib, int base, int64_t e0, int64_t nloc,
                                const uint8_t *d_color8)
{
    hipStream_t s = p->ctx->stream;
    PbTemps tmp;
    const size_t padded = (size_t)((std::max<int64_t>(nloc, 1) + kListPad - 1) / kListPad * kListPad);
    const size_t ntiles = padded / kSortTile, step = std::max<size_t>(1, ntiles / 64), nsamp = (ntiles + step - 1) / step;
    int32_t *d_sr = nullptr, *d_sc = nullptr;
    if (hipMalloc((void **)&d_sr, sizeof(int32_t) * nsamp * kSortTile) != hipSuccess) return false;
    tmp.add(d_sr);
    if (hipMalloc((void **)&d_sc, sizeof(int32_t) * nsamp * kSortTile) != hipSuccess) return false;
    tmp.add(d_sc);
    hipLaunchKernelGGL(k_pb_sample_expand, dim3((unsigned)nsamp), dim3(kBlock), 0, s, d_colptr, d_rowval, ib, base, p->col0, p->col1, e0,
                       nloc, d_color8, (int64_t)step, d_sr, d_sc);
    int *d_cnt = nullptr;
    if (hipMalloc((void **)&d_cnt, sizeof(int) * 2 * nsamp) != hipSuccess) return false;
    tmp.add(d_cnt);
    hipLaunchKernelGGL(k_pb_coherence, dim3((unsigned)nsamp), dim3(kBlock), 0, s, d_sr, d_sc, d_cnt);
    std::vector<int> cnt(2 * nsamp);
    if (hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return false;
    // (the host's gather_coherence: sums of per-gather line counts / number of gathers; integers, so the same doubles)
    double ld = 0, ls = 0;
    for (size_t t = 0; t < nsamp; ++t) { ld += (double)cnt[2 * t]; ls += (double)cnt[2 * t + 1]; }
    const size_t ninstr = nsamp * 2 * (kSortTile / 128);
    p->lines_direct = ld / (double)std::max<size_t>(ninstr, 1);
    p->lines_sorted = ls / (double)std::max<size_t>(ninstr, 1);
    return true;
}

// PBR_DONE: the list arrays of the plan are set (d_rowval, d_nzcolor, and for a sorted-gather plan d_spos / d_fxwin / d_tile_order);
// the caller finishes the plan (colours, cyclic test, scratch).  PBR_DECLINED: some window tile size might work -- the host decides.
static int device_build_lists(fd_plan *p, const void *d_colptr, const void *d_rowval, int ib, int base, int64_t e0, int64_t nloc,
                              const uint8_t *d_color8, int64_t C, PbTimer &tm, int *row0_out, int *row1_out, int *rc_out)
{
    hipStream_t s = p->ctx->stream;
    const char *fw1 = fdjac::test_switch("FDJAC_WINDOW");
    if (fw1 && *fw1) return PBR_DECLINED;                             // (forced kernel variants: the host builder)
    if (nloc < 1 || p->kind != K_CSC) return PBR_DECLINED;
    const size_t padded = (size_t)((std::max<int64_t>(nloc, 1) + kListPad - 1) / kListPad * kListPad);
    const size_t ntiles = padded / kSortTile;
    // finish_list_plan's "scattered"
    bool scattered = false;
    if (nloc >= 4 * kSortTile) {
        if (!(p->lines_direct > 0) && !pb_coherence_sample(p, d_colptr, d_rowval, ib, base, e0, nloc, d_color8)) return PBR_DECLINED;
        scattered = p->lines_direct > 16.0 && p->lines_direct > 1.5 * p->lines_sorted;
    }
    const char *fs = fdjac::test_switch("FDJAC_SORTED");
    bool sorted = scattered;
    if (fs && *fs) sorted = atoi(fs) != 0 && nloc >= 4 * kSortTile;
    // the window tile sizes try_window_plan would try
    const char *ft = fdjac::test_switch("FDJAC_WIN_TILE");
    const int force_t = (ft && *ft) ? atoi(ft) : 0;
    int tmask = 0;
    for (int T : {2048, 1024, 512}) {
        if (force_t && T != force_t) continue;
        if (!force_t && T == 2048 && sizeof(real_t) >= 8) continue;
        tmask |= 1 << (T == 2048 ? 0 : T == 1024 ? 1 : 2);
    }
    const bool want_fx = sorted && p->fdtype == FD_FORWARD;
    PbTemps tmp;                                                      // (freed on every return; ownership moves to the plan at the end)
    int32_t *d_rows = nullptr, *d_fxw = nullptr;
    uint8_t *d_nzc = nullptr;
    uint16_t *d_spos = nullptr;
    int64_t *d_tcol = nullptr;
    int4 *d_wstat = nullptr;
    PblStats *d_st = nullptr, h;
    if (hipMalloc((void **)&d_rows, sizeof(int32_t) * padded) != hipSuccess) return PBR_DECLINED;
    tmp.add(d_rows);
    if (hipMalloc((void **)&d_nzc, padded) != hipSuccess) return PBR_DECLINED;
    tmp.add(d_nzc);
    if (sorted) { if (hipMalloc((void **)&d_spos, sizeof(uint16_t) * padded) != hipSuccess) return PBR_DECLINED; tmp.add(d_spos); }
    if (want_fx) { if (hipMalloc((void **)&d_fxw, sizeof(int32_t) * 2 * kFxWin * ntiles) != hipSuccess) return PBR_DECLINED; tmp.add(d_fxw); }
    if (hipMalloc((void **)&d_tcol, sizeof(int64_t) * ntiles) != hipSuccess) return PBR_DECLINED;
    tmp.add(d_tcol);
    if (hipMalloc((void **)&d_wstat, sizeof(int4) * 3 * ntiles) != hipSuccess) return PBR_DECLINED;
    tmp.add(d_wstat);
    if (hipMalloc((void **)&d_st, sizeof(PblStats)) != hipSuccess) return PBR_DECLINED;
    tmp.add(d_st);
    memset(&h, 0, sizeof h);
    h.row_min = 0x7fffffff; h.row_max = -1;
    if (hipMemcpyAsync(d_st, &h, sizeof h, hipMemcpyHostToDevice, s) != hipSuccess) return PBR_DECLINED;
    if (sorted)
        hipLaunchKernelGGL(k_pbl_tiles<true>, dim3((unsigned)ntiles), dim3(kBlock), 0, s, d_colptr, d_rowval, ib, base, p->col0, p->col1, e0, nloc,
                           p->M, d_color8, tmask, d_rows, d_nzc, d_spos, d_fxw, d_tcol, d_wstat, d_st);
    else
        hipLaunchKernelGGL(k_pbl_tiles<false>, dim3((unsigned)ntiles), dim3(kBlock), 0, s, d_colptr, d_rowval, ib, base, p->col0, p->col1, e0, nloc,
                           p->M, d_color8, tmask, d_rows, d_nzc, d_spos, d_fxw, d_tcol, d_wstat, d_st);
    std::vector<int4> wstat(3 * ntiles);
    if (hipMemcpyAsync(&h, d_st, sizeof h, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(wstat.data(), d_wstat, sizeof(int4) * wstat.size(), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) return PBR_DECLINED;
    tm.mark("lists: tiles");
    if (h.flags & PB_BAD_ROW) {
        set_error("colptr / rowval are inconsistent (an entry outside 1..%lld)", (long long)p->M);
        *rc_out = FD_ERR_SHAPE;
        return PBR_DONE;
    }
    // try_window_plan's verdict on every tile size it would try (clustered windows, the pattern being `scattered` or not): a size
    // that passes means a row-window plan -- the host builds those
    for (int ti = 0; ti < 3; ++ti) {
        if (!((tmask >> ti) & 1)) continue;
        bool fail = false;
        int max_slots = 0, max_ncol = 0;
        double elems = 0;
        for (size_t t = 0; t < ntiles; ++t) {
            const int4 v = wstat[3 * t + (size_t)ti];
            fail = fail || v.x != 0;
            max_slots = std::max(max_slots, v.y);
            max_ncol = std::max(max_ncol, v.z);
            elems += (double)v.w;
        }
        if (fail || max_slots <= 0) continue;
        if (window_lds_bytes(p->fdtype, max_slots, max_ncol) > (size_t)kWinMaxLds) continue;
        const double overread = elems / (double)std::max<int64_t>(nloc, 1);
        if (overread <= 1.25 || (scattered && overread <= kWinMaxOverread)) return PBR_DECLINED;
    }
    std::vector<int32_t> order;
    if (sorted) {
        std::vector<int64_t> tcol(ntiles);
        if (hipMemcpyAsync(tcol.data(), d_tcol, sizeof(int64_t) * ntiles, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            return PBR_DECLINED;
        p->nnz_local = nloc;
        p->C = C;
        order = far_band_tile_order(p, tcol, (int64_t)h.reach, p->col1 - p->col0, ntiles);
        if (!order.empty() && dev_upload(&p->d_tile_order, order) != FD_OK) return PBR_DECLINED;
        tm.mark("lists: tile order");
    }
    p->sorted_gather = sorted;
    p->d_rowval = d_rows;
    p->d_nzcolor = d_nzc;
    p->d_spos = d_spos;
    if (want_fx && (size_t)h.eligible * 2 >= ntiles) p->d_fxwin = d_fxw;
    for (int i = 0; i < tmp.n; ++i)                                   // the plan owns them now
        if (tmp.ptrs[i] == (void *)d_rows || tmp.ptrs[i] == (void *)d_nzc || tmp.ptrs[i] == (void *)d_spos ||
            (p->d_fxwin && tmp.ptrs[i] == (void *)d_fxw)) tmp.ptrs[i] = nullptr;
    *row0_out = h.row_max >= 0 ? h.row_min : 0;
    *row1_out = h.row_max >= 0 ? h.row_max + 1 : 0;
    return PBR_DONE;
}

}  // namespace fdjac
