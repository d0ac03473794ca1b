// Plan compilation for index-list patterns on the HOST (the checker of the device builder, fdjac_planbuild.hip, and the builder of
// the patterns it declines): row windows (1-D / 2-D tiles), periodic entry codes, computed band descriptors, the verification of exact
// bands / 5-point stencils for the storing launches, colour-sorted gather tiles and their far-band order.  Replaces the per-call
// pattern work of src/jacobians.jl:524-535,547 and ext/FiniteDiffSparseArraysExt.jl:38-47,51-52.
// Included by fdjac_api.hip (inside namespace fdjac).
// Row windows (k_decompress_window).  Per tile of T entries: the rows its coloured entries touch, clustered
// into at most kWinMaxWin windows (a new window starts after a gap of more than kWinGap rows), and the range of
// colours.  The kernel loads every window of every colour of the tile densely, so the variant is used when
//   * every tile has <= kWinMaxWin windows, <= kWinMaxCol consecutive colours, <= 2048 window rows in total,
//   * the LDS tile (colours x window rows x 8 B) leaves several workgroups per CU, and
//   * the dense loads bring in at most 1.25 f! values per stored entry (banded patterns: exactly 1), or, for
//     patterns whose gathers are scattered anyway (5-point stencils: 3), at most kWinMaxOverread -- re-reads
//     that are served by the L2, traded for divergence-free 16-B loads (the gather kernels are TA-bound there).
// rows / nzc: row and colour (>= 0; -1 = column without colour, written as 0; -2 = padding, never written) of
// every output slot in storage order, padded to a multiple of kListPad.  Sets p->window on success.
// LDS of one workgroup of the row-window kernels: the differences of the tile's row windows, one array per colour.
static size_t window_lds_bytes(int fdtype, int max_slots, int max_ncol)
{
    (void)fdtype;
    const size_t wp = (((size_t)max_slots + 31) & ~(size_t)31) + 2;
    return wp * (size_t)max_ncol * sizeof(real_t) + sizeof(real_t) * (size_t)kWinMaxCol + 4 * (size_t)kW2Desc + kWinHeadBytes;
}

// single_only: accept a tile only if its rows form ONE window (what the device builder's k_pb_tiles can describe)
static int try_window_plan(fd_plan *p, const std::vector<int32_t> &rows, const std::vector<int32_t> &nzc, size_t padded,
                           bool scattered, bool single_only = false)
{
    int rc;
    struct WinBuild {
        bool ok = false;
        int T = 0, max_slots = 0, max_ncol = 0;
        double overread = 0;
        std::vector<int4> wt;          // 3 x int4 per tile: {cmin, ncol, pairs, nwin}, {rmin0,end0,rmin1,end1}, {rmin2,end2,rmin3,end3}
        std::vector<uint16_t> code;
    };
    auto build_windows = [&](int T, bool want_codes) {
        WinBuild w;
        w.T = T;
        const size_t ntiles = padded / (size_t)T;
        w.wt.assign(3 * ntiles, int4{0, 0, 0, 0});
        if (want_codes) w.code.assign(padded, (uint16_t)0x8000);
        std::vector<int32_t> rr;
        double elems = 0;
        for (size_t t = 0; t < ntiles; ++t) {
            const size_t b0 = t * (size_t)T;
            int32_t rmin = std::numeric_limits<int32_t>::max(), rmax = -1, cmin = rmin, cmax = -1;
            int64_t ncoloured = 0;
            for (size_t e = b0; e < b0 + (size_t)T; ++e) {
                if (nzc[e] < 0) continue;   // uncoloured / padding entries load nothing
                rmin = std::min(rmin, rows[e]); rmax = std::max(rmax, rows[e]);
                cmin = std::min(cmin, nzc[e]); cmax = std::max(cmax, nzc[e]);
                ++ncoloured;
            }
            int32_t wr[kWinMaxWin], wn[kWinMaxWin];   // first row (even), pairs
            int nwin = 0;
            if (rmax >= 0) {
                if (cmax - cmin + 1 > kWinMaxCol) return w;
                // one window is already tight (or short): no need to look at the rows again
                if (rmax - rmin < 2 * kWinGap || (double)(rmax - rmin + 2) * (cmax - cmin + 1) <= 1.25 * (double)ncoloured) {
                    wr[0] = rmin & ~1;
                    wn[0] = (rmax - wr[0]) / 2 + 1;
                    nwin = 1;
                } else {
                    if (single_only) return w;
                    rr.clear();
                    for (size_t e = b0; e < b0 + (size_t)T; ++e) if (nzc[e] >= 0) rr.push_back(rows[e]);
                    std::sort(rr.begin(), rr.end());
                    int32_t start = rr[0] & ~1, last = rr[0];
                    for (size_t k = 1; k <= rr.size(); ++k) {
                        if (k == rr.size() || rr[k] - last > kWinGap) {
                            if (nwin == kWinMaxWin) return w;
                            wr[nwin] = start;
                            wn[nwin] = (last - start) / 2 + 1;
                            ++nwin;
                            if (k < rr.size()) start = rr[k] & ~1;
                        }
                        if (k < rr.size()) last = rr[k];
                    }
                }
            }
            int pairs = 0;
            int32_t ends[kWinMaxWin];
            for (int k = 0; k < kWinMaxWin; ++k) {
                if (k < nwin) pairs += wn[k];
                ends[k] = pairs;                 // unused windows are empty: end == total
                if (k >= nwin) wr[k] = 0;
            }
            if (2 * pairs > 2048) return w;     // the slot field of the entry code has 11 bits
            const int ncol = rmax >= 0 ? cmax - cmin + 1 : 0;
            w.wt[3 * t] = int4{rmax >= 0 ? cmin : 0, ncol, pairs, nwin};
            w.wt[3 * t + 1] = int4{wr[0], ends[0], wr[1], ends[1]};
            w.wt[3 * t + 2] = int4{wr[2], ends[2], wr[3], ends[3]};
            w.max_slots = std::max(w.max_slots, 2 * pairs);
            w.max_ncol = std::max(w.max_ncol, ncol);
            elems += 2.0 * pairs * ncol;
            if (want_codes)
                for (size_t e = b0; e < b0 + (size_t)T; ++e) {
                    if (nzc[e] == -2) continue;                            // padding
                    if (nzc[e] < 0) { w.code[e] = 0x4000; continue; }      // column without colour
                    int k = 0;
                    while (!(rows[e] >= wr[k] && rows[e] < wr[k] + 2 * wn[k])) ++k;
                    const int slot = 2 * (k ? ends[k - 1] : 0) + (rows[e] - wr[k]);
                    w.code[e] = (uint16_t)(slot | ((nzc[e] - cmin) << 11));
                }
        }
        w.overread = elems / (double)std::max<int64_t>(p->nnz_local, 1);
        w.ok = w.max_slots > 0;
        return w;
    };
    {
        const char *fw = fdjac::test_switch("FDJAC_WINDOW"), *fs = fdjac::test_switch("FDJAC_SORTED");
        const int force_w = (fw && *fw) ? atoi(fw) : -1, force_s = (fs && *fs) ? atoi(fs) : -1;
        WinBuild best;
        if (force_w != 0 && force_s != 1) {
            const char *ft = fdjac::test_switch("FDJAC_WIN_TILE");   // test / tuning switch: force the tile size (2048, 1024 or 512)
            const int force_t = (ft && *ft) ? atoi(ft) : 0;
            // fewer than ~24 tiles of 2048 entries per CU: the half-size tile balances the launch better (tridiagonal
            // forward, same process: N = 10^6 14.6 -> 13.5 us, N = 3*10^6 30.7 -> 30.0 us, N = 10^7 equal)
            // round 2, N = 10^7 as well (two boxes, separate processes, 40 steps each: 111.8 / 113.2 us with 2048-entry tiles,
            // 109.4 / 108.7 us with 1024; profiles/r02_d_win_ab.txt): the half-size tile is the default at every size,
            // the 2048-entry tile remains for FDJAC_WIN_TILE=2048.  What matters is the tile's BYTES: Float32 keeps the
            // 2048-entry tile (N = 10^7: 52.3 vs 64.3 us with 1024 entries; Float64: 100.1 vs 98.0 us)
            const bool prefer_small = sizeof(real_t) >= 8;
            for (int T : {2048, 1024, 512}) {
                if (force_t && T != force_t) continue;
                if (!force_t && T == 2048 && prefer_small) continue;
                WinBuild w = build_windows(T, false);
                if (!w.ok) continue;
                const size_t lds = window_lds_bytes(p->fdtype, w.max_slots, w.max_ncol);
                if (lds > (size_t)kWinMaxLds) continue;
                const bool cheap = w.overread <= 1.25 || (scattered && w.overread <= kWinMaxOverread);
                if (!(cheap || force_w == 1)) continue;
                best = std::move(w);
                if (lds <= (size_t)32 * 1024 || T == 1024) break;   // the large tile already leaves >= 5 workgroups per CU
            }
        }
        p->win_overread = best.overread;
        if (best.ok) {
            best = build_windows(best.T, true);
            // Regular patterns (a band coloured cyclically: tridiagonal CSC, BandedMatrix) repeat their entry codes:
            // code[q + P] == code[q] + S inside a tile (the slot field advances by S rows, the colour comes back).  Tiles
            // where that holds throughout are flagged; the kernel reads only their first kWinPeriodMax codes and
            // computes the rest -- 2 B of index traffic per stored entry less (tridiagonal: 60 of 620 MB).
            {
                const char *fp = fdjac::test_switch("FDJAC_WIN_PERIODIC");
                const size_t T = (size_t)best.T, ntiles = padded / T;
                int P = 0, S = 0;
                if (!(fp && *fp && atoi(fp) == 0) && ntiles >= 3) {
                    // (three sample tiles: one of them may hold a column without colour)
                    for (size_t sample : {ntiles / 2, ntiles / 4, (3 * ntiles) / 4}) {
                        const uint16_t *c = &best.code[sample * T];
                        for (int cand = 1; cand <= kWinPeriodMax && !P; ++cand) {
                            const int s0 = (int)c[cand] - (int)c[0];
                            bool okp = true;
                            for (size_t q = 0; q < T && okp; ++q)
                                okp = c[q] < 0x4000 && (q + cand >= T || (int)c[q + cand] - (int)c[q] == s0);
                            if (okp) { P = cand; S = s0; }
                        }
                        if (P) break;
                    }
                }
                int magic = 0;
                if (P) {
                    magic = (int)(((1u << 20) + (unsigned)P - 1) / (unsigned)P);
                    for (size_t q = 0; q < T; ++q)
                        if ((int)(((int64_t)q * magic) >> 20) != (int)(q / (size_t)P)) { P = 0; break; }
                }
                size_t regular = 0;
                if (P) {
                    for (size_t t = 0; t < ntiles; ++t) {
                        const uint16_t *c = &best.code[t * T];
                        bool okt = true;
                        for (size_t q = 0; q < T && okt; ++q)
                            okt = c[q] < 0x4000 && (q + (size_t)P >= T || (int)c[q + P] - (int)c[q] == S);
                        if (okt) { best.wt[3 * t].w |= 0x100; ++regular; }
                    }
                    if (2 * regular < ntiles) {   // not worth the second code path
                        for (size_t t = 0; t < ntiles; ++t) best.wt[3 * t].w &= ~0x100;
                        P = 0;
                    }
                }
                p->win_per_P = P;
                p->win_per_S = P ? S : 0;
                p->win_per_magic = P ? magic : 0;
            }
            p->window = true;
            p->win_tile = best.T;
            p->win_pairs = best.max_slots / 2;
            p->win_ncol = best.max_ncol;
            if ((rc = dev_upload(&p->d_wtiles, best.wt))) return rc;
            if ((rc = dev_upload(&p->d_wcode, best.code))) return rc;
        }
    }
    return FD_OK;
}

// Shape of the 2-D tiles (L positions x R column runs) -- shared by the host builder below and the device builder
// (fdjac_planbuild.hip).  false: no usable shape.
static bool w2_shape(const fd_plan *p, int ecmax, int halo, int *L_out, int *R_out)
{
    // 62 positions (a window row of L + 2*halo (+ alignment) values = 33 row pairs) and as many runs as keep the window pairs of a
    // tile within ONE load round of the 256 threads -- 5-point central at N = 10^7, same process: 62 x 5 299 us, 62 x 6 303,
    // 62 x 4 302, 64 x 6 310, 64 x 5 304, 94 x 3 304, 126 x 2 313
    const int L = 62;
    int R = (int)(2048 / ((int64_t)ecmax * L));
    R = std::max(1, std::min(R, kW2MaxRun));
    // keep the LDS tile (R+2 windows of L+2*halo rows, every staged array) near 32 KB
    const int ncol_guess = std::min<int>((int)std::max<int64_t>(p->C, 1), kWinMaxCol);
    while (R > 2 && window_lds_bytes(p->fdtype, (R + 2) * (L + 2 * halo + 2), ncol_guess) > (size_t)36 * 1024) --R;
    while (R > 2 && (R + 2) * ((L + 2 * halo + 2) / 2) > kBlock) --R;   // one load round
    *L_out = L; *R_out = R;
    return R >= 2;
}


// 2-D (strided) tiles for the row-window kernel (k_decompress_window2d): 2-D stencil patterns in natural ordering.
// Detection: apart from a few near-diagonal offsets (|row - col| <= 8) every entry sits one "stride" s away from the
// diagonal (within +-4), the same s for (almost) the whole pattern, s >= 64.  Tiles are then R consecutive grid rows
// (column runs s apart) x L positions; the row windows each tile needs are found from its entries as for the 1-D
// tiles.  colstart[j - col0] = local index of the first entry of column j (size ncols + 1).
static int try_window2d_plan(fd_plan *p, const std::vector<int32_t> &rows, const std::vector<int32_t> &nzc,
                             const std::vector<int64_t> &colstart)
{
    int rc;
    const char *fw = fdjac::test_switch("FDJAC_WINDOW2D");
    if (fw && *fw && atoi(fw) == 0) return FD_OK;
    const int64_t ncols = (int64_t)colstart.size() - 1;
    if (ncols < 1024 || p->nnz_local < 8192) return FD_OK;
    // --- the stride: most common far offset over a sample of columns
    int64_t s = 0;
    int ecmax = 0, halo = 0;
    {
        std::vector<int64_t> fars;
        const int64_t step = std::max<int64_t>(1, ncols / 4096);
        for (int64_t jj = 0; jj < ncols; jj += step) {
            const int64_t j = p->col0 + jj;
            for (int64_t e = colstart[(size_t)jj]; e < colstart[(size_t)jj + 1]; ++e) {
                const int64_t o = (int64_t)rows[(size_t)e] - j;
                if (o > 8 || o < -8) fars.push_back(o < 0 ? -o : o);
            }
        }
        if (fars.empty()) return FD_OK;
        std::sort(fars.begin(), fars.end());
        s = fars[fars.size() / 2];
        if (s < 64 || ncols < 4 * s) return FD_OK;
        int64_t bad = 0, total = 0;
        for (int64_t jj = 0; jj < ncols; ++jj) {
            const int64_t j = p->col0 + jj;
            const int cnt = (int)(colstart[(size_t)jj + 1] - colstart[(size_t)jj]);
            ecmax = std::max(ecmax, cnt);
            for (int64_t e = colstart[(size_t)jj]; e < colstart[(size_t)jj + 1]; ++e, ++total) {
                int64_t o = (int64_t)rows[(size_t)e] - j;
                if (o < 0) o = -o;
                if (o <= 8) { halo = std::max<int>(halo, (int)o); continue; }
                if (o < s - 4 || o > s + 4) ++bad;
                else halo = std::max<int>(halo, (int)(o > s ? o - s : s - o));
            }
        }
        if (bad * 1000 > total || ecmax < 1 || ecmax > 32) return FD_OK;   // > 0.1 % of the entries off-stride
    }
    int L, R;
    if (!w2_shape(p, ecmax, halo, &L, &R)) return FD_OK;

    const int64_t g_lo = p->col0 / s, g_hi = (p->col1 - 1) / s;          // grid rows touched by the local columns
    const int64_t nG = (g_hi - g_lo + R) / R, nI = (s + L - 1) / L;
    std::vector<int> desc;
    std::vector<uint16_t> code;
    desc.reserve((size_t)(nG * nI) * kW2Desc);
    code.reserve((size_t)p->nnz_local + (size_t)(nG * nI) * 2 * R);
    std::vector<int32_t> rr;
    int max_slots = 0, max_ncol = 0;
    double elems = 0;
    int64_t ntiles = 0, covered = 0;
    for (int64_t G = 0; G < nG; ++G)
        for (int64_t I = 0; I < nI; ++I) {
            int d[kW2Desc] = {0};
            int nruns = 0;
            int64_t run_a[kW2MaxRun], run_b[kW2MaxRun];   // local entry ranges
            for (int q = 0; q < R; ++q) {
                const int64_t g = g_lo + G * R + q;
                if (g > g_hi) break;
                int64_t c0 = g * s + I * L, c1 = g * s + std::min<int64_t>((I + 1) * L, s);
                c0 = std::max<int64_t>(c0, p->col0);
                c1 = std::min<int64_t>(c1, p->col1);
                if (c1 <= c0) continue;
                const int64_t a = colstart[(size_t)(c0 - p->col0)], b = colstart[(size_t)(c1 - p->col0)];
                if (b <= a) continue;
                run_a[nruns] = a; run_b[nruns] = b; ++nruns;
            }
            if (nruns == 0) continue;
            // rows / colours of the tile
            rr.clear();
            int32_t cmin = std::numeric_limits<int32_t>::max(), cmax = -1;
            for (int r = 0; r < nruns; ++r)
                for (int64_t e = run_a[r]; e < run_b[r]; ++e) {
                    if (nzc[(size_t)e] < 0) continue;
                    rr.push_back(rows[(size_t)e]);
                    cmin = std::min(cmin, nzc[(size_t)e]); cmax = std::max(cmax, nzc[(size_t)e]);
                }
            int32_t wr[kW2MaxWin], wn[kW2MaxWin], ends[kW2MaxWin];
            int nwin = 0;
            if (!rr.empty()) {
                if (cmax - cmin + 1 > kWinMaxCol) return FD_OK;
                std::sort(rr.begin(), rr.end());
                int32_t start = rr[0] & ~1, last = rr[0];
                for (size_t k = 1; k <= rr.size(); ++k) {
                    if (k == rr.size() || rr[k] - last > 16) {     // rows of one grid row are contiguous; next one is a stride away
                        if (nwin == kW2MaxWin) return FD_OK;
                        wr[nwin] = start; wn[nwin] = (last - start) / 2 + 1; ++nwin;
                        if (k < rr.size()) start = rr[k] & ~1;
                    }
                    if (k < rr.size()) last = rr[k];
                }
            }
            int pairs = 0;
            for (int k = 0; k < nwin; ++k) { pairs += wn[k]; ends[k] = pairs; }
            if (2 * pairs > 2048) return FD_OK;
            const int ncol = rr.empty() ? 0 : cmax - cmin + 1;
            d[0] = rr.empty() ? 0 : cmin; d[1] = ncol; d[2] = pairs; d[3] = nwin; d[4] = nruns;
            const int64_t code0 = (int64_t)code.size();
            d[6] = (int)(uint32_t)(code0 & 0xFFFFFFFFll); d[7] = (int)(code0 >> 32);
            for (int k = 0; k < nwin; ++k) { d[8 + 2 * k] = wr[k]; d[9 + 2 * k] = ends[k]; }
            int nent = 0;
            for (int r = 0; r < nruns; ++r) {
                for (int64_t e = run_a[r]; e < run_b[r]; ++e) {
                    uint16_t c = 0x8000;
                    if (nzc[(size_t)e] == -1) c = 0x4000;
                    else if (nzc[(size_t)e] >= 0) {
                        int k = 0;
                        while (!(rows[(size_t)e] >= wr[k] && rows[(size_t)e] < wr[k] + 2 * wn[k])) ++k;
                        const int slot = 2 * (k ? ends[k - 1] : 0) + (rows[(size_t)e] - wr[k]);
                        c = (uint16_t)(slot | ((nzc[(size_t)e] - cmin) << 11));
                    }
                    code.push_back(c);
                }
                nent += (int)(run_b[r] - run_a[r]);
                if (nent & 1) { code.push_back(0x8000); ++nent; }   // runs start on even code slots: pairs never straddle
                d[32 + 3 * r] = (int)(uint32_t)(run_a[r] & 0xFFFFFFFFll); d[33 + 3 * r] = (int)(run_a[r] >> 32);
                d[34 + 3 * r] = nent;
                covered += run_b[r] - run_a[r];
            }
            if (nent > 2048 + 2 * kW2MaxRun) return FD_OK;
            d[5] = nent;
            desc.insert(desc.end(), d, d + kW2Desc);
            max_slots = std::max(max_slots, 2 * pairs);
            max_ncol = std::max(max_ncol, ncol);
            elems += 2.0 * pairs * ncol;
            ++ntiles;
        }
    if (covered != p->nnz_local) return FD_OK;   // every stored entry must belong to exactly one run
    const double overread = elems / (double)std::max<int64_t>(p->nnz_local, 1);
    const size_t lds = window_lds_bytes(p->fdtype, max_slots, max_ncol);
    if (max_slots == 0 || lds > (size_t)kWinMaxLds || overread > 2.2) return FD_OK;
    code.push_back(0x8000); code.push_back(0x8000);   // the last pair load may touch one code past the end
    p->window = true;
    p->window2d = true;
    p->w2_ntiles = ntiles;
    p->w2_codes = (int64_t)code.size();
    p->win_tile = 0;
    p->win_pairs = max_slots / 2;
    p->win_ncol = max_ncol;
    p->win_overread = overread;
    if ((rc = dev_upload(&p->d_w2desc, desc))) return rc;
    if ((rc = dev_upload(&p->d_wcode, code))) return rc;
    return FD_OK;
}

// sort key of an entry: colour first (uncoloured, then padding, last), row second, storage position third
// Host loops over independent tiles, on up to 32 host threads (a plan for 5.6e7 entries sorts 27 000 tiles: 2.4 s on one core).
template <class F> static int parallel_tiles(size_t ntiles, F body)      // body(first_tile, last_tile); FD_OK or FD_ERR_NOMEM
{
    unsigned hw = std::thread::hardware_concurrency();
    const char *pt = getenv("FDJAC_PLAN_THREADS");
    if (pt && *pt) hw = (unsigned)std::max(1, atoi(pt));
    const size_t nthr = std::min<size_t>({(size_t)std::max(1u, hw), (size_t)32, (ntiles + 63) / 64});
    // an exception inside a worker (std::bad_alloc of a tile's vectors) must not reach std::terminate, nor cross the C ABI
    std::atomic<bool> failed{false};
    auto guarded = [&](size_t a, size_t b) {
        try { body(a, b); } catch (...) { failed.store(true); }
    };
    if (nthr <= 1) {
        guarded((size_t)0, ntiles);
    } else {
        std::vector<std::thread> th;
        const size_t per = (ntiles + nthr - 1) / nthr;
        size_t done_to = 0;                  // tiles [0, done_to) have a thread; the rest run here if a thread cannot be started
        try {
            for (size_t k = 0; k < nthr; ++k) {
                const size_t a = k * per, b = std::min(ntiles, a + per);
                if (a < b) th.emplace_back([=, &guarded] { guarded(a, b); });
                done_to = b;
            }
        } catch (...) {
        }
        if (done_to < ntiles) guarded(done_to, ntiles);
        for (auto &t : th) t.join();
    }
    if (failed.load()) {
        set_error("out of host memory while compiling the plan's tiles");
        return FD_ERR_NOMEM;
    }
    return FD_OK;
}

static void sort_tile_entries(const int32_t *rows, const int32_t *nzc, std::vector<std::pair<uint64_t, int32_t>> &ord)
{
    for (int k = 0; k < kSortTile; ++k) {
        const uint64_t c = nzc[k] >= 0 ? (uint64_t)nzc[k] : (nzc[k] == -1 ? 0xFFFFFFFEull : 0xFFFFFFFFull);
        ord[(size_t)k] = {(c << 32) | (uint64_t)(uint32_t)rows[k], k};
    }
    std::sort(ord.begin(), ord.end());
}

// Gather coherence of the storage order vs a (colour,row)-sorted order, estimated on a sample of tiles (every step-th
// of ntiles tiles of kSortTile entries; rows_of(t) / nzc_of(t) point at tile t's entries): distinct 128-B lines touched
// by one wave-level gather (64 lanes, the kernels' lane->entry maps).  Shared by the host and the device builder.
template <class RowsOf, class NzcOf>
static void gather_coherence(size_t ntiles, size_t step, RowsOf rows_of, NzcOf nzc_of, double *lines_direct, double *lines_sorted)
{
    std::vector<std::pair<uint64_t, int32_t>> ord(kSortTile);
    auto line_key = [&](int32_t c, int32_t r) { return ((int64_t)c << 40) | (int64_t)(r >> 4); };
    double ld = 0, ls = 0;
    size_t ninstr = 0;
    std::vector<int64_t> keys;
    for (size_t t = 0; t < ntiles; t += step) {
        const int32_t *rw = rows_of(t), *nz = nzc_of(t);
        sort_tile_entries(rw, nz, ord);
        for (int g = 0; g < kSortTile / 128; ++g)
            for (int half = 0; half < 2; ++half) {
                keys.clear();
                for (int l = 0; l < 64; ++l) { const size_t e = (size_t)(g * 128 + 2 * l + half); keys.push_back(line_key(nz[e], rw[e])); }
                std::sort(keys.begin(), keys.end());
                ld += (double)(std::unique(keys.begin(), keys.end()) - keys.begin());
                keys.clear();
                for (int l = 0; l < 64; ++l) { const size_t e = (size_t)ord[(size_t)(g * 128 + 64 * half + l)].second; keys.push_back(line_key(nz[e], rw[e])); }
                std::sort(keys.begin(), keys.end());
                ls += (double)(std::unique(keys.begin(), keys.end()) - keys.begin());
                ++ninstr;
            }
    }
    *lines_direct = ld / std::max<size_t>(ninstr, 1);
    *lines_sorted = ls / std::max<size_t>(ninstr, 1);
}

// Uniform band with cyclic colours -> k_decompress_band for the whole tiles inside it (shared by the host and the device
// builder).  The columns [ju0, ju1) hold w consecutive rows j - u .. j - u + w - 1 each, the first of them starts at the
// local entry e_ju0; colours are (j + shift) mod C for every column.
// The descriptor the row-window kernel would load for tile t of a uniform band (fdjac_kernels.hip computes the same):
// entries Q0 .. Q1 = w*j + k, rows j - u + k
static inline void band_tile_desc(int64_t t, int T, int64_t nnz_local, int64_t off, int w, int u, int C, int *wr0, int *pairs)
{
    const int64_t q0 = t * T, q1 = std::min<int64_t>(q0 + T, nnz_local) - 1;
    const int64_t Q0 = q0 + off, Q1 = q1 + off;
    const int64_t j0 = Q0 / w, k0 = Q0 - j0 * w, j1 = Q1 / w, k1 = Q1 - j1 * w;
    const int64_t rmin = j1 > j0 ? j0 - u + std::min<int64_t>(k0, 1) : j0 - u + k0;
    const int64_t rmax = j1 > j0 ? j1 - u + std::max<int64_t>(k1, w - 2) : j1 - u + k1;
    (void)C;
    *wr0 = (int)(rmin & ~(int64_t)1);
    *pairs = (int)((rmax - *wr0) / 2 + 1);
}

// Uniform band with cyclic colours (shared by the host and the device builder).  The columns [ju0, ju1) hold w consecutive
// rows j - u .. j - u + w - 1 each, the first of them starts at the local entry e_ju0; colours are (j + shift) mod C for
// every column.  Sets the band parameters and the tile range whose descriptors the row-window kernel computes (wt_host: the
// plan's 1-D tile descriptors if the caller has them on the host).
static void finish_band_plan(fd_plan *p, int64_t w, int64_t u, int64_t e_ju0, int64_t ju0, int64_t ju1, int64_t C, int shift,
                             const int4 *wt_host = nullptr)
{
    p->bd_t0 = p->bd_t1 = 0;
    if (!p->bd_allowed || !p->window || p->window2d || p->win_tile <= 0 || w < 1 || w > 64 || C < 1 || C > 64 || ju1 <= ju0) return;
    const int64_t T = p->win_tile, all_tiles = (p->nnz_local + T - 1) / T;
    const int64_t pu0 = e_ju0, pu1 = e_ju0 + w * (ju1 - ju0);
    const int64_t off = w * ju0 - e_ju0;
    int64_t t0 = (pu0 + T - 1) / T, t1 = pu1 / T;
    if (pu1 >= p->nnz_local) t1 = all_tiles;
    if (t1 <= t0) return;
    if (off + t0 * T < 0 || off + p->nnz_local + 2 >= ((int64_t)1 << 31) || ju1 + C + 64 >= ((int64_t)1 << 31)) return;
    if (u < -((int64_t)1 << 30) || u > ((int64_t)1 << 30)) return;
    p->band_off = off; p->band_C = C;
    p->band_w = (int)w; p->band_u = (int)u; p->band_shift = shift;
    p->band_mw = fd_magic31((uint32_t)w);
    p->band_mc = fd_magic31((uint32_t)C);
    {   // (self-check of the two dividers on the values that matter most: the ends of the range and multiples of the divisor)
        const uint32_t top = (uint32_t)(off + p->nnz_local + 1);
        for (uint32_t n : {0u, 1u, (uint32_t)w - 1, (uint32_t)w, top - 1, top, top / 2, 0x7FFFFFFFu, (uint32_t)((top / (uint32_t)w) * (uint32_t)w), (uint32_t)((top / (uint32_t)w) * (uint32_t)w) - 1u})
            if (fd_div31(n, p->band_mw) != n / (uint32_t)w || fd_div31(n, p->band_mc) != n / (uint32_t)C) return;
    }
    // computed descriptors: the largest run of tiles around the middle of [t0, t1) whose STORED descriptor is what
    // band_tile_desc computes (regular tiles: periodic codes, every colour of the band, one row window)
    if (p->bd_allowed && p->win_per_P > 0 && T >= 2 * w && (T % 2) == 0) {
        std::vector<int4> tmp;
        const int4 *wt = wt_host;
        if (!wt) {
            tmp.resize((size_t)(3 * all_tiles));
            if (hipMemcpy(tmp.data(), p->d_wtiles, sizeof(int4) * tmp.size(), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
            wt = tmp.data();
        }
        auto matches = [&](int64_t t) {
            int wr0, pairs;
            band_tile_desc(t, (int)T, p->nnz_local, off, (int)w, (int)u, (int)C, &wr0, &pairs);
            const int4 a = wt[3 * t], b = wt[3 * t + 1], c = wt[3 * t + 2];
            return a.x == 0 && a.y == (int)C && a.z == pairs && a.w == (1 | 0x100) && b.x == wr0 && b.y == pairs && b.z == 0 && b.w == pairs &&
                   c.x == 0 && c.y == pairs && c.z == 0 && c.w == pairs;
        };
        const int64_t tm = (t0 + t1) / 2;
        if (matches(tm)) {
            int64_t a = tm, b = tm + 1;
            while (a > t0 && matches(a - 1)) --a;
            while (b < t1 && matches(b)) ++b;
            p->bd_t0 = a; p->bd_t1 = b;
        }
    }
}

// colorvec == (j + shift) mod C for every column (no column without colour)?
static bool colors_cyclic(const std::vector<int32_t> &col0, int64_t C, int *shift_out)
{
    if (col0.empty() || C < 1 || col0[0] < 0) return false;
    const int64_t sh = col0[0];
    for (size_t j = 0; j < col0.size(); ++j)
        if (col0[j] != (int32_t)(((int64_t)j + sh) % C)) return false;
    *shift_out = (int)sh;
    return true;
}

// host detection for a common-pattern CSC plan: the largest run of columns around the middle one with the middle
// column's number of consecutive rows and an affine colptr
static void try_band_plan_csc(fd_plan *p, const std::vector<int32_t> &col0, const std::vector<int32_t> &rows, const std::vector<int64_t> &colstart)
{
    if (!p->bd_allowed || !p->window || p->window2d || p->col1 - p->col0 < 4) return;
    int shift = 0;
    if (!colors_cyclic(col0, p->C, &shift)) return;
    const int64_t jm = (p->col0 + p->col1) / 2;
    auto cs = [&](int64_t j) { return colstart[(size_t)(j - p->col0)]; };
    const int64_t w = cs(jm + 1) - cs(jm);
    if (w < 1 || w > 64) return;
    const int64_t u = jm - rows[(size_t)cs(jm)];
    auto viol = [&](int64_t j) {
        if (cs(j + 1) - cs(j) != w || cs(j) != cs(jm) + w * (j - jm)) return true;
        for (int64_t k = 0; k < w; ++k)
            if (rows[(size_t)(cs(j) + k)] != j - u + k) return true;
        return false;
    };
    if (viol(jm)) return;
    int64_t ju0 = jm, ju1 = jm + 1;
    while (ju0 > p->col0 && !viol(ju0 - 1)) --ju0;
    while (ju1 < p->col1 && !viol(ju1)) ++ju1;
    finish_band_plan(p, w, u, cs(ju0), ju0, ju1, p->C, shift);
}

// EXPERIMENTAL store capability (FDJAC_LAZY_STORE=1): is the pattern EXACTLY the band (corners included) that the arithmetic of
// include/fdjac_device.h describes, with cyclic colours and at least as many colours as the band is wide?
static void try_store_plan_csc(fd_plan *p, const std::vector<int32_t> &col0, const std::vector<int32_t> &rows, const std::vector<int64_t> &colstart)
{
    p->store_ok = false;
    if (!p->store_allowed || p->col1 - p->col0 < 4 || p->nnz_local < 1) return;
    int shift = 0;
    if (!colors_cyclic(col0, p->C, &shift)) return;
    const int64_t jm = (p->col0 + p->col1) / 2;
    auto cs = [&](int64_t j) { return colstart[(size_t)(j - p->col0)]; };
    const int64_t w = cs(jm + 1) - cs(jm);
    if (w < 1 || w > 64 || p->C < w) return;
    const int64_t u = jm - rows[(size_t)cs(jm)];
    if (u < 0 || w - 1 - u < 0) return;
    fd_band_store d;
    memset(&d, 0, sizeof d);
    d.M = p->M; d.N = p->N; d.l = (int)(w - 1 - u); d.u = (int)u; d.C = (int)p->C; d.shift = shift;
    bool ok = true;
    for (int64_t j = p->col0; j < p->col1 && ok; ++j) {
        const int64_t first = std::max<int64_t>(j - u, 0), last = std::min<int64_t>(p->M - 1, j + d.l);
        ok = cs(j) + p->entry_begin == fd_band_colptr(&d, j) && cs(j + 1) - cs(j) == last - first + 1 && last >= first;
        for (int64_t k = 0; ok && k < cs(j + 1) - cs(j); ++k) ok = rows[(size_t)(cs(j) + k)] == first + k;
    }
    ok = ok && cs(p->col1) + p->entry_begin == fd_band_colptr(&d, p->col1);
    if (ok) { p->store_ok = true; p->store_l = d.l; p->store_u = d.u; p->store_C = d.C; p->store_shift = shift; }
}

// ... and for the 5-point stencil on an nx x ny grid in natural ordering (fd_stencil5_store): the pattern must be exactly the
// stencil's, colorvec a valid colouring of it (columns that share a row differ in colour)
static void try_store_plan_stencil5(fd_plan *p, const std::vector<int32_t> &col0, const std::vector<int32_t> &rows, const std::vector<int64_t> &colstart)
{
    p->store5_ok = false;
    if (!p->store_allowed || p->store_ok || p->M != p->N || p->col1 - p->col0 < 16 || p->nnz_local < 16 || p->C < 5) return;
    auto cs = [&](int64_t j) { return colstart[(size_t)(j - p->col0)]; };
    // the grid width from a column with all five entries near the middle of the local range (the middle one itself may be the
    // first or the last of its grid row)
    int64_t jm = (p->col0 + p->col1) / 2;
    for (int t = 0; t < 2 && jm + 1 < p->col1 && cs(jm + 1) - cs(jm) != 5; ++t) ++jm;
    if (cs(jm + 1) - cs(jm) != 5) return;
    const int64_t nx = (int64_t)rows[(size_t)cs(jm) + 4] - jm;
    if (nx < 4 || (nx & 1) || p->N % nx != 0 || p->N / nx < 3) return;
    fd_stencil5_store d;
    memset(&d, 0, sizeof d);
    d.nx = nx; d.ny = p->N / nx;
    bool ok = cs(p->col1) + p->entry_begin == fd_stencil5_colptr(&d, p->col1);
    for (int64_t k = p->col0; k < p->col1 && ok; ++k) {
        const int64_t j = k / nx, i = k - j * nx;
        int64_t want[5];
        int n = 0;
        if (j > 0) want[n++] = k - nx;
        if (i > 0) want[n++] = k - 1;
        want[n++] = k;
        if (i < nx - 1) want[n++] = k + 1;
        if (j < d.ny - 1) want[n++] = k + nx;
        ok = cs(k) + p->entry_begin == fd_stencil5_colptr(&d, k) && cs(k + 1) - cs(k) == n;
        for (int q = 0; ok && q < n; ++q) ok = rows[(size_t)(cs(k) + q)] == want[q];
    }
    // valid colouring: the (up to five) columns of every row differ in colour -- every row, whoever owns its columns
    for (int64_t r = 0; r < p->N && ok; ++r) {
        const int64_t j = r / nx, i = r - j * nx;
        int32_t c[5];
        int n = 0;
        c[n++] = col0[(size_t)r];
        if (i > 0) c[n++] = col0[(size_t)(r - 1)];
        if (i < nx - 1) c[n++] = col0[(size_t)(r + 1)];
        if (j > 0) c[n++] = col0[(size_t)(r - nx)];
        if (j < d.ny - 1) c[n++] = col0[(size_t)(r + nx)];
        for (int a = 0; a < n && ok; ++a) {
            if (c[a] < 0) ok = false;
            for (int b = a + 1; b < n; ++b) if (c[a] == c[b]) ok = false;
        }
    }
    if (ok) { p->store5_ok = true; p->store5_nx = nx; p->store5_ny = d.ny; }
}

// the same capability for the storage types whose band is implicit (BandedMatrix data, Tridiagonal): only the colours need
// checking -- cyclic (the step-size reduction's test, host or device builder), at least as many as the band is wide
static void store_caps_implicit_band(fd_plan *p, int64_t l, int64_t u)
{
    p->store_ok = false;
    if (!p->store_allowed || p->cyc_C <= 0 || p->has_none || l < 0 || u < 0 || l + u + 1 > 64 || p->cyc_C < l + u + 1 ||
        p->col1 <= p->col0) return;
    if (p->N - 1 - u > p->M - 1) return;   // (a column without a row inside the matrix: nobody would write its zero slots)
    p->store_ok = true;
    p->store_l = (int)l; p->store_u = (int)u; p->store_C = p->cyc_C; p->store_shift = p->cyc_shift;
}

// Tile ORDER of a sorted-gather plan whose pattern has a far band (3-D stencils: offsets 0, +-1, +-nx, +-nx*ny).  A tile's gathers
// reach the rows a whole "plane" D = max |row - column| away; walking the tiles in storage order the three planes in use are
// 3 * C * D * 8 bytes (6.7 MB for 200^3, 7 colours) against 4 MB of L2 per XCD, and the plane above / below is fetched through the
// fabric a second and third time (rocprofv3: 2.2 x the distinct bytes).  Walk instead: for each in-plane region of Rg columns, all
// planes in turn -- the three-plane working set of a region is 3 * C * Rg * 8 bytes <= 2 MiB.  Pure scheduling: which workgroup
// takes which tile (results do not depend on it).  tcol[t] = local column of tile t's first entry, reach = D.  Shared by the host
// and the device builder.  Empty result: storage order.
static std::vector<int32_t> far_band_tile_order(const fd_plan *p, const std::vector<int64_t> &tcol, int64_t reach, int64_t ncols, size_t ntiles)
{
    std::vector<int32_t> order;
    const char *to = fdjac::test_switch("FDJAC_TILE_ORDER");
    const int want = (to && *to) ? atoi(to) : 1;
    if (want == 0 || ntiles < (want == 2 ? 16u : 256u)) return order;
    const int64_t D = reach;
    const int64_t cpt = std::max<int64_t>(1, ncols / (int64_t)ntiles);
    if (!(D >= (want == 2 ? 2 : 16) * cpt && D < ncols)) return order;
    int64_t Rg = ((int64_t)2 << 20) / (3 * std::max<int64_t>(p->C, 1) * (int64_t)sizeof(real_t));
    Rg = std::max<int64_t>(4 * cpt, std::min<int64_t>(Rg, D / 2));
    // only the tiles the kernel walks: ceil(nnz_local / kSortTile) -- the lists are padded to kListPad (two tiles), and
    // an all-padding tile in the order would displace a real one (its values would never be written)
    const size_t nreal = (size_t)((p->nnz_local + kSortTile - 1) / kSortTile);
    // (region, plane, position in the plane, tile): the tile number last makes the order that of a stable sort
    struct Key { int64_t k, l, u; int32_t t; };
    std::vector<Key> keys(nreal);
    for (size_t t = 0; t < nreal; ++t) {
        const int64_t u = tcol[t] % D;
        keys[t] = Key{u / Rg, tcol[t] / D, u, (int32_t)t};
    }
    std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
        if (a.k != b.k) return a.k < b.k;
        if (a.l != b.l) return a.l < b.l;
        if (a.u != b.u) return a.u < b.u;
        return a.t < b.t;
    });
    order.resize(nreal);
    for (size_t t = 0; t < nreal; ++t) order[t] = keys[t].t;
    return order;
}

// Shared by the three index-list kinds: local entries [e0,e1) with rows, columns (0-based).
static int finish_list_plan(fd_plan *p, const std::vector<int32_t> &col0, std::vector<int32_t> &rows,
                            std::vector<int32_t> &nzc, std::vector<int64_t> &dest,
                            const std::vector<int64_t> *colstart = nullptr)
{
    int rc;
    PbTimer tm(p->ctx->stream);
    p->nnz_local = (int64_t)rows.size();
    int64_t r0 = p->M, r1 = 0;
    for (int32_t r : rows) {
        if (r < r0) r0 = r;
        if (r + 1 > r1) r1 = (int64_t)r + 1;
    }
    if (rows.empty()) r0 = r1 = 0;
    p->row0 = r0;
    p->row1 = r1;
    // pad the lists to whole tiles: row 0, colour "pad" (-2), destination 0 -- never written
    const size_t padded = (size_t)round_up(std::max<int64_t>(p->nnz_local, 1), kListPad);
    const bool has_dest = !dest.empty() || p->kind != K_CSC;
    rows.resize(padded, 0);
    nzc.resize(padded, -2);
    if (has_dest) dest.resize(padded, 0);
    for (int32_t c : col0) if (c < 0) { p->has_none = true; break; }

    bool scattered = false;
    if (!has_dest && p->nnz_local >= 4 * kSortTile) {
        const size_t ntiles = padded / kSortTile;
        const size_t step = std::max<size_t>(1, ntiles / 64);
        gather_coherence(ntiles, step, [&](size_t t) { return rows.data() + t * kSortTile; },
                         [&](size_t t) { return nzc.data() + t * kSortTile; }, &p->lines_direct, &p->lines_sorted);
        scattered = p->lines_direct > 16.0 && p->lines_direct > 1.5 * p->lines_sorted;
    }

    tm.mark("list: rows, padding, coherence");
    if (!has_dest && p->nnz_local > 0) {
        const char *fw1 = fdjac::test_switch("FDJAC_WINDOW"), *fs1 = fdjac::test_switch("FDJAC_SORTED");
        const bool win_allowed = !(fw1 && *fw1 && atoi(fw1) == 0) && !(fs1 && *fs1 && atoi(fs1) == 1);
        // a scattered storage order whose tiles still form ONE tight row window each (2-D stencils on narrow grids) is
        // served by the 1-D tiles -- the order in which the device builder decides, too; then 2-D (strided) tiles; then
        // clustered / sorted windows
        if (scattered && colstart && win_allowed && (rc = try_window_plan(p, rows, nzc, padded, false, true))) return rc;
        if (!p->window && scattered && colstart && win_allowed && (rc = try_window2d_plan(p, rows, nzc, *colstart))) return rc;
        if (!p->window && (rc = try_window_plan(p, rows, nzc, padded, scattered))) return rc;
        if (p->window && colstart && p->kind == K_CSC) try_band_plan_csc(p, col0, rows, *colstart);
        if (colstart && p->kind == K_CSC) try_store_plan_csc(p, col0, rows, *colstart);
        if (colstart && p->kind == K_CSC) try_store_plan_stencil5(p, col0, rows, *colstart);
        if (p->window) {
            // the window kernel needs neither rowval nor the per-entry colours on the device
            rows.clear();
            nzc.clear();
        } else {
            const char *fs = fdjac::test_switch("FDJAC_SORTED");
            p->sorted_gather = scattered;
            if (fs && *fs) p->sorted_gather = atoi(fs) != 0 && p->nnz_local >= 4 * kSortTile;
        }
    }
    tm.mark("list: window / store attempts");
    // (far-band tile order, below: the reach D = max |row - column| and every tile's first column, from the storage order)
    std::vector<int64_t> tcol;
    int64_t reach = 0;
    if (p->sorted_gather && colstart) {
        const size_t ntl = padded / kSortTile;
        tcol.resize(ntl);
        size_t jc = 0;
        for (int64_t e = 0; e < p->nnz_local; ++e) {
            while (jc + 1 < colstart->size() && (*colstart)[jc + 1] <= e) ++jc;
            if ((e % kSortTile) == 0) tcol[(size_t)(e / kSortTile)] = (int64_t)jc;
            if (nzc[(size_t)e] >= 0) reach = std::max<int64_t>(reach, std::llabs((int64_t)rows[(size_t)e] - (p->col0 + (int64_t)jc)));
        }
        for (size_t t = (size_t)((p->nnz_local + kSortTile - 1) / kSortTile); t < ntl; ++t) tcol[t] = (int64_t)colstart->size() - 2;
    }
    tm.mark("list: reach, tile columns");
    if (p->sorted_gather) {
        std::vector<uint16_t> spos(padded);
        const size_t ntiles = padded / kSortTile;
        if ((rc = parallel_tiles(ntiles, [&](size_t ta, size_t tb) {
            std::vector<std::pair<uint64_t, int32_t>> ordl(kSortTile);
            std::vector<int32_t> r2(kSortTile), c2(kSortTile);
            for (size_t t = ta; t < tb; ++t) {
                const size_t b0 = t * kSortTile;
                sort_tile_entries(rows.data() + b0, nzc.data() + b0, ordl);
                for (int q = 0; q < kSortTile; ++q) {
                    const int k = ordl[(size_t)q].second;
                    r2[(size_t)q] = rows[b0 + (size_t)k];
                    c2[(size_t)q] = nzc[b0 + (size_t)k];
                    spos[b0 + (size_t)q] = (uint16_t)k;
                }
                std::copy(r2.begin(), r2.end(), rows.begin() + (ptrdiff_t)b0);
                std::copy(c2.begin(), c2.end(), nzc.begin() + (ptrdiff_t)b0);
            }
        }))) return rc;
        tm.mark("list: tile sorts");
        if ((rc = dev_upload(&p->d_spos, spos))) return rc;
        tm.mark("list: upload positions");
        // f(x) through LDS (forward differences, k_decompress_sorted FXL): the runs of rows every tile touches
        {
            if (p->fdtype == FD_FORWARD) {
                std::vector<int32_t> fxw(ntiles * 2 * kFxWin, 0);
                std::atomic<size_t> eligible{0};
                if ((rc = parallel_tiles(ntiles, [&](size_t ta, size_t tb) {
                std::vector<int32_t> tr;
                std::vector<std::pair<int32_t, int32_t>> runs;
                for (size_t t = ta; t < tb; ++t) {
                    int32_t *w = fxw.data() + t * 2 * kFxWin;
                    tr.clear();
                    for (size_t q = t * kSortTile; q < (t + 1) * kSortTile; ++q)
                        if (nzc[q] >= 0) tr.push_back(rows[q]);
                    w[0] = -1;
                    if (tr.empty()) continue;
                    std::sort(tr.begin(), tr.end());
                    runs.clear();
                    runs.push_back({tr[0], tr[0]});
                    for (int32_t r : tr) {
                        if (r <= runs.back().second + 16) runs.back().second = std::max(runs.back().second, r);   // (gaps of <= 15 rows stay inside a run)
                        else runs.push_back({r, r});
                    }
                    while (runs.size() > (size_t)kFxWin) {        // too many runs: close the smallest gap
                        size_t best = 1;
                        for (size_t i = 2; i < runs.size(); ++i)
                            if (runs[i].first - runs[i - 1].second < runs[best].first - runs[best - 1].second) best = i;
                        runs[best - 1].second = runs[best].second;
                        runs.erase(runs.begin() + (ptrdiff_t)best);
                    }
                    int64_t total = 0;
                    for (auto &ru : runs) total += (int64_t)ru.second - ru.first + 1;
                    if (total > kFxRows) continue;
                    for (size_t i = 0; i < runs.size(); ++i) { w[2 * i] = runs[i].first; w[2 * i + 1] = runs[i].second - runs[i].first + 1; }
                    ++eligible;
                }
                }))) return rc;
                if (eligible.load() * 2 >= ntiles && (rc = dev_upload(&p->d_fxwin, fxw))) return rc;
            }
        }
        if (colstart) {
            const std::vector<int32_t> order = far_band_tile_order(p, tcol, reach, (int64_t)colstart->size() - 1, ntiles);
            if (!order.empty() && (rc = dev_upload(&p->d_tile_order, order))) return rc;
        }
    }
    tm.mark("list: tile order");
    if ((rc = dev_upload(&p->d_rowval, rows))) return rc;
    if ((rc = upload_colors(p, col0, nzc))) return rc;
    if (has_dest && (rc = dev_upload(&p->d_dest, dest))) return rc;
    tm.mark("list: uploads (rows, colours)");
    rc = alloc_scratch(p, col0);
    tm.mark("list: scratch");
    return rc;
}

