#include <algorithm>
#include <atomic>
// The compact device copy of a plan's local CSC pattern (FD_PLAN_STORE_CSC; fd_csc_store in include/fdjac_device.h): int32 column
// offsets relative to the local column range and int32 0-based rows, converted ON THE DEVICE from the caller's colptr / rowval (any
// index width / base).  A column-centric f! kernel walks it to store the Jacobian of ANY pattern itself (the reference's
// decompression loop, ext/FiniteDiffSparseArraysExt.jl:38-47, run by the kernel that evaluates f!).
// Included by fdjac_api.hip (namespace fdjac).

namespace fdjac {

template <typename IT>
__global__ void __launch_bounds__(kBlock) k_csc_compact_colptr(const IT *__restrict__ colptr, int64_t base, int64_t col0, int64_t ncols1, int64_t e0,
                                                               int *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k < ncols1) out[k] = (int)((int64_t)colptr[col0 + k] - base - e0);
}
template <typename IT>
__global__ void __launch_bounds__(kBlock) k_csc_compact_rowval(const IT *__restrict__ rowval, int64_t base, int64_t e0, int64_t n, int *__restrict__ out)
{
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (q < n) out[q] = (int)((int64_t)rowval[e0 + q] - base);
}

// max |row - column| over the local entries (fd_csc_store.reach: how far a workgroup's columns reach into the rows, hence how much of
// x a staging kernel keeps in LDS)
__global__ void __launch_bounds__(kBlock) k_csc_reach(const int *__restrict__ colptr, const int *__restrict__ rowval, int64_t col0, int64_t ncols,
                                                      int *__restrict__ reach)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int m = 0;
    if (k < ncols) {
        const int a = colptr[k], b = colptr[k + 1];
        if (b > a) {       // (rows ascend within a column: the first and the last entry bound the column's reach)
            const int64_t d0 = (int64_t)rowval[a] - (col0 + k), d1 = (int64_t)rowval[b - 1] - (col0 + k);
            const int64_t m0 = d0 < 0 ? -d0 : d0, m1 = d1 < 0 ? -d1 : d1;
            m = (int)(m0 > m1 ? m0 : m1);
            for (int q = a + 1; q + 1 < b; ++q) { const int64_t d = (int64_t)rowval[q] - (col0 + k); const int md = (int)(d < 0 ? -d : d); m = md > m ? md : m; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(reach, m);
}

// Is colorvec a VALID colouring of the local pattern -- do the columns that share a row differ in colour?  One thread per local
// column sets its colour's bit in the mask of every row it touches; a bit that was set already is a conflict.  (Columns without a
// colour conflict with nothing.)  Knowing it lets a storing kernel perturb ONE coordinate instead of testing colours.
template <typename CT>
__global__ void __launch_bounds__(kBlock) k_csc_valid_coloring(const int *__restrict__ colptr, const int *__restrict__ rowval, int64_t col0, int64_t ncols,
                                                               const CT *__restrict__ color, int64_t row0, int words,
                                                               unsigned long long *__restrict__ mask, int *__restrict__ conflict)
{
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= ncols) return;
    const int c = (int)color[col0 + k];
    if (c == (int)(CT)(-1)) return;
    const unsigned long long bit = 1ull << (c & 63);
    bool bad = false;
    for (int q = colptr[k]; q < colptr[k + 1]; ++q) {
        const unsigned long long old = atomicOr(mask + ((int64_t)rowval[q] - row0) * words + (c >> 6), bit);
        bad = bad || (old & bit) != 0;
    }
    if (bad) atomicOr(conflict, 1);
}

// The same question for a plan with a COLUMN WINDOW: the colour's point perturbs every column of the colour, also those outside the
// window, so a column outside that shares a LOCAL row with a local column of its colour makes the colouring invalid for this plan
// too (found by the randomised sweep: a locally valid, globally invalid colouring stored single-coordinate differences where the
// reference's columns collide).  All N columns of the caller's raw arrays, rows of the local row range only.
template <typename IT, typename CT>
__global__ void __launch_bounds__(kBlock) k_csc_valid_coloring_raw(const IT *__restrict__ colptr, const IT *__restrict__ rowval, int64_t base, int64_t N,
                                                                   const CT *__restrict__ color, int64_t row0, int64_t row1, int words,
                                                                   unsigned long long *__restrict__ mask, int *__restrict__ conflict)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= N) return;
    const int c = (int)color[j];
    if (c == (int)(CT)(-1)) return;
    const unsigned long long bit = 1ull << (c & 63);
    bool bad = false;
    for (int64_t q = (int64_t)colptr[j] - base; q < (int64_t)colptr[j + 1] - base; ++q) {
        const int64_t r = (int64_t)rowval[q] - base;
        if (r < row0 || r >= row1) continue;
        const unsigned long long old = atomicOr(mask + (r - row0) * words + (c >> 6), bit);
        bad = bad || (old & bit) != 0;
    }
    if (bad) atomicOr(conflict, 1);
}

// colptr_dev / rowval_dev: device pointers addressed with ABSOLUTE column / entry indices (as device_build_csc takes them).
// full: ALL columns / entries of the pattern are behind them (not only the local slice).
template <typename IT>
static int build_store_csc_t(fd_plan *p, const IT *colptr_dev, const IT *rowval_dev, int idx_base, bool full)
{
    hipStream_t s = p->ctx->stream;
    const int64_t n = p->nnz_local, ncols = p->col1 - p->col0;
    FD_HIP_CHECK(hipMalloc((void **)&p->d_sc_colptr, sizeof(int) * (size_t)(ncols + 1)));
    FD_HIP_CHECK(hipMalloc((void **)&p->d_sc_rowval, sizeof(int) * (size_t)(n + 8)));      // (+ a pad: launchers may read one entry past an empty last column)
    FD_HIP_CHECK(hipMemsetAsync(p->d_sc_rowval + n, 0, sizeof(int) * 8, p->ctx->stream));
    FD_HIP_CHECK(hipMalloc((void **)&p->d_sc_note, 4 * sizeof(unsigned long long)));
    FD_HIP_CHECK(hipMemsetAsync(p->d_sc_note, 0, 4 * sizeof(unsigned long long), p->ctx->stream));
    {
        static std::atomic<unsigned long long> serial{0};
        p->sc_serial = ++serial;
    }
    hipLaunchKernelGGL((k_csc_compact_colptr<IT>), dim3((unsigned)((ncols + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, colptr_dev, (int64_t)idx_base,
                       p->col0, ncols + 1, p->entry_begin, p->d_sc_colptr);
    if (n > 0)
        hipLaunchKernelGGL((k_csc_compact_rowval<IT>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, rowval_dev, (int64_t)idx_base,
                           p->entry_begin, n, p->d_sc_rowval);
    FD_HIP_CHECK(hipGetLastError());
    // valid colouring?  (row masks: ceil(C / 64) words per local row; skipped -- "not verified" -- beyond 1 GiB of masks)
    p->sc_valid = false;
    const int64_t R = p->row1 - p->row0;
    const int words = (int)((std::max<int64_t>(p->C, 1) + 63) / 64);
    const bool windowed = !(p->col0 == 0 && p->col1 == p->N);
    if (n > 0 && R > 0 && R * words * 8 <= ((int64_t)1 << 30) && (!windowed || full)) {
        unsigned long long *d_mask = nullptr;
        int *d_conflict = nullptr;
        if (hipMalloc((void **)&d_mask, (size_t)(R * words) * 8) == hipSuccess && hipMalloc((void **)&d_conflict, sizeof(int)) == hipSuccess) {
            (void)hipMemsetAsync(d_mask, 0, (size_t)(R * words) * 8, s);
            (void)hipMemsetAsync(d_conflict, 0, sizeof(int), s);
            const unsigned g = (unsigned)((ncols + kBlock - 1) / kBlock), gN = (unsigned)((p->N + kBlock - 1) / kBlock);
            if (windowed && p->color8)
                hipLaunchKernelGGL((k_csc_valid_coloring_raw<IT, uint8_t>), dim3(gN), dim3(kBlock), 0, s, colptr_dev, rowval_dev, (int64_t)idx_base, p->N,
                                   (const uint8_t *)p->d_color, p->row0, p->row1, words, d_mask, d_conflict);
            else if (windowed)
                hipLaunchKernelGGL((k_csc_valid_coloring_raw<IT, int32_t>), dim3(gN), dim3(kBlock), 0, s, colptr_dev, rowval_dev, (int64_t)idx_base, p->N,
                                   (const int32_t *)p->d_color, p->row0, p->row1, words, d_mask, d_conflict);
            else if (p->color8)
                hipLaunchKernelGGL((k_csc_valid_coloring<uint8_t>), dim3(g), dim3(kBlock), 0, s, p->d_sc_colptr, p->d_sc_rowval, p->col0, ncols,
                                   (const uint8_t *)p->d_color, p->row0, words, d_mask, d_conflict);
            else
                hipLaunchKernelGGL((k_csc_valid_coloring<int32_t>), dim3(g), dim3(kBlock), 0, s, p->d_sc_colptr, p->d_sc_rowval, p->col0, ncols,
                                   (const int32_t *)p->d_color, p->row0, words, d_mask, d_conflict);
            int conflict = 1;
            if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&conflict, d_conflict, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess &&
                hipStreamSynchronize(s) == hipSuccess)
                p->sc_valid = conflict == 0;
        }
        if (d_mask) (void)hipFree(d_mask);
        if (d_conflict) (void)hipFree(d_conflict);
        (void)hipGetLastError();
    }
    p->sc_reach = -1;
    if (n > 0) {
        int *d_reach = nullptr;
        if (hipMalloc((void **)&d_reach, sizeof(int)) == hipSuccess) {
            int r = 0;
            (void)hipMemsetAsync(d_reach, 0, sizeof(int), s);
            hipLaunchKernelGGL(k_csc_reach, dim3((unsigned)((ncols + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, p->d_sc_colptr, p->d_sc_rowval, p->col0, ncols, d_reach);
            if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&r, d_reach, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess &&
                hipStreamSynchronize(s) == hipSuccess)
                p->sc_reach = r;
            (void)hipFree(d_reach);
        }
        (void)hipGetLastError();
    }
    FD_HIP_CHECK(hipStreamSynchronize(s));
    // the pattern by rows (FD_PLAN_STORE_CSC_ROWS): a counting sort of the compact copy on the host -- walking the columns in order
    // leaves every row's entries in ascending column order; plan time, once
    if (p->want_store_rows && !windowed && n > 0 && p->M > 0) {
        std::vector<int> cp((size_t)ncols + 1), rv((size_t)n), rp((size_t)p->M + 1, 0), rc((size_t)n), rs((size_t)n);
        FD_HIP_CHECK(hipMemcpy(cp.data(), p->d_sc_colptr, sizeof(int) * cp.size(), hipMemcpyDeviceToHost));
        FD_HIP_CHECK(hipMemcpy(rv.data(), p->d_sc_rowval, sizeof(int) * rv.size(), hipMemcpyDeviceToHost));
        bool ok = true;
        for (int64_t q = 0; q < n; ++q) {
            if (rv[(size_t)q] < 0 || rv[(size_t)q] >= p->M) { ok = false; break; }
            rp[(size_t)rv[(size_t)q] + 1] += 1;
        }
        if (ok) {
            for (int64_t r = 0; r < p->M; ++r) rp[(size_t)r + 1] += rp[(size_t)r];
            std::vector<int> cur(rp.begin(), rp.end() - 1);
            for (int64_t j = 0; j < ncols; ++j)
                for (int q = cp[(size_t)j]; q < cp[(size_t)j + 1]; ++q) {
                    const int at = cur[(size_t)rv[(size_t)q]]++;
                    rc[(size_t)at] = (int)j;
                    rs[(size_t)at] = q;
                }
            // ... and, tile of 256 rows by tile, the rows in order of DESCENDING length (stable): fd_csc_store_rows gives a wavefront 64
            // rows of similar length -- a thread per row idles while the longest row of its wavefront is worked through
            std::vector<int> ro((size_t)p->M);
            for (int64_t t0 = 0; t0 < p->M; t0 += 256) {
                const int64_t t1 = std::min<int64_t>(t0 + 256, p->M);
                for (int64_t r = t0; r < t1; ++r) ro[(size_t)r] = (int)r;
                std::stable_sort(ro.begin() + t0, ro.begin() + t1, [&](int a, int b) { return rp[(size_t)a + 1] - rp[(size_t)a] > rp[(size_t)b + 1] - rp[(size_t)b]; });
            }
            // what a thread of fd_csc_store_rows needs about its row in ONE load, and what a tile needs about its run of the lists in
            // one load from a table small enough to stay in the L2 (both instead of chains through row_ptr: round trips of the prologue)
            const int64_t ntile = (p->M + 255) / 256;
            std::vector<int> pack(2 * (size_t)p->M), tptr((size_t)ntile + 1);
            for (int64_t t = 0; t <= ntile; ++t) tptr[(size_t)t] = rp[(size_t)std::min<int64_t>(256 * t, p->M)];
            for (int64_t q = 0; q < p->M; ++q) {
                const int r = ro[(size_t)q];
                const int64_t b0 = (int64_t)rp[(size_t)r] - tptr[(size_t)(q / 256)], len = (int64_t)rp[(size_t)r + 1] - rp[(size_t)r];
                pack[2 * (size_t)q] = r;
                pack[2 * (size_t)q + 1] = (int)(std::min<int64_t>(b0, 65535) | (std::min<int64_t>(len, 32767) << 16));
            }
            FD_HIP_CHECK(hipMalloc((void **)&p->d_sr_order, sizeof(int) * pack.size()));
            FD_HIP_CHECK(hipMemcpy(p->d_sr_order, pack.data(), sizeof(int) * pack.size(), hipMemcpyHostToDevice));
            FD_HIP_CHECK(hipMalloc((void **)&p->d_sr_tile, sizeof(int) * tptr.size()));
            FD_HIP_CHECK(hipMemcpy(p->d_sr_tile, tptr.data(), sizeof(int) * tptr.size(), hipMemcpyHostToDevice));
            // ... and the ENTRIES of every tile in that order of the rows (fd_csc_store_ents: a thread per entry; 64 consecutive entries are
            // then entries of rows of one length, so the loop over a row's terms runs the same number of trips in every lane): column,
            // slot, and {the row's place in the tile's order | the entry's place in its row << 8 | the row's length << 16}.  Only when
            // every tile's entries fit the kernel's registers (8 per thread) and every row has at most 255 entries.
            {
                int64_t max_tile = 0;
                int max_len = 0;
                for (int64_t t = 0; t < ntile; ++t) max_tile = std::max<int64_t>(max_tile, (int64_t)tptr[(size_t)t + 1] - tptr[(size_t)t]);
                for (int64_t r = 0; r < p->M; ++r) max_len = std::max(max_len, rp[(size_t)r + 1] - rp[(size_t)r]);
                // (measured equal to the row form on the random band, 98-104 us both: built -- 12 more bytes per entry -- only where asked for)
                const char *sw = test_switch("FDJAC_ROWS_ENTS");
                if (sw && *sw == '1' && max_tile <= 2048 && max_len <= 255) {
                    std::vector<int> ec((size_t)n), es((size_t)n), ei((size_t)n);
                    for (int64_t q = 0, at = 0; q < p->M; ++q) {
                        const int r = ro[(size_t)q], len = rp[(size_t)r + 1] - rp[(size_t)r];
                        for (int k = 0; k < len; ++k, ++at) {
                            ec[(size_t)at] = rc[(size_t)rp[(size_t)r] + k];
                            es[(size_t)at] = rs[(size_t)rp[(size_t)r] + k];
                            ei[(size_t)at] = (int)(q & 255) | (k << 8) | (len << 16);
                        }
                    }
                    p->se_tile_max = (int)max_tile;
                    FD_HIP_CHECK(hipMalloc((void **)&p->d_se_col, sizeof(int) * ec.size()));
                    FD_HIP_CHECK(hipMalloc((void **)&p->d_se_slot, sizeof(int) * es.size()));
                    FD_HIP_CHECK(hipMalloc((void **)&p->d_se_info, sizeof(int) * ei.size()));
                    FD_HIP_CHECK(hipMemcpy(p->d_se_col, ec.data(), sizeof(int) * ec.size(), hipMemcpyHostToDevice));
                    FD_HIP_CHECK(hipMemcpy(p->d_se_slot, es.data(), sizeof(int) * es.size(), hipMemcpyHostToDevice));
                    FD_HIP_CHECK(hipMemcpy(p->d_se_info, ei.data(), sizeof(int) * ei.size(), hipMemcpyHostToDevice));
                }
            }
            FD_HIP_CHECK(hipMalloc((void **)&p->d_sr_ptr, sizeof(int) * rp.size()));
            FD_HIP_CHECK(hipMalloc((void **)&p->d_sr_col, sizeof(int) * rc.size()));
            FD_HIP_CHECK(hipMalloc((void **)&p->d_sr_slot, sizeof(int) * rs.size()));
            FD_HIP_CHECK(hipMemcpy(p->d_sr_ptr, rp.data(), sizeof(int) * rp.size(), hipMemcpyHostToDevice));
            FD_HIP_CHECK(hipMemcpy(p->d_sr_col, rc.data(), sizeof(int) * rc.size(), hipMemcpyHostToDevice));
            FD_HIP_CHECK(hipMemcpy(p->d_sr_slot, rs.data(), sizeof(int) * rs.size(), hipMemcpyHostToDevice));
        }
    }
    p->sc_entries = n;
    p->store_csc_ok = true;
    return FD_OK;
}

static bool store_csc_wanted(const fd_plan *p)
{
    return p->want_store_csc && p->kind == K_CSC && !p->cx && (p->store_csc_always || (!p->store_ok && !p->store5_ok)) && p->store_allowed && p->nnz_local > 0 &&
           p->nnz_local < ((int64_t)1 << 31) && p->M < ((int64_t)1 << 31) && p->d_color != nullptr;
}

static int build_store_csc(fd_plan *p, const void *colptr_dev, const void *rowval_dev, int idx_bytes, int idx_base, bool full)
{
    if (!store_csc_wanted(p)) return FD_OK;
    return idx_bytes == 8 ? build_store_csc_t<int64_t>(p, (const int64_t *)colptr_dev, (const int64_t *)rowval_dev, idx_base, full)
                          : build_store_csc_t<int32_t>(p, (const int32_t *)colptr_dev, (const int32_t *)rowval_dev, idx_base, full);
}

// the same from HOST arrays: the local slices are uploaded first -- all of the pattern for a plan with a column window (its
// colouring is checked against every column, see k_csc_valid_coloring_raw)
static int build_store_csc_host(fd_plan *p, const void *colptr, const void *rowval, int idx_bytes, int idx_base)
{
    if (!store_csc_wanted(p)) return FD_OK;
    const size_t ib = (size_t)idx_bytes;
    const bool windowed = !(p->col0 == 0 && p->col1 == p->N);
    const int64_t c0 = windowed ? 0 : p->col0, c1 = windowed ? p->N : p->col1;
    const int64_t e0 = load_idx(colptr, idx_bytes, c0) - idx_base, e1 = load_idx(colptr, idx_bytes, c1) - idx_base;
    void *d_cp = nullptr, *d_rv = nullptr;
    hipError_t e = hipMalloc(&d_cp, ib * (size_t)(c1 - c0 + 1));
    if (e == hipSuccess) e = hipMalloc(&d_rv, ib * (size_t)std::max<int64_t>(e1 - e0, 1));
    if (e == hipSuccess) e = hipMemcpyAsync(d_cp, (const char *)colptr + ib * (size_t)c0, ib * (size_t)(c1 - c0 + 1), hipMemcpyHostToDevice, p->ctx->stream);
    if (e == hipSuccess && e1 > e0) e = hipMemcpyAsync(d_rv, (const char *)rowval + ib * (size_t)e0, ib * (size_t)(e1 - e0), hipMemcpyHostToDevice, p->ctx->stream);
    int rc = FD_OK;
    if (e != hipSuccess) {
        set_error("uploading the pattern for the storing launch failed: %s", hipGetErrorString(e));
        rc = FD_ERR_HIP;
    } else {
        rc = build_store_csc(p, (const char *)d_cp - ib * (size_t)c0, (const char *)d_rv - ib * (size_t)e0, idx_bytes, idx_base, windowed);
    }
    (void)hipStreamSynchronize(p->ctx->stream);
    if (d_cp) (void)hipFree(d_cp);
    if (d_rv) (void)hipFree(d_rv);
    return rc;
}

}  // namespace fdjac
