// The per-(row, colour) destination table of a general SparseMatrixCSC pattern (FD_PLAN_STORE_TABLE; fd_rowlist_store in
// include/fdjac_device.h), compiled ON THE DEVICE from the caller's colptr / rowval (any index width / base) and the plan's
// colours.  It is the transpose of the local CSC pattern -- the stored entries grouped by ROW, a row's entries by ascending
// column -- with the entry's position in nzval and the 0-based colour of its column: what a row-centric f! kernel needs to store
// the finished difference quotients of ANY pattern itself (the reference's decompression, ext/FiniteDiffSparseArraysExt.jl:38-47,
// asked from the other side: not "which row feeds this stored entry" but "which stored entries does this row feed").
//
//   k_rl_count    one thread per local column: rows of its entries -> atomic increments of the row counters
//   scan          exclusive prefix sum of the counters -> rowptr (three small kernels, int32)
//   k_rl_fill     one thread per local column: every entry takes the next free slot of its row (atomic cursor)
//   k_rl_finish   one thread per row: orders the row's entries by position (= by column), verifies that their colours are
//                 pairwise different (the VALID colouring the storing launch relies on), records the longest row
// Rows longer than kRlMaxRow entries (dense rows: they need as many colours as entries anyway) make the plan keep the hand-over
// path.  Included by fdjac_api.hip (namespace fdjac).

namespace fdjac {

constexpr int kRlMaxRow = 96;

template <typename IT>
__global__ void __launch_bounds__(kBlock) k_rl_count(const IT *__restrict__ colptr, const IT *__restrict__ rowval, int64_t base, int64_t col0, int64_t col1,
                                                     int64_t row0, int *__restrict__ cnt)
{
    const int64_t j = col0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= col1) return;
    const int64_t a = (int64_t)colptr[j] - base, b = (int64_t)colptr[j + 1] - base;
    for (int64_t q = a; q < b; ++q) atomicAdd(cnt + ((int64_t)rowval[q] - base - row0), 1);
}

// exclusive scan of n ints, 1024 per workgroup: local scan + block totals, scan of the totals (one workgroup), add
__global__ void __launch_bounds__(kBlock) k_rl_scan_local(int *__restrict__ a, int64_t n, int *__restrict__ totals)
{
    __shared__ int s_w[kBlock / 64];
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i0 + k < n ? a[i0 + k] : 0;
    const int mine = v[0] + v[1] + v[2] + v[3];
    int incl = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += s_w[w];
    int run = before + incl - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < n) a[i0 + k] = run;
        run += v[k];
    }
    if (threadIdx.x == kBlock - 1) totals[blockIdx.x] = before + incl;
}
__global__ void __launch_bounds__(kBlock) k_rl_scan_totals(int *__restrict__ totals, int64_t nb, int *__restrict__ grand)
{
    // one workgroup walks the block totals in chunks of kBlock (nb <= ~10^4 for 10^7 rows)
    __shared__ int s_w[kBlock / 64];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t c0 = 0; c0 < nb; c0 += kBlock) {
        const int64_t i = c0 + threadIdx.x;
        const int v = i < nb ? totals[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_w[w];
        if (i < nb) totals[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand = s_carry;
}
__global__ void __launch_bounds__(kBlock) k_rl_scan_add(int *__restrict__ a, int64_t n, const int *__restrict__ totals, const int *__restrict__ grand)
{
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    const int add = totals[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n) a[i0 + k] += add;
    if (blockIdx.x == 0 && threadIdx.x == 0) a[n] = *grand;      // rowptr[R] = number of entries
}

template <typename IT, typename CT>
__global__ void __launch_bounds__(kBlock) k_rl_fill(const IT *__restrict__ colptr, const IT *__restrict__ rowval, int64_t base, int64_t col0, int64_t col1,
                                                    int64_t row0, int64_t e0, const int *__restrict__ rowptr, int *__restrict__ cursor,
                                                    const CT *__restrict__ color, int *__restrict__ dest, CT *__restrict__ ecolor)
{
    const int64_t j = col0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= col1) return;
    const int64_t a = (int64_t)colptr[j] - base, b = (int64_t)colptr[j + 1] - base;
    const CT c = color[j];
    for (int64_t q = a; q < b; ++q) {
        const int64_t r = (int64_t)rowval[q] - base - row0;
        const int pos = rowptr[r] + atomicAdd(cursor + r, 1);
        dest[pos] = (int)(q - e0);
        ecolor[pos] = c;
    }
}

// flags[0]: a row with two entries of one colour (invalid colouring) or longer than kRlMaxRow; flags[1]: longest row
template <typename CT>
__global__ void __launch_bounds__(kBlock) k_rl_finish(const int *__restrict__ rowptr, int64_t R, int *__restrict__ dest, CT *__restrict__ ecolor,
                                                      int *__restrict__ flags)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= R) return;
    const int a = rowptr[r], n = rowptr[r + 1] - a;
    if (n <= 0) return;
    atomicMax(flags + 1, n);
    if (n > kRlMaxRow) { atomicOr(flags, 1); return; }
    // insertion sort by position in nzval (rows are short; the atomic cursors filled them in arrival order)
    for (int i = 1; i < n; ++i) {
        const int d = dest[a + i];
        const CT c = ecolor[a + i];
        int k = i - 1;
        while (k >= 0 && dest[a + k] > d) {
            dest[a + k + 1] = dest[a + k];
            ecolor[a + k + 1] = ecolor[a + k];
            --k;
        }
        dest[a + k + 1] = d;
        ecolor[a + k + 1] = c;
    }
    const CT none = (CT)(-1);
    bool dup = false;
    for (int i = 1; i < n && !dup; ++i) {
        const CT c = ecolor[a + i];
        if (c == none) continue;
        for (int k = 0; k < i; ++k) dup = dup || ecolor[a + k] == c;
    }
    if (dup) atomicOr(flags, 1);
}

// colptr_dev / rowval_dev: device pointers addressed with ABSOLUTE column / entry indices (as device_build_csc takes them).
// Leaves p->store_rl_ok = false (and frees what it allocated) if the table cannot serve: invalid colouring, a very long row,
// 2^31 or more local entries.  Errors of the HIP runtime are returned; "cannot serve" is FD_OK.
template <typename IT, typename CT>
static int build_store_table_t(fd_plan *p, const IT *colptr_dev, const IT *rowval_dev, int idx_base)
{
    hipStream_t s = p->ctx->stream;
    const int64_t n = p->nnz_local, R = p->row1 - p->row0, ncols = p->col1 - p->col0;
    int *d_cursor = nullptr, *d_totals = nullptr, *d_flags = nullptr;
    const int64_t nb = (R + 4 * kBlock - 1) / (4 * kBlock);
    auto cleanup = [&]() {
        if (d_cursor) (void)hipFree(d_cursor);
        if (d_totals) (void)hipFree(d_totals);
        if (d_flags) (void)hipFree(d_flags);
    };
    auto fail = [&](hipError_t e, const char *what) {
        set_error("%s failed: %s", what, hipGetErrorString(e));
        cleanup();
        return FD_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&p->d_rl_rowptr, sizeof(int) * (size_t)(R + 1))) != hipSuccess) return fail(e, "hipMalloc(rowptr)");
    if ((e = hipMalloc((void **)&p->d_rl_dest, sizeof(int) * (size_t)n)) != hipSuccess) return fail(e, "hipMalloc(dest)");
    if ((e = hipMalloc((void **)&p->d_rl_ecolor, sizeof(CT) * (size_t)n)) != hipSuccess) return fail(e, "hipMalloc(ecolor)");
    if ((e = hipMalloc((void **)&d_cursor, sizeof(int) * (size_t)R)) != hipSuccess) return fail(e, "hipMalloc(cursor)");
    if ((e = hipMalloc((void **)&d_totals, sizeof(int) * (size_t)(nb + 1))) != hipSuccess) return fail(e, "hipMalloc(totals)");
    if ((e = hipMalloc((void **)&d_flags, sizeof(int) * 2)) != hipSuccess) return fail(e, "hipMalloc(flags)");
    (void)hipMemsetAsync(p->d_rl_rowptr, 0, sizeof(int) * (size_t)(R + 1), s);
    (void)hipMemsetAsync(d_cursor, 0, sizeof(int) * (size_t)R, s);
    (void)hipMemsetAsync(d_flags, 0, sizeof(int) * 2, s);
    const unsigned gc = (unsigned)((ncols + kBlock - 1) / kBlock), gr = (unsigned)((R + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((k_rl_count<IT>), dim3(gc), dim3(kBlock), 0, s, colptr_dev, rowval_dev, (int64_t)idx_base, p->col0, p->col1, p->row0, p->d_rl_rowptr);
    hipLaunchKernelGGL(k_rl_scan_local, dim3((unsigned)nb), dim3(kBlock), 0, s, p->d_rl_rowptr, R, d_totals);
    hipLaunchKernelGGL(k_rl_scan_totals, dim3(1), dim3(kBlock), 0, s, d_totals, nb, d_totals + nb);
    hipLaunchKernelGGL(k_rl_scan_add, dim3((unsigned)nb), dim3(kBlock), 0, s, p->d_rl_rowptr, R, d_totals, d_totals + nb);
    hipLaunchKernelGGL((k_rl_fill<IT, CT>), dim3(gc), dim3(kBlock), 0, s, colptr_dev, rowval_dev, (int64_t)idx_base, p->col0, p->col1, p->row0,
                       p->entry_begin, p->d_rl_rowptr, d_cursor, (const CT *)p->d_color, p->d_rl_dest, (CT *)p->d_rl_ecolor);
    hipLaunchKernelGGL((k_rl_finish<CT>), dim3(gr), dim3(kBlock), 0, s, p->d_rl_rowptr, R, p->d_rl_dest, (CT *)p->d_rl_ecolor, d_flags);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "destination-table kernels");
    int flags[2] = {0, 0};
    if ((e = hipMemcpyAsync(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, s)) != hipSuccess) return fail(e, "hipMemcpyAsync");
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return fail(e, "hipStreamSynchronize");
    cleanup();
    if (flags[0]) {      // invalid colouring / very long rows: the hand-over path serves this plan
        (void)hipFree(p->d_rl_rowptr); (void)hipFree(p->d_rl_dest); (void)hipFree(p->d_rl_ecolor);
        p->d_rl_rowptr = p->d_rl_dest = nullptr;
        p->d_rl_ecolor = nullptr;
        return FD_OK;
    }
    p->rl_row0 = p->row0;
    p->rl_row1 = p->row1;
    p->rl_entries = n;
    p->rl_maxrow = flags[1];
    p->store_rl_ok = true;
    return FD_OK;
}

static int build_store_table(fd_plan *p, const void *colptr_dev, const void *rowval_dev, int idx_bytes, int idx_base)
{
    if (!p->want_table || p->kind != K_CSC || p->cx || p->store_ok || p->store5_ok || !p->store_allowed) return FD_OK;
    if (p->nnz_local <= 0 || p->nnz_local >= ((int64_t)1 << 31) || p->row1 - p->row0 >= ((int64_t)1 << 31) || !p->d_color) return FD_OK;
    if (idx_bytes == 8)
        return p->color8 ? build_store_table_t<int64_t, uint8_t>(p, (const int64_t *)colptr_dev, (const int64_t *)rowval_dev, idx_base)
                         : build_store_table_t<int64_t, int32_t>(p, (const int64_t *)colptr_dev, (const int64_t *)rowval_dev, idx_base);
    return p->color8 ? build_store_table_t<int32_t, uint8_t>(p, (const int32_t *)colptr_dev, (const int32_t *)rowval_dev, idx_base)
                     : build_store_table_t<int32_t, int32_t>(p, (const int32_t *)colptr_dev, (const int32_t *)rowval_dev, idx_base);
}

// the same from HOST arrays: the local slices are uploaded first
static int build_store_table_host(fd_plan *p, const void *colptr, const void *rowval, int idx_bytes, int idx_base)
{
    if (!p->want_table || p->kind != K_CSC || p->cx || p->store_ok || p->store5_ok || !p->store_allowed || p->nnz_local <= 0) return FD_OK;
    const size_t ib = (size_t)idx_bytes;
    const int64_t ncols = p->col1 - p->col0, e0 = p->entry_begin, n = p->nnz_local;
    void *d_cp = nullptr, *d_rv = nullptr;
    hipError_t e = hipMalloc(&d_cp, ib * (size_t)(ncols + 1));
    if (e == hipSuccess) e = hipMalloc(&d_rv, ib * (size_t)n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_cp, (const char *)colptr + ib * (size_t)p->col0, ib * (size_t)(ncols + 1), hipMemcpyHostToDevice, p->ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rv, (const char *)rowval + ib * (size_t)e0, ib * (size_t)n, hipMemcpyHostToDevice, p->ctx->stream);
    int rc = FD_OK;
    if (e != hipSuccess) {
        set_error("uploading the pattern for the destination table failed: %s", hipGetErrorString(e));
        rc = FD_ERR_HIP;
    } else {
        rc = build_store_table(p, (const char *)d_cp - ib * (size_t)p->col0, (const char *)d_rv - ib * (size_t)e0, idx_bytes, idx_base);
    }
    (void)hipStreamSynchronize(p->ctx->stream);
    if (d_cp) (void)hipFree(d_cp);
    if (d_rv) (void)hipFree(d_rv);
    return rc;
}

}  // namespace fdjac
