// The plan constructors of the C ABI (include/fdjac.h): one per `_colorediteration!` overload of the reference, the lowering of
// complex-valued x, and the wrappers that record content fingerprints (FD_PLAN_FINGERPRINT).
// Included by fdjac_api.hip (inside extern "C").
static int csc_common(fd_ctx *ctx, int kind, int64_t M, int64_t N, const void *colptr, const void *rowval,
                      int idx_bytes, int idx_base, const void *colorvec, int color_bytes,
                      const fd_plan_opts *opts, fd_plan **out, bool device_declined = false)
{
    FD_REQUIRE(colptr && rowval, FD_ERR_ARG, "colptr/rowval is NULL");
    FD_REQUIRE(idx_bytes == 4 || idx_bytes == 8, FD_ERR_ARG, "idx_bytes must be 4 or 8");
    FD_REQUIRE(idx_base == 0 || idx_base == 1, FD_ERR_ARG, "idx_base must be 0 or 1");
    int rc = new_plan(ctx, kind, M, N, out);
    if (rc) return rc;
    fd_plan *p = *out;
    FD_TRY(apply_opts(p, opts));
    PbTimer tm(ctx->stream);
    std::vector<int32_t> col0;
    if (!(colorvec != nullptr && (color_bytes == 4 || color_bytes == 8))) FD_TRY(ingest_colors(p, colorvec, color_bytes, col0));   // (reports the argument error)
    const int64_t e0 = load_idx(colptr, idx_bytes, p->col0) - idx_base;
    const int64_t e1 = load_idx(colptr, idx_bytes, p->col1) - idx_base;
    if (!(e0 >= 0 && e1 >= e0)) {
        set_error("colptr is not monotone");
        fd_plan_destroy(p);
        *out = nullptr;
        return FD_ERR_SHAPE;
    }
    p->entry_begin = e0;
    // Large common-pattern plans are compiled on the device: the raw arrays are uploaded as they are (the caller's index
    // width and base) and the kernels of fdjac_planbuild.hip produce the plan arrays.  FDJAC_PLAN_DEVICE=0 keeps the
    // host loops below (the checker: tests compare both builds bit for bit); patterns the device builder declines
    // (scattered stencils, many colours) fall through to them as well.
    {
        const char *pd = fdjac::test_switch("FDJAC_PLAN_DEVICE");
        const int want = (pd && *pd) ? atoi(pd) : -1;      // -1 auto (>= 2^17 entries), 0 never, 1 whenever possible
        // (device_declined: fd_plan_create_csc_device already ran the device builder on this pattern and it declined)
        if (kind == K_CSC && want != 0 && !device_declined && (want == 1 || e1 - e0 >= ((int64_t)1 << 17))) {
            const size_t ib = (size_t)idx_bytes;
            void *d_cp = nullptr, *d_rv = nullptr, *d_cv = nullptr;
            const int64_t ncols = p->col1 - p->col0;
            bool ok = hipMalloc(&d_cp, ib * (size_t)(ncols + 1)) == hipSuccess && hipMalloc(&d_rv, ib * (size_t)std::max<int64_t>(e1 - e0, 1)) == hipSuccess &&
                      hipMalloc(&d_cv, (size_t)color_bytes * (size_t)N) == hipSuccess;
            ok = ok && hipMemcpyAsync(d_cp, (const char *)colptr + ib * (size_t)p->col0, ib * (size_t)(ncols + 1), hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                 hipMemcpyAsync(d_rv, (const char *)rowval + ib * (size_t)e0, ib * (size_t)(e1 - e0), hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                 hipMemcpyAsync(d_cv, colorvec, (size_t)color_bytes * (size_t)N, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
            int brc = FD_OK, res = PBR_DECLINED;
            if (ok)   // (kernels index with absolute column / entry numbers: shift the slice bases accordingly)
                res = device_build_csc(p, (const char *)d_cp - ib * (size_t)p->col0, (const char *)d_rv - ib * (size_t)e0, idx_bytes, idx_base,
                                       d_cv, color_bytes, e0, e1, &brc);
            (void)hipStreamSynchronize(ctx->stream);
            tm.mark(res == PBR_DONE ? "csc: device builder" : "csc: device builder declined");
            if (res == PBR_DONE && brc == FD_OK) {
                // (the raw arrays are still on the device: the compact copy comes from them -- unless the plan has a column window: its
                //  colouring is checked against ALL columns, which only the host holds here)
                if (p->col0 == 0 && p->col1 == p->N)
                    brc = build_store_csc(p, (const char *)d_cp - ib * (size_t)p->col0, (const char *)d_rv - ib * (size_t)e0, idx_bytes, idx_base, true);
                else
                    brc = build_store_csc_host(p, colptr, rowval, idx_bytes, idx_base);
            }
            if (d_cp) (void)hipFree(d_cp);
            if (d_rv) (void)hipFree(d_rv);
            if (d_cv) (void)hipFree(d_cv);
            (void)hipGetLastError();
            if (res == PBR_DONE) {
                if (brc != FD_OK) { fd_plan_destroy(p); *out = nullptr; return brc; }
                p->nouts = 1;
                p->out_len[0] = e1 - e0;
                return FD_OK;
            }
        }
    }
    FD_TRY(ingest_colors(p, colorvec, color_bytes, col0));
    tm.mark("csc: colours (host)");
    std::vector<int32_t> rows((size_t)(e1 - e0)), nzc((size_t)(e1 - e0));
    std::vector<int64_t> dest;
    if (kind == K_CSC_DENSE) dest.resize((size_t)(e1 - e0));
    for (int64_t j = p->col0; j < p->col1; ++j) {
        const int64_t a = load_idx(colptr, idx_bytes, j) - idx_base, b = load_idx(colptr, idx_bytes, j + 1) - idx_base;
        if (!(a <= b && a >= e0 && b <= e1)) {
            set_error("colptr is not monotone at column %lld", (long long)j);
            fd_plan_destroy(p);
            *out = nullptr;
            return FD_ERR_SHAPE;
        }
        for (int64_t q = a; q < b; ++q) {
            const int64_t r = load_idx(rowval, idx_bytes, q) - idx_base;
            if (r < 0 || r >= M) {
                set_error("rowval[%lld] = %lld outside 1..%lld", (long long)q, (long long)(r + idx_base), (long long)M);
                fd_plan_destroy(p);
                *out = nullptr;
                return FD_ERR_SHAPE;
            }
            rows[(size_t)(q - e0)] = (int32_t)r;
            nzc[(size_t)(q - e0)] = col0[(size_t)j];
            if (kind == K_CSC_DENSE) dest[(size_t)(q - e0)] = r + M * j;
        }
    }
    std::vector<int64_t> colstart;
    if (kind == K_CSC) {
        colstart.resize((size_t)(p->col1 - p->col0) + 1);
        for (int64_t j = p->col0; j <= p->col1; ++j) colstart[(size_t)(j - p->col0)] = load_idx(colptr, idx_bytes, j) - idx_base - e0;
    }
    tm.mark("csc: entry lists (host)");
    FD_TRY(finish_list_plan(p, col0, rows, nzc, dest, kind == K_CSC ? &colstart : nullptr));
    tm.mark("csc: list plan (total)");
    if (kind == K_CSC) FD_TRY(build_store_csc_host(p, colptr, rowval, idx_bytes, idx_base));
    tm.mark("csc: compact pattern copy");
    p->nouts = 1;
    p->out_len[0] = kind == K_CSC ? (e1 - e0) : M * N;
    return FD_OK;
}

// ---- complex-valued x (FD_PLAN_COMPLEX_X; returntype <: Complex with Val(:forward) / Val(:central), src/jacobians.jl:94-128,
// 537-622, test/finitedifftests.jl:480-513).  The reference's loop is generic in eltype(x): the masked norm is over complex
// elements (|x_j|^2), epsilon is REAL, x1 .+= epsilon * mask perturbs the real parts, f! runs on complex arrays, the quotient is a
// complex number divided by a real one, J is complex.  Seen as reals that IS a real problem of twice the size: element 2j / 2j+1 =
// re / im of x_j, row 2r / 2r+1 = re / im of f_r; only the even columns carry colours; the stored entry (r, j) becomes the two
// entries (2r, 2j), (2r+1, 2j) -- consecutive in every storage order the plans write, i.e. exactly the (re, im) layout of a
// Complex nzval / dense J.  So the plan is built for that real problem (the thread-local marker t_lowered_cx says so: pair norms in
// the step-size kernels, f! called with is_complex = 1) and every kernel of the real path serves it unchanged.
struct LoweredCx {
    std::vector<int64_t> colptr, rows, cols, dest, colors;
    fd_plan_opts opts;
};
static int lower_colors_opts(int64_t N, const void *colorvec, int color_bytes, const fd_plan_opts *opts, LoweredCx &L)
{
    FD_REQUIRE(opts != nullptr, FD_ERR_ARG, "opts is NULL");
    FD_REQUIRE(opts->fdtype != FD_COMPLEX, FD_ERR_UNSUPPORTED, "fdtype_error: Val(:complex) needs a real returntype (src/jacobians.jl:106)");
    FD_REQUIRE(colorvec != nullptr && (color_bytes == 4 || color_bytes == 8), FD_ERR_ARG, "colorvec is NULL / color_bytes must be 4 or 8");
    L.colors.assign((size_t)(2 * N), 0);
    for (int64_t j = 0; j < N; ++j) L.colors[(size_t)(2 * j)] = load_idx(colorvec, color_bytes, j);   // (odd = imaginary parts: colour 0, never perturbed)
    L.opts = *opts;
    FD_REQUIRE((opts->flags & ~kPlanKnownFlags) == 0, FD_ERR_ARG, "unknown fd_plan_opts.flags bits 0x%x", (unsigned)(opts->flags & ~kPlanKnownFlags));
    L.opts.flags = opts->flags & ~(FD_PLAN_COMPLEX_X | FD_PLAN_FINGERPRINT);   // (fingerprints: of the caller's arrays, by the public entry point)
    L.opts.col_begin *= 2; L.opts.col_end *= 2; L.opts.x_begin *= 2; L.opts.x_end *= 2;
    return FD_OK;
}
static int lowered_csc(fd_ctx *ctx, int kind_dense, int64_t M, int64_t N, const void *colptr, const void *rowval, int idx_bytes, int idx_base,
                       const void *colorvec, int color_bytes, const fd_plan_opts *opts, fd_plan **out)
{
    FD_REQUIRE(colptr && rowval && out, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(idx_bytes == 4 || idx_bytes == 8, FD_ERR_ARG, "idx_bytes must be 4 or 8");
    LoweredCx L;
    int rc = lower_colors_opts(N, colorvec, color_bytes, opts, L);
    if (rc) return rc;
    const int64_t nnz = load_idx(colptr, idx_bytes, N) - idx_base;
    FD_REQUIRE(nnz >= 0, FD_ERR_SHAPE, "colptr is not monotone");
    if (!kind_dense) {   // common-pattern CSC: column 2j holds (2r, 2r+1) for every row r of column j, column 2j+1 nothing
        L.colptr.resize((size_t)(2 * N + 1));
        L.rows.resize((size_t)(2 * nnz));
        for (int64_t j = 0; j < N; ++j) {
            const int64_t a = load_idx(colptr, idx_bytes, j) - idx_base, b = load_idx(colptr, idx_bytes, j + 1) - idx_base;
            FD_REQUIRE(a >= 0 && a <= b && b <= nnz, FD_ERR_SHAPE, "colptr is not monotone at column %lld", (long long)j);
            L.colptr[(size_t)(2 * j)] = 2 * a;
            L.colptr[(size_t)(2 * j + 1)] = 2 * b;
            for (int64_t q = a; q < b; ++q) {
                const int64_t r = load_idx(rowval, idx_bytes, q) - idx_base;
                FD_REQUIRE(r >= 0 && r < M, FD_ERR_SHAPE, "rowval[%lld] outside 1..%lld", (long long)q, (long long)M);
                L.rows[(size_t)(2 * q)] = 2 * r;
                L.rows[(size_t)(2 * q + 1)] = 2 * r + 1;
            }
        }
        L.colptr[(size_t)(2 * N)] = 2 * nnz;
        LoweredScope lowered;
        return csc_common(ctx, K_CSC, 2 * M, 2 * N, L.colptr.data(), L.rows.data(), 8, 0, L.colors.data(), 8, &L.opts, out, true);
    }
    // dense complex J (M x N column-major = 2M x N reals): explicit destinations
    L.rows.resize((size_t)(2 * nnz)); L.cols.resize((size_t)(2 * nnz)); L.dest.resize((size_t)(2 * nnz));
    for (int64_t j = 0; j < N; ++j)
        for (int64_t q = load_idx(colptr, idx_bytes, j) - idx_base; q < load_idx(colptr, idx_bytes, j + 1) - idx_base; ++q) {
            const int64_t r = load_idx(rowval, idx_bytes, q) - idx_base;
            FD_REQUIRE(q >= 0 && q < nnz && r >= 0 && r < M, FD_ERR_SHAPE, "inconsistent pattern");
            for (int h = 0; h < 2; ++h) { L.rows[(size_t)(2 * q + h)] = 2 * r + h; L.cols[(size_t)(2 * q + h)] = 2 * j; L.dest[(size_t)(2 * q + h)] = 2 * r + h + 2 * M * j; }
        }
    LoweredScope lowered;
    return fd_plan_create_entries(ctx, 2 * M, 2 * N, L.rows.data(), L.cols.data(), L.dest.data(), 2 * nnz, 2 * M * N, 8, 0, L.colors.data(), 8, &L.opts, out);
}
static int lowered_coo(fd_ctx *ctx, int64_t M, int64_t N, const void *rows_index, const void *cols_index, const int64_t *dest, int64_t nnz,
                       int64_t out_len, int idx_bytes, int idx_base, const void *colorvec, int color_bytes, const fd_plan_opts *opts, fd_plan **out)
{
    FD_REQUIRE((rows_index && cols_index) || nnz == 0, FD_ERR_ARG, "NULL index list");
    FD_REQUIRE(idx_bytes == 4 || idx_bytes == 8, FD_ERR_ARG, "idx_bytes must be 4 or 8");
    LoweredCx L;
    int rc = lower_colors_opts(N, colorvec, color_bytes, opts, L);
    if (rc) return rc;
    L.rows.resize((size_t)(2 * nnz)); L.cols.resize((size_t)(2 * nnz)); L.dest.resize((size_t)(2 * nnz));
    for (int64_t q = 0; q < nnz; ++q) {
        const int64_t r = load_idx(rows_index, idx_bytes, q) - idx_base, c = load_idx(cols_index, idx_bytes, q) - idx_base;
        FD_REQUIRE(r >= 0 && r < M && c >= 0 && c < N, FD_ERR_SHAPE, "entry %lld outside the matrix", (long long)q);
        const int64_t d = dest ? dest[q] : r + M * c;       // (complex elements)
        for (int h = 0; h < 2; ++h) { L.rows[(size_t)(2 * q + h)] = 2 * r + h; L.cols[(size_t)(2 * q + h)] = 2 * c; L.dest[(size_t)(2 * q + h)] = 2 * d + h; }
    }
    LoweredScope lowered;
    return fd_plan_create_entries(ctx, 2 * M, 2 * N, L.rows.data(), L.cols.data(), L.dest.data(), 2 * nnz, 2 * out_len, 8, 0, L.colors.data(), 8, &L.opts, out);
}

// Complex-valued x on STRUCTURED storage (Tridiagonal, BandedMatrix, BlockBandedMatrix: src/jacobians.jl:94-128, 537-622 are generic in
// the matrix type too): the storage is enumerated once as (row, column, position) triples in complex elements and lowered like any
// entry list; a Tridiagonal's three arrays are one concatenated output (dl | d | du) that the call splits afterwards.
static int lowered_structured(fd_ctx *ctx, int64_t M, int64_t N, const std::vector<int64_t> &rows, const std::vector<int64_t> &cols,
                              const std::vector<int64_t> &dest, int64_t out_len, const void *colorvec, int color_bytes, const fd_plan_opts *opts,
                              fd_plan **out)
{
    FD_REQUIRE(opts->col_begin == 0 && (opts->col_end == 0 || opts->col_end == N), FD_ERR_UNSUPPORTED,
               "column windows are not supported for complex-valued x on structured storage");
    return lowered_coo(ctx, M, N, rows.data(), cols.data(), dest.data(), (int64_t)rows.size(), out_len, 8, 0, colorvec, color_bytes, opts, out);
}

int fd_plan_create_csc(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr, const void *rowval, int idx_bytes,
                       int idx_base, const void *colorvec, int color_bytes, const fd_plan_opts *opts,
                       fd_plan **out)
{
    const int rc = (opts && (opts->flags & FD_PLAN_COMPLEX_X))
                       ? lowered_csc(ctx, 0, M, N, colptr, rowval, idx_bytes, idx_base, colorvec, color_bytes, opts, out)
                       : csc_common(ctx, K_CSC, M, N, colptr, rowval, idx_bytes, idx_base, colorvec, color_bytes, opts, out);
    return finish_fingerprint(rc, out, opts, 1, colptr, N + 1, rowval, std::numeric_limits<int64_t>::max(), idx_bytes, idx_base, colorvec,
                              color_bytes, N, FD_HOST, N);
}

static int csc_device_impl(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr_dev, const void *rowval_dev,
                           int idx_bytes, int idx_base, const void *colorvec_dev, int color_bytes,
                           const fd_plan_opts *opts, fd_plan **out)
{
    FD_REQUIRE(colptr_dev && rowval_dev && colorvec_dev, FD_ERR_ARG, "NULL pattern array");
    FD_REQUIRE(idx_bytes == 4 || idx_bytes == 8, FD_ERR_ARG, "idx_bytes must be 4 or 8");
    FD_REQUIRE(idx_base == 0 || idx_base == 1, FD_ERR_ARG, "idx_base must be 0 or 1");
    FD_REQUIRE(color_bytes == 4 || color_bytes == 8, FD_ERR_ARG, "color_bytes must be 4 or 8");
    int rc = new_plan(ctx, K_CSC, M, N, out);
    if (rc) return rc;
    fd_plan *p = *out;
    FD_TRY(apply_opts(p, opts));
    // the two colptr values that bound the local columns
    int64_t cp[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {
        const int64_t j = k ? p->col1 : p->col0;
        int64_t v64 = 0;
        int32_t v32 = 0;
        // (on the context's stream: the caller's kernels that PRODUCE the pattern on that stream are then complete -- a blocking
        //  hipMemcpy runs on the null stream, which a non-blocking stream does not wait for)
        hipError_t e = idx_bytes == 8 ? hipMemcpyAsync(&v64, (const char *)colptr_dev + 8 * (size_t)j, 8, hipMemcpyDeviceToHost, ctx->stream)
                                      : hipMemcpyAsync(&v32, (const char *)colptr_dev + 4 * (size_t)j, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { set_error("reading colptr from the device failed: %s", hipGetErrorString(e)); fd_plan_destroy(p); *out = nullptr; return FD_ERR_HIP; }
        cp[k] = (idx_bytes == 8 ? v64 : (int64_t)v32) - idx_base;
    }
    if (!(cp[0] >= 0 && cp[1] >= cp[0])) { set_error("colptr is not monotone"); fd_plan_destroy(p); *out = nullptr; return FD_ERR_SHAPE; }
    p->entry_begin = cp[0];
    int brc = FD_OK;
    const char *pd = fdjac::test_switch("FDJAC_PLAN_DEVICE");
    int res = (pd && *pd && atoi(pd) == 0) ? (int)PBR_DECLINED
                                           : device_build_csc(p, colptr_dev, rowval_dev, idx_bytes, idx_base, colorvec_dev, color_bytes, cp[0], cp[1], &brc);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipGetLastError();
    if (res == PBR_DONE) {
        if (brc == FD_OK) brc = build_store_csc(p, colptr_dev, rowval_dev, idx_bytes, idx_base, true);
        if (brc != FD_OK) { fd_plan_destroy(p); *out = nullptr; return brc; }
        p->nouts = 1;
        p->out_len[0] = cp[1] - cp[0];
        return FD_OK;
    }
    // declined (scattered pattern, many colours, forced variants): bring the pattern to the host once and build there
    fd_plan_destroy(p);
    *out = nullptr;
    const size_t ib = (size_t)idx_bytes;
    std::vector<char> h_cp(ib * (size_t)(N + 1)), h_cv((size_t)color_bytes * (size_t)N);
    FD_HIP_CHECK(hipMemcpy(h_cp.data(), colptr_dev, h_cp.size(), hipMemcpyDeviceToHost));
    FD_HIP_CHECK(hipMemcpy(h_cv.data(), colorvec_dev, h_cv.size(), hipMemcpyDeviceToHost));
    const int64_t nnz_all = load_idx(h_cp.data(), idx_bytes, N) - idx_base;
    FD_REQUIRE(nnz_all >= 0, FD_ERR_SHAPE, "colptr is not monotone");
    std::vector<char> h_rv(ib * (size_t)std::max<int64_t>(nnz_all, 1));
    if (nnz_all > 0) FD_HIP_CHECK(hipMemcpy(h_rv.data(), rowval_dev, ib * (size_t)nnz_all, hipMemcpyDeviceToHost));
    return csc_common(ctx, K_CSC, M, N, h_cp.data(), h_rv.data(), idx_bytes, idx_base, h_cv.data(), color_bytes, opts, out, true);
}

// FNV-1a over the plan's compiled pattern (device arrays copied back) and its scalar parameters: two plans with the
// same checksum drive the kernels identically.  Diagnostic / test entry point (the device builder is checked against
// the host builder with it).
int fd_plan_checksum(fd_plan *p, uint64_t *out)
{
    FD_REQUIRE(p && out, FD_ERR_ARG, "NULL argument");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    FD_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *data, size_t n) {
        const unsigned char *b = (const unsigned char *)data;
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    };
    const char *tr = getenv("FDJAC_CHECKSUM_TRACE");      // the running hash after every array, on stderr (which array differs?)
    const bool trace = tr && *tr && atoi(tr) != 0;
    int part = 0;
    auto mix_dev = [&](const void *d, size_t n) -> int {
        ++part;
        if (!d || !n) return FD_OK;
        std::vector<char> tmp(n);
        FD_HIP_CHECK(hipMemcpy(tmp.data(), d, n, hipMemcpyDeviceToHost));
        mix(tmp.data(), n);
        if (trace) fprintf(stderr, "[fdjac checksum] array %d (%zu bytes): %016llx\n", part, n, (unsigned long long)h);
        return FD_OK;
    };
    const int64_t scal[] = {p->kind, p->fdtype, p->M, p->N, p->C, p->color8, p->col0, p->col1, p->row0, p->row1, p->nnz_local,
                            p->entry_begin, p->window, p->window2d, p->sorted_gather, p->win_tile, p->win_pairs, p->win_ncol,
                            p->win_per_P, p->win_per_S, p->win_per_magic, p->has_none, p->cyc_C, p->cyc_shift,
                            p->n_partial_blocks, p->chunkB, p->nchunks, (int64_t)(p->win_overread * 1e6),
                            p->band_off, p->band_C, p->band_w, p->band_u, p->band_shift,
                            (int64_t)p->band_mw, (int64_t)p->band_mc, p->bd_t0, p->bd_t1, p->store_ok, p->store_l, p->store_u,
                            p->store5_ok, p->store5_nx, p->store5_ny};
    mix(scal, sizeof scal);
    if (trace) {
        fprintf(stderr, "[fdjac checksum] scalars: %016llx :", (unsigned long long)h);
        for (int64_t v : scal) fprintf(stderr, " %lld", (long long)v);
        fprintf(stderr, "\n");
    }
    int rc;
    if ((rc = mix_dev(p->d_color, (size_t)p->N * (p->color8 ? 1 : 4)))) return rc;
    if (p->window2d) {
        const int64_t w2[] = {p->w2_ntiles, p->w2_codes};
        mix(w2, sizeof w2);
        if ((rc = mix_dev(p->d_w2desc, sizeof(int) * (size_t)kW2Desc * (size_t)p->w2_ntiles))) return rc;
        if ((rc = mix_dev(p->d_wcode, sizeof(uint16_t) * (size_t)p->w2_codes))) return rc;
    }
    if (p->window && !p->window2d) {
        const size_t padded = (size_t)round_up(std::max<int64_t>(p->nnz_local, 1), kListPad);
        if ((rc = mix_dev(p->d_wtiles, sizeof(int4) * 3 * (padded / (size_t)p->win_tile)))) return rc;
        if ((rc = mix_dev(p->d_wcode, sizeof(uint16_t) * padded))) return rc;
    }
    if (!p->window && p->kind == K_CSC && p->d_rowval) {          // index lists (sorted-gather or plain)
        const size_t padded = (size_t)round_up(std::max<int64_t>(p->nnz_local, 1), kListPad), ntiles = padded / kSortTile;
        const size_t nreal = (size_t)((p->nnz_local + kSortTile - 1) / kSortTile);
        const int64_t have[] = {p->d_spos != nullptr, p->d_fxwin != nullptr, p->d_tile_order != nullptr, (int64_t)(p->lines_direct * 1e6),
                                (int64_t)(p->lines_sorted * 1e6)};
        mix(have, sizeof have);
        if ((rc = mix_dev(p->d_rowval, sizeof(int32_t) * padded))) return rc;
        if ((rc = mix_dev(p->d_nzcolor, (size_t)(p->color8 ? 1 : 4) * padded))) return rc;
        if ((rc = mix_dev(p->d_spos, sizeof(uint16_t) * padded))) return rc;
        if ((rc = mix_dev(p->d_fxwin, sizeof(int32_t) * 2 * kFxWin * ntiles))) return rc;
        if ((rc = mix_dev(p->d_tile_order, sizeof(int32_t) * nreal))) return rc;
    }
    *out = h;
    return FD_OK;
}

int fd_plan_create_csc_dense(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr, const void *rowval,
                             int idx_bytes, int idx_base, const void *colorvec, int color_bytes,
                             const fd_plan_opts *opts, fd_plan **out)
{
    if (opts && !(opts->col_begin == 0 && opts->col_end == 0) && !(opts->col_begin == 0 && opts->col_end == N)) {
        set_error("column windows are not supported for dense J");
        return FD_ERR_UNSUPPORTED;
    }
    const int rc = (opts && (opts->flags & FD_PLAN_COMPLEX_X))
                       ? lowered_csc(ctx, 1, M, N, colptr, rowval, idx_bytes, idx_base, colorvec, color_bytes, opts, out)
                       : csc_common(ctx, K_CSC_DENSE, M, N, colptr, rowval, idx_bytes, idx_base, colorvec, color_bytes, opts, out);
    return finish_fingerprint(rc, out, opts, 1, colptr, N + 1, rowval, std::numeric_limits<int64_t>::max(), idx_bytes, idx_base, colorvec,
                              color_bytes, N, FD_HOST, N);
}

static int entries_common(fd_ctx *ctx, int64_t M, int64_t N, const void *rows_index, const void *cols_index,
                          const int64_t *dest_in, int64_t nnz, int64_t out_len, int idx_bytes, int idx_base,
                          const void *colorvec, int color_bytes, const fd_plan_opts *opts, fd_plan **out)
{
    FD_REQUIRE((rows_index && cols_index) || nnz == 0, FD_ERR_ARG, "rows_index/cols_index is NULL");
    FD_REQUIRE(idx_bytes == 4 || idx_bytes == 8, FD_ERR_ARG, "idx_bytes must be 4 or 8");
    FD_REQUIRE(idx_base == 0 || idx_base == 1, FD_ERR_ARG, "idx_base must be 0 or 1");
    FD_REQUIRE(nnz >= 0 && out_len >= 0, FD_ERR_ARG, "nnz/out_len < 0");
    if (opts && !(opts->col_begin == 0 && opts->col_end == 0) && !(opts->col_begin == 0 && opts->col_end == N)) {
        set_error("column windows are not supported for entry-list plans");
        return FD_ERR_UNSUPPORTED;
    }
    int rc = new_plan(ctx, K_COO_DENSE, M, N, out);
    if (rc) return rc;
    fd_plan *p = *out;
    FD_TRY(apply_opts(p, opts));
    std::vector<int32_t> col0;
    FD_TRY(ingest_colors(p, colorvec, color_bytes, col0));
    std::vector<int32_t> rows((size_t)nnz), nzc((size_t)nnz);
    std::vector<int64_t> dest((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) {
        const int64_t r = load_idx(rows_index, idx_bytes, k) - idx_base, c = load_idx(cols_index, idx_bytes, k) - idx_base;
        const int64_t d = dest_in ? dest_in[k] : r + M * c;
        if (r < 0 || r >= M || c < 0 || c >= N || d < 0 || d >= out_len) {
            set_error("entry %lld: index (%lld,%lld) / destination %lld outside the %lld x %lld matrix / %lld values",
                      (long long)k, (long long)(r + idx_base), (long long)(c + idx_base), (long long)d, (long long)M,
                      (long long)N, (long long)out_len);
            fd_plan_destroy(p);
            *out = nullptr;
            return FD_ERR_SHAPE;
        }
        rows[(size_t)k] = (int32_t)r;
        nzc[(size_t)k] = col0[(size_t)c];
        dest[(size_t)k] = d;
    }
    FD_TRY(finish_list_plan(p, col0, rows, nzc, dest));
    p->nouts = 1;
    p->out_len[0] = out_len;
    return FD_OK;
}

int fd_plan_create_coo_dense(fd_ctx *ctx, int64_t M, int64_t N, const void *rows_index, const void *cols_index,
                             int64_t nnz, int idx_bytes, int idx_base, const void *colorvec, int color_bytes,
                             const fd_plan_opts *opts, fd_plan **out)
{
    const int rc = (opts && (opts->flags & FD_PLAN_COMPLEX_X))
                       ? lowered_coo(ctx, M, N, rows_index, cols_index, nullptr, nnz, M * N, idx_bytes, idx_base, colorvec, color_bytes, opts, out)
                       : entries_common(ctx, M, N, rows_index, cols_index, nullptr, nnz, M * N, idx_bytes, idx_base, colorvec, color_bytes, opts, out);
    return finish_fingerprint(rc, out, opts, 2, rows_index, nnz, cols_index, nnz, idx_bytes, idx_base, colorvec, color_bytes, N, FD_HOST, N);
}

int fd_plan_create_entries(fd_ctx *ctx, int64_t M, int64_t N, const void *rows_index, const void *cols_index,
                           const int64_t *dest, int64_t nnz, int64_t out_len, int idx_bytes, int idx_base,
                           const void *colorvec, int color_bytes, const fd_plan_opts *opts, fd_plan **out)
{
    FD_REQUIRE(dest || nnz == 0, FD_ERR_ARG, "dest is NULL");
    const int rc = (opts && (opts->flags & FD_PLAN_COMPLEX_X))
                       ? lowered_coo(ctx, M, N, rows_index, cols_index, dest, nnz, out_len, idx_bytes, idx_base, colorvec, color_bytes, opts, out)
                       : entries_common(ctx, M, N, rows_index, cols_index, dest, nnz, out_len, idx_bytes, idx_base, colorvec, color_bytes, opts, out);
    return finish_fingerprint(rc, out, opts, 2, rows_index, nnz, cols_index, nnz, idx_bytes, idx_base, colorvec, color_bytes, N, FD_HOST, N);
}

static int tridiagonal_impl(fd_ctx *ctx, int64_t N, const void *colorvec, int color_bytes, const fd_plan_opts *opts, fd_plan **out)
{
    int rc = new_plan(ctx, K_TRIDIAG, N, N, out);
    if (rc) return rc;
    fd_plan *p = *out;
    FD_TRY(apply_opts(p, opts));
    std::vector<int32_t> col0;
    {
        // large problems: the colours are converted and tested on the device (FDJAC_PLAN_DEVICE=0: host loops, the checker)
        const char *pd = fdjac::test_switch("FDJAC_PLAN_DEVICE");
        const int want = (pd && *pd) ? atoi(pd) : -1;
        int res = PBR_DECLINED;
        if (want != 0 && (want == 1 || N >= ((int64_t)1 << 17)) && colorvec && (color_bytes == 4 || color_bytes == 8))
            res = device_colors_only(p, colorvec, color_bytes);
        (void)hipGetLastError();
        if (res != PBR_DONE) {
            FD_TRY(ingest_colors(p, colorvec, color_bytes, col0));
            FD_TRY(upload_colors(p, col0, {}));
        }
    }
    p->row0 = std::max<int64_t>(p->col0 - 1, 0);
    p->row1 = std::min<int64_t>(p->col1 + 1, N);
    if (p->col1 == p->col0) p->row0 = p->row1 = 0;
    {
        // row-window variant: few colours (every loaded f! value is used when C == 3), an even first column (16-B
        // aligned pairs); otherwise the gather kernel.  FDJAC_WINDOW=0 forces the gather kernel.
        const char *fw = fdjac::test_switch("FDJAC_WINDOW");
        p->tri_window = !(fw && *fw && atoi(fw) == 0) && p->C <= 4 && (p->col0 % 2) == 0 && p->col1 > p->col0;
    }
    FD_TRY(alloc_scratch(p, col0));
    p->nouts = 3;
    const int64_t j0 = p->col0, j1 = p->col1;
    p->out_len[1] = j1 - j0;                                           // d
    p->out_len[0] = std::max<int64_t>(std::min<int64_t>(j1, N - 1) - j0, 0);  // dl
    p->out_len[2] = j1 > j0 ? (j1 - 1) - std::max<int64_t>(j0 - 1, 0) : 0;     // du
    store_caps_implicit_band(p, 1, 1);
    return FD_OK;
}

static int banded_impl(fd_ctx *ctx, int64_t M, int64_t N, int64_t l, int64_t u, const void *colorvec,
                       int color_bytes, const fd_plan_opts *opts, fd_plan **out)
{
    FD_REQUIRE(l + u + 1 >= 1 && l > -N && u > -M, FD_ERR_ARG, "bad bandwidths (%lld,%lld)", (long long)l, (long long)u);
    int rc = new_plan(ctx, K_BANDED, M, N, out);
    if (rc) return rc;
    fd_plan *p = *out;
    p->l = l;
    p->u = u;
    FD_TRY(apply_opts(p, opts));
    p->row0 = std::min<int64_t>(std::max<int64_t>(p->col0 - u, 0), M);
    p->row1 = std::max<int64_t>(std::min<int64_t>(p->col1 + l, M), p->row0);
    p->nouts = 1;
    p->out_len[0] = (p->col1 - p->col0) * (l + u + 1);
    {
        // large narrow bands are compiled on the device like a banded SparseMatrixCSC (fdjac_planbuild.hip, BAND tiles:
        // no index arrays at all); FDJAC_PLAN_DEVICE=0 keeps the host loops below, which are also its checker
        const char *pd = fdjac::test_switch("FDJAC_PLAN_DEVICE");
        const int want = (pd && *pd) ? atoi(pd) : -1;
        const int64_t w = l + u + 1, slots = p->out_len[0];
        if (want != 0 && w <= 64 && (want == 1 ? slots > 0 : slots >= ((int64_t)1 << 17)) && colorvec && (color_bytes == 4 || color_bytes == 8)) {
            void *d_cv = nullptr;
            bool ok = hipMalloc(&d_cv, (size_t)color_bytes * (size_t)N) == hipSuccess &&
                      hipMemcpyAsync(d_cv, colorvec, (size_t)color_bytes * (size_t)N, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
            int brc = FD_OK, res = PBR_DECLINED;
            const PbBand band{w, u};
            if (ok) res = device_build_csc(p, nullptr, nullptr, 4, 0, d_cv, color_bytes, 0, slots, &brc, &band);
            (void)hipStreamSynchronize(ctx->stream);
            if (d_cv) (void)hipFree(d_cv);
            (void)hipGetLastError();
            if (res == PBR_DONE) {
                if (brc != FD_OK) { fd_plan_destroy(p); *out = nullptr; return brc; }
                store_caps_implicit_band(p, l, u);
                return FD_OK;
            }
        }
    }
    std::vector<int32_t> col0;
    FD_TRY(ingest_colors(p, colorvec, color_bytes, col0));
    FD_TRY(upload_colors(p, col0, {}));
    {
        // the band's column-major storage as an entry list with implicit indices (slot k of column j <-> row j-u+k;
        // slots outside the matrix and columns without colour are written as 0): narrow bands go through the
        // row-window kernel, exactly like a banded SparseMatrixCSC
        const int64_t w = l + u + 1, slots = p->out_len[0];
        if (slots > 0 && w <= 64) {
            const size_t padded = (size_t)round_up(slots, kListPad);
            std::vector<int32_t> rows(padded, 0), nzc(padded, -2);
            size_t e = 0;
            for (int64_t j = p->col0; j < p->col1; ++j)
                for (int64_t k = 0; k < w; ++k, ++e) {
                    const int64_t r = j - u + k;
                    const bool in = r >= 0 && r < M && col0[(size_t)j] >= 0;
                    rows[e] = in ? (int32_t)r : 0;
                    nzc[e] = in ? col0[(size_t)j] : -1;
                }
            p->nnz_local = slots;
            FD_TRY(try_window_plan(p, rows, nzc, padded, false));
            int shift = 0;
            if (p->window && colors_cyclic(col0, p->C, &shift)) finish_band_plan(p, w, u, 0, p->col0, p->col1, p->C, shift);
        }
    }
    FD_TRY(alloc_scratch(p, col0));
    store_caps_implicit_band(p, l, u);
    return FD_OK;
}

int fd_plan_create_dense(fd_ctx *ctx, int64_t M, int64_t N, int64_t ncols, const fd_plan_opts *opts, fd_plan **out)
{
    if (opts && (opts->flags & FD_PLAN_COMPLEX_X)) {   // complex-valued x: the same arm on (re, im) pairs
        fd_plan_opts o = *opts;
        FD_REQUIRE(o.fdtype != FD_COMPLEX, FD_ERR_UNSUPPORTED, "fdtype_error: Val(:complex) needs a real returntype (src/jacobians.jl:106)");
        FD_REQUIRE(ncols >= 0 && ncols <= N, FD_ERR_ARG, "ncols = maximum(colorvec) must be in 0..N");
        FD_REQUIRE((o.flags & ~kPlanKnownFlags) == 0, FD_ERR_ARG, "unknown fd_plan_opts.flags bits 0x%x", (unsigned)(o.flags & ~kPlanKnownFlags));
        o.flags &= ~(FD_PLAN_COMPLEX_X | FD_PLAN_FINGERPRINT);
        o.col_begin *= 2; o.col_end *= 2; o.x_begin *= 2; o.x_end *= 2;
        int rc;
        {
            LoweredScope lowered;
            rc = fd_plan_create_dense(ctx, 2 * M, 2 * N, ncols, &o, out);
        }
        return finish_fingerprint(rc, out, opts, 0, nullptr, 0, nullptr, 0, 8, 0, nullptr, 8, 0, FD_HOST, N);
    }
    FD_REQUIRE(ncols >= 0 && ncols <= (t_lowered_cx ? N / 2 : N), FD_ERR_ARG, "ncols = maximum(colorvec) must be in 0..N");
    if (opts && !(opts->col_begin == 0 && opts->col_end == 0) && !(opts->col_begin == 0 && opts->col_end == N)) {
        set_error("column windows are not supported for the dense arm");
        return FD_ERR_UNSUPPORTED;
    }
    int rc = new_plan(ctx, K_DENSE, M, N, out);
    if (rc) return rc;
    fd_plan *p = *out;
    FD_TRY(apply_opts(p, opts));
    // "colour" i == column i: identity colours drive the shared perturbation kernel
    std::vector<int32_t> col0((size_t)N);
    for (int64_t j = 0; j < N; ++j) col0[(size_t)j] = j < ncols ? (int32_t)j : -1;
    if (p->cx)   // lowered complex-valued x: column i perturbs re(x_i) = element 2i; imaginary parts are never perturbed
        for (int64_t j = 0; j < N; ++j) col0[(size_t)j] = ((j & 1) == 0 && j / 2 < ncols) ? (int32_t)(j / 2) : -1;
    p->C = ncols;
    p->color8 = false;
    FD_TRY(upload_colors(p, col0, {}));
    p->row0 = 0;
    p->row1 = M;
    FD_TRY(alloc_scratch(p, col0));
    p->nouts = 1;
    p->out_len[0] = M * ncols;
    return finish_fingerprint(FD_OK, out, opts, 0, nullptr, 0, nullptr, 0, 8, 0, nullptr, 8, 0, FD_HOST, N);   // (no arrays: always matches)
}

static int blockbanded_impl(fd_ctx *ctx, int64_t nblk, const void *blk_sizes, int64_t bl, int64_t bu,
                            const void *block_starts, const void *block_strides, int idx_bytes, int idx_base,
                            const void *colorvec, int color_bytes, const fd_plan_opts *opts, fd_plan **out)
{
    FD_REQUIRE(blk_sizes && block_starts && block_strides, FD_ERR_ARG, "NULL block layout array");
    FD_REQUIRE(idx_bytes == 4 || idx_bytes == 8, FD_ERR_ARG, "idx_bytes must be 4 or 8");
    FD_REQUIRE(nblk >= 1 && bl >= 0 && bu >= 0, FD_ERR_ARG, "bad block structure");
    std::vector<int64_t> off((size_t)nblk + 1, 0);
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t s = load_idx(blk_sizes, idx_bytes, b);
        FD_REQUIRE(s >= 0, FD_ERR_SHAPE, "negative block size");
        off[(size_t)b + 1] = off[(size_t)b] + s;
    }
    const int64_t N = off[(size_t)nblk];
    int rc = new_plan(ctx, K_COLRANGE, N, N, out);
    if (rc) return rc;
    fd_plan *p = *out;
    FD_TRY(apply_opts(p, opts));
    PbTimer tm(ctx->stream);
    std::vector<int32_t> col0;
    FD_TRY(ingest_colors(p, colorvec, color_bytes, col0));
    tm.mark("blockbanded: colours (host)");
    FD_TRY(upload_colors(p, col0, {}));
    tm.mark("blockbanded: colour upload");
    const int64_t nloc = p->col1 - p->col0;
    std::vector<int32_t> rlo((size_t)nloc), cnt((size_t)nloc);
    std::vector<int64_t> offs((size_t)nloc);
    const int64_t w = bl + bu + 1;
    int64_t r0 = N, r1 = 0, dmin = std::numeric_limits<int64_t>::max(), dmax = 0;
    bool pairs_ok = true;   // every column: even first row, even row count, even destination -> 16-B work items
    // (one pass over the block-columns that meet the local column range; the columns of a block-column share everything but their offset)
    for (int64_t J = 0; J < nblk; ++J) {
        const int64_t ja = std::max<int64_t>(off[(size_t)J], p->col0), jb = std::min<int64_t>(off[(size_t)J + 1], p->col1);
        if (ja >= jb) continue;
        const int64_t K0 = std::max<int64_t>(J - bu, 0), K1 = std::min<int64_t>(J + bl, nblk - 1);
        const int64_t stride = load_idx(block_strides, idx_bytes, J);
        const int64_t start0 = load_idx(block_starts, idx_bytes, (bu + K0 - J) + w * J) - idx_base;
        // the in-band blocks of a block-column must be stacked contiguously (the BlockSkyline layout)
        int64_t expect = start0;
        for (int64_t K = K0; K <= K1; ++K) {
            const int64_t st = load_idx(block_starts, idx_bytes, (bu + K - J) + w * J) - idx_base;
            if (st != expect) {
                set_error("block (%lld,%lld) is not stacked under its block-column (start %lld, expected %lld)",
                          (long long)K, (long long)J, (long long)st, (long long)expect);
                fd_plan_destroy(p);
                *out = nullptr;
                return FD_ERR_UNSUPPORTED;
            }
            expect += off[(size_t)K + 1] - off[(size_t)K];
        }
        const int64_t rows_n = off[(size_t)K1 + 1] - off[(size_t)K0];
        if (stride < rows_n) {
            set_error("block_strides[%lld] = %lld < rows in band %lld", (long long)J, (long long)stride, (long long)rows_n);
            fd_plan_destroy(p);
            *out = nullptr;
            return FD_ERR_SHAPE;
        }
        const int32_t lo32 = (int32_t)off[(size_t)K0], n32 = (int32_t)rows_n;
        int64_t o = start0 + (ja - off[(size_t)J]) * stride;
        for (int64_t j = ja; j < jb; ++j, o += stride) {
            const size_t jj = (size_t)(j - p->col0);
            rlo[jj] = lo32;
            cnt[jj] = n32;
            offs[jj] = o;
        }
        pairs_ok = pairs_ok && (((off[(size_t)K0] | rows_n) & 1) == 0);
        r0 = std::min<int64_t>(r0, off[(size_t)K0]);
        r1 = std::max<int64_t>(r1, off[(size_t)K1 + 1]);
        dmin = std::min<int64_t>(dmin, start0 + (ja - off[(size_t)J]) * stride);
        dmax = std::max<int64_t>(dmax, start0 + (jb - 1 - off[(size_t)J]) * stride + rows_n);
    }
    if (nloc == 0) { r0 = r1 = 0; dmin = dmax = 0; }
    // outputs are relative to the first local stored value
    for (auto &o : offs) {
        o -= dmin;
        pairs_ok = pairs_ok && ((o & 1) == 0);
    }
    p->cr_pairs = pairs_ok && nloc > 0;
    tm.mark("blockbanded: column ranges");
    {
        // the store capability (fd_colrange_store): colorvec must be a valid colouring -- the columns of the block-columns that touch
        // a block-row (K - bl .. K + bu) pairwise differ in colour, none without colour -- and the block structure is recorded
        bool valid = p->store_allowed && nloc > 0 && p->C >= 1;
        std::vector<int64_t> stamp((size_t)std::max<int64_t>(p->C, 1), 0);
        for (int64_t K = 0; K < nblk && valid; ++K)
            for (int64_t Jc = std::max<int64_t>(K - bl, 0); Jc <= std::min<int64_t>(K + bu, nblk - 1) && valid; ++Jc)
                for (int64_t j = off[(size_t)Jc]; j < off[(size_t)Jc + 1] && valid; ++j) {
                    const int32_t c = col0[(size_t)j];
                    if (c < 0 || stamp[(size_t)c] == K + 1) valid = false;
                    else stamp[(size_t)c] = K + 1;
                }
        p->store_cr_ok = valid;
        bool uniform = nblk > 0;
        for (int64_t b = 0; b < nblk; ++b) uniform = uniform && (off[(size_t)b + 1] - off[(size_t)b]) == (off[1] - off[0]);
        p->cr_nblk = nblk; p->cr_bs = uniform ? off[1] - off[0] : 0; p->cr_bl = (int)bl; p->cr_bu = (int)bu;
    }
    p->entry_begin = dmin;
    p->row0 = r0;
    p->row1 = r1;
    tm.mark("blockbanded: colouring check");
    FD_TRY(dev_upload(&p->d_cr_rlo, rlo));
    FD_TRY(dev_upload(&p->d_cr_cnt, cnt));
    FD_TRY(dev_upload(&p->d_cr_off, offs));
    tm.mark("blockbanded: uploads");
    FD_TRY(alloc_scratch(p, col0));
    tm.mark("blockbanded: scratch");
    p->nouts = 1;
    p->out_len[0] = dmax - dmin;
    return FD_OK;
}

// The remaining public constructors: the implementation above, then (FD_PLAN_FINGERPRINT) the fingerprints of the caller's arrays.
int fd_plan_create_csc_device(fd_ctx *ctx, int64_t M, int64_t N, const void *colptr_dev, const void *rowval_dev,
                              int idx_bytes, int idx_base, const void *colorvec_dev, int color_bytes,
                              const fd_plan_opts *opts, fd_plan **out)
{
    const int rc = csc_device_impl(ctx, M, N, colptr_dev, rowval_dev, idx_bytes, idx_base, colorvec_dev, color_bytes, opts, out);
    return finish_fingerprint(rc, out, opts, 1, colptr_dev, N + 1, rowval_dev, std::numeric_limits<int64_t>::max(), idx_bytes, idx_base,
                              colorvec_dev, color_bytes, N, FD_DEVICE, N);
}

int fd_plan_create_tridiagonal(fd_ctx *ctx, int64_t N, const void *colorvec, int color_bytes,
                               const fd_plan_opts *opts, fd_plan **out)
{
    int rc;
    if (opts && (opts->flags & FD_PLAN_COMPLEX_X)) {
        // J[j+1, j] -> dl[j], J[j, j] -> d[j], J[j-1, j] -> du[j-1]; the concatenated output is (dl | d | du) in complex elements
        FD_REQUIRE(N >= 1, FD_ERR_ARG, "N < 1");
        std::vector<int64_t> rows, cols, dest;
        rows.reserve((size_t)(3 * N)); cols.reserve((size_t)(3 * N)); dest.reserve((size_t)(3 * N));
        for (int64_t j = 0; j < N; ++j) {
            if (j > 0) { rows.push_back(j - 1); cols.push_back(j); dest.push_back((N - 1) + N + (j - 1)); }
            rows.push_back(j); cols.push_back(j); dest.push_back((N - 1) + j);
            if (j + 1 < N) { rows.push_back(j + 1); cols.push_back(j); dest.push_back(j); }
        }
        rc = lowered_structured(ctx, N, N, rows, cols, dest, 3 * N - 2, colorvec, color_bytes, opts, out);
        if (rc == FD_OK) {
            fd_plan *p = *out;
            p->split_n = 3;
            p->split_len[0] = 2 * (N - 1); p->split_len[1] = 2 * N; p->split_len[2] = 2 * (N - 1);
        }
    } else {
        rc = tridiagonal_impl(ctx, N, colorvec, color_bytes, opts, out);
    }
    return finish_fingerprint(rc, out, opts, 0, nullptr, 0, nullptr, 0, 8, 0, colorvec, color_bytes, N, FD_HOST, N);
}

int fd_plan_create_banded(fd_ctx *ctx, int64_t M, int64_t N, int64_t l, int64_t u, const void *colorvec,
                          int color_bytes, const fd_plan_opts *opts, fd_plan **out)
{
    int rc;
    if (opts && (opts->flags & FD_PLAN_COMPLEX_X)) {
        // data[(u + r - j) + (l + u + 1) j] = J[r, j]; the slots of rows outside the matrix are zero (the entry-list plan zero-fills)
        FD_REQUIRE(l + u + 1 >= 1 && l > -N && u > -M && M >= 1 && N >= 1, FD_ERR_ARG, "bad shape / bandwidths (%lld,%lld)", (long long)l, (long long)u);
        const int64_t w = l + u + 1;
        std::vector<int64_t> rows, cols, dest;
        for (int64_t j = 0; j < N; ++j)
            for (int64_t r = std::max<int64_t>(j - u, 0); r <= std::min<int64_t>(j + l, M - 1); ++r) {
                rows.push_back(r); cols.push_back(j); dest.push_back((u + r - j) + w * j);
            }
        rc = lowered_structured(ctx, M, N, rows, cols, dest, w * N, colorvec, color_bytes, opts, out);
    } else {
        rc = banded_impl(ctx, M, N, l, u, colorvec, color_bytes, opts, out);
    }
    return finish_fingerprint(rc, out, opts, 0, nullptr, 0, nullptr, 0, 8, 0, colorvec, color_bytes, N, FD_HOST, N);
}

// BandedBlockBandedMatrix J (ext/FiniteDiffBlockBandedMatricesExt.jl:16-42): a structural plan -- block sizes, block bandwidths
// (bl, bu), sub-block bandwidths (lam, mu), the start of every in-band block's banded-data slab and the slabs' column stride per
// block-column, as the reference reads them (pointer(bandeddata(view(J, K, J))), stride(data, 2)).  No entry list is built.
int fd_plan_create_bandedblockbanded(fd_ctx *ctx, int64_t nblk, const void *blk_sizes, int64_t bl, int64_t bu, int64_t lam, int64_t mu,
                                     const void *block_starts, const void *block_strides, int64_t data_len, int idx_bytes, int idx_base,
                                     const void *colorvec, int color_bytes, const fd_plan_opts *opts, fd_plan **out)
{
    FD_REQUIRE(blk_sizes && block_starts && block_strides && out, FD_ERR_ARG, "NULL block layout array");
    FD_REQUIRE(idx_bytes == 4 || idx_bytes == 8, FD_ERR_ARG, "idx_bytes must be 4 or 8");
    FD_REQUIRE(nblk >= 1 && bl >= 0 && bu >= 0 && lam >= 0 && mu >= 0 && bl + bu + 1 <= 1024 && lam + mu + 1 <= 1024, FD_ERR_ARG, "bad block structure");
    FD_REQUIRE(!(opts && (opts->flags & FD_PLAN_COMPLEX_X)), FD_ERR_UNSUPPORTED, "complex-valued x on BandedBlockBandedMatrix storage: use fd_plan_create_entries");
    std::vector<int32_t> off((size_t)nblk + 1, 0);
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t sz = load_idx(blk_sizes, idx_bytes, b);
        FD_REQUIRE(sz >= 0 && (int64_t)off[(size_t)b] + sz < ((int64_t)1 << 31), FD_ERR_SHAPE, "bad block size");
        off[(size_t)b + 1] = off[(size_t)b] + (int32_t)sz;
    }
    const int64_t N = off[(size_t)nblk];
    int rc = new_plan(ctx, K_BBB, N, N, out);
    if (rc) return rc;
    fd_plan *p = *out;
    FD_TRY(apply_opts(p, opts));
    if (!(p->col0 == 0 && p->col1 == N)) {
        set_error("column windows are not supported for BandedBlockBandedMatrix storage");
        fd_plan_destroy(p);
        *out = nullptr;
        return FD_ERR_UNSUPPORTED;
    }
    std::vector<int32_t> col0;
    FD_TRY(ingest_colors(p, colorvec, color_bytes, col0));
    FD_TRY(upload_colors(p, col0, {}));
    const int64_t w = bl + bu + 1, sw = lam + mu + 1;
    std::vector<int32_t> blk((size_t)N);
    std::vector<int64_t> start((size_t)(w * nblk), -1), stride((size_t)nblk, 0);
    int64_t out_len = 0, covered = 0;
    bool dense = true;      // every column owns (bl+bu+1)(lam+mu+1) consecutive slots: the slabs of blocks outside the matrix are reserved too
    for (int64_t J = 0; J < nblk; ++J) {
        const int64_t nJ = off[(size_t)J + 1] - off[(size_t)J];
        for (int64_t j = off[(size_t)J]; j < off[(size_t)J + 1]; ++j) blk[(size_t)j] = (int32_t)J;
        stride[(size_t)J] = load_idx(block_strides, idx_bytes, J);
        for (int64_t K = std::max<int64_t>(J - bu, 0); K <= std::min<int64_t>(J + bl, nblk - 1); ++K) {
            const int64_t st = load_idx(block_starts, idx_bytes, (bu + K - J) + w * J) - idx_base;
            if (!(st >= 0 && (nJ == 0 || stride[(size_t)J] >= sw))) {
                set_error("inconsistent slab of block (%lld,%lld): start %lld, column stride %lld < %lld", (long long)K, (long long)J,
                          (long long)st, (long long)stride[(size_t)J], (long long)sw);
                fd_plan_destroy(p);
                *out = nullptr;
                return FD_ERR_SHAPE;
            }
            start[(size_t)((bu + K - J) + w * J)] = st;
            if (nJ > 0) out_len = std::max<int64_t>(out_len, st + (nJ - 1) * stride[(size_t)J] + sw);
        }
        // BlockBandedMatrices' layout: slab d of block-column J at base_J + d (lam+mu+1), column stride >= (bl+bu+1)(lam+mu+1)
        int64_t base = -1;
        bool ok = stride[(size_t)J] >= w * sw;
        for (int64_t d = 0; d < w && ok; ++d) {
            const int64_t st = start[(size_t)(d + w * J)];
            if (st < 0) continue;
            if (base < 0) base = st - d * sw;
            ok = st - d * sw == base && base >= 0;
        }
        if (ok && base >= 0 && nJ > 0) {
            for (int64_t d = 0; d < w; ++d)
                if (start[(size_t)(d + w * J)] < 0) start[(size_t)(d + w * J)] = base + d * sw;      // (a reserved slab: zeros)
            out_len = std::max<int64_t>(out_len, base + (nJ - 1) * stride[(size_t)J] + w * sw);
            covered += nJ * w * sw;
        } else if (nJ > 0) {
            dense = false;
        }
    }
    if (data_len < out_len) {
        set_error("data_len %lld < the end of the last slab %lld", (long long)data_len, (long long)out_len);
        fd_plan_destroy(p);
        *out = nullptr;
        return FD_ERR_SHAPE;
    }
    p->bbb_fill = !(dense && covered == data_len);      // (slots no slab reaches: zero-filled before every launch, as fill!(J, 0) does)
    p->bbb_nb = nblk; p->bbb_bl = (int)bl; p->bbb_bu = (int)bu; p->bbb_lam = (int)lam; p->bbb_mu = (int)mu;
    p->row0 = 0;
    p->row1 = N;
    {
        // may a FD_LAZY_CAP_STORE launcher fill the slabs itself (fd_bbb_store)?  Uniform blocks, and colorvec a VALID colouring of
        // the BBB pattern: the columns that share a row -- blocks J in [K - bl, K + bu], in-block columns [k - lam, k + mu] -- differ in
        // colour.  Then a slot whose row does not depend on its column holds exactly 0 in the reference too.
        const int64_t bs0 = nblk > 0 ? (int64_t)off[1] - off[0] : 0;
        bool ok = bs0 > 0 && !p->cx;
        for (int64_t b = 0; b < nblk && ok; ++b) ok = (int64_t)off[(size_t)b + 1] - off[(size_t)b] == bs0;
        if (ok) {
            std::vector<int64_t> stamp((size_t)std::max<int64_t>(p->C, 1), -1);
            for (int64_t r = 0; r < N && ok; ++r) {
                const int64_t K = r / bs0, k = r - K * bs0;
                for (int64_t J = std::max<int64_t>(K - bl, 0); J <= std::min<int64_t>(K + bu, nblk - 1) && ok; ++J)
                    for (int64_t jj = std::max<int64_t>(k - lam, 0); jj <= std::min<int64_t>(k + mu, bs0 - 1); ++jj) {
                        const int32_t c = col0[(size_t)(J * bs0 + jj)];
                        if (c < 0) continue;
                        if (stamp[(size_t)c] == r) { ok = false; break; }
                        stamp[(size_t)c] = r;
                    }
            }
        }
        p->store_bbb_ok = ok;
        p->bbb_bs = ok ? bs0 : 0;
    }
    FD_TRY(dev_upload(&p->d_bbb_off, off));
    FD_TRY(dev_upload(&p->d_bbb_blk, blk));
    FD_TRY(dev_upload(&p->d_bbb_start, start));
    FD_TRY(dev_upload(&p->d_bbb_stride, stride));
    FD_TRY(alloc_scratch(p, col0));
    p->nouts = 1;
    p->out_len[0] = data_len;
    return finish_fingerprint(FD_OK, out, opts, 0, nullptr, 0, nullptr, 0, 8, 0, colorvec, color_bytes, N, FD_HOST, N);
}

int fd_plan_create_blockbanded(fd_ctx *ctx, int64_t nblk, const void *blk_sizes, int64_t bl, int64_t bu,
                               const void *block_starts, const void *block_strides, int idx_bytes, int idx_base,
                               const void *colorvec, int color_bytes, const fd_plan_opts *opts, fd_plan **out)
{
    int rc;
    if (opts && (opts->flags & FD_PLAN_COMPLEX_X)) {
        // every in-band block (K, J) is dense: row t of the block, local column c -> data[block_starts(K, J) + t + block_strides[J] c]
        FD_REQUIRE(blk_sizes && block_starts && block_strides, FD_ERR_ARG, "NULL block layout array");
        FD_REQUIRE(idx_bytes == 4 || idx_bytes == 8, FD_ERR_ARG, "idx_bytes must be 4 or 8");
        FD_REQUIRE(nblk >= 1 && bl >= 0 && bu >= 0, FD_ERR_ARG, "bad block structure");
        std::vector<int64_t> off((size_t)nblk + 1, 0);
        for (int64_t b = 0; b < nblk; ++b) {
            const int64_t sz = load_idx(blk_sizes, idx_bytes, b);
            FD_REQUIRE(sz >= 0, FD_ERR_SHAPE, "negative block size");
            off[(size_t)b + 1] = off[(size_t)b] + sz;
        }
        const int64_t Nn = off[(size_t)nblk], w = bl + bu + 1;
        std::vector<int64_t> rows, cols, dest;
        int64_t out_len = 0;
        for (int64_t J = 0; J < nblk; ++J) {
            const int64_t stride = load_idx(block_strides, idx_bytes, J);
            for (int64_t K = std::max<int64_t>(J - bu, 0); K <= std::min<int64_t>(J + bl, nblk - 1); ++K) {
                const int64_t st = load_idx(block_starts, idx_bytes, (bu + K - J) + w * J) - idx_base;
                FD_REQUIRE(st >= 0 && stride >= off[(size_t)K + 1] - off[(size_t)K], FD_ERR_SHAPE, "inconsistent block layout at block (%lld,%lld)", (long long)K, (long long)J);
                for (int64_t c = 0; c < off[(size_t)J + 1] - off[(size_t)J]; ++c)
                    for (int64_t t = 0; t < off[(size_t)K + 1] - off[(size_t)K]; ++t) {
                        rows.push_back(off[(size_t)K] + t); cols.push_back(off[(size_t)J] + c); dest.push_back(st + t + stride * c);
                        out_len = std::max<int64_t>(out_len, st + t + stride * c + 1);
                    }
            }
        }
        rc = lowered_structured(ctx, Nn, Nn, rows, cols, dest, out_len, colorvec, color_bytes, opts, out);
    } else {
        rc = blockbanded_impl(ctx, nblk, blk_sizes, bl, bu, block_starts, block_strides, idx_bytes, idx_base, colorvec, color_bytes, opts, out);
    }
    const int64_t N = (rc == FD_OK && out && *out) ? ((*out)->cx ? (*out)->N / 2 : (*out)->N) : 0;
    return finish_fingerprint(rc, out, opts, 0, nullptr, 0, nullptr, 0, 8, 0, colorvec, color_bytes, N, FD_HOST, N);
}

