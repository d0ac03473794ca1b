// libfdjac C ABI (include/fdjac.h): contexts, plan construction, and the orchestration of one
// coloured Jacobian evaluation -- the device-side body of the reference's cached in-place
// finite_difference_jacobian! (src/jacobians.jl:504-653).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <new>

#include <atomic>
#include <thread>
#include "fdjac_internal.h"

namespace fdjac {

#ifndef FDJAC_F32
static thread_local char g_err[512] = "";
#endif

int launch_eps(fd_plan *p, const real_t *x, double relstep, double absstep, double dir);
int launch_eps_partial(fd_plan *p, const real_t *x, int b0, int nb);
int launch_eps_finalize(fd_plan *p, int nparts, int ldp, double relstep, double absstep, double dir);
constexpr int kMaxEpsShards = 1024;   // the partial buffer has room for this many padded shards
int launch_eps_perturb_small(fd_plan *p, const real_t *x, double relstep, double absstep, double dir, int pmode,
                             int base_row);
int launch_perturb(fd_plan *p, const real_t *x, int c_lo, int B);
int launch_decompress(fd_plan *p, const real_t *fx, int c_lo, int c_hi, real_t *const *outs, int mode);
int launch_fill(fd_ctx *ctx, real_t *ptr, int64_t n, real_t v);
int launch_scale(fd_ctx *ctx, real_t *dst, const real_t *src, int64_t n, real_t factor);
int launch_stream_copy(fd_ctx *ctx, const void *src, void *dst, int64_t n16);
int balanced_grid(int64_t tiles, int64_t cap);

static inline int64_t load_idx(const void *p, int bytes, int64_t i)
{
    return bytes == 8 ? ((const int64_t *)p)[i] : (int64_t)((const int32_t *)p)[i];
}

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

template <typename T> static int dev_upload(T **dst, const std::vector<T> &src)
{
    *dst = nullptr;
    if (src.empty()) return FD_OK;
    FD_HIP_CHECK(hipMalloc((void **)dst, sizeof(T) * src.size()));
    FD_HIP_CHECK(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
    return FD_OK;
}

template <typename T> static int dev_alloc(T **dst, int64_t nelem)
{
    *dst = nullptr;
    if (nelem <= 0) nelem = 1;
    hipError_t e = hipMalloc((void **)dst, sizeof(T) * (size_t)nelem);
    if (e == hipErrorOutOfMemory) {
        set_error("hipMalloc of %lld bytes failed: out of memory", (long long)(nelem * (int64_t)sizeof(T)));
        return FD_ERR_NOMEM;
    }
    FD_HIP_CHECK(e);
    return FD_OK;
}

// Colours: validate, find C = maximum(colorvec), convert to 0-based with "none" for < 1.
static int ingest_colors(fd_plan *p, const void *colorvec, int color_bytes, std::vector<int32_t> &col0)
{
    FD_REQUIRE(colorvec != nullptr, FD_ERR_ARG, "colorvec is NULL");
    FD_REQUIRE(color_bytes == 4 || color_bytes == 8, FD_ERR_ARG, "color_bytes must be 4 or 8");
    const int64_t N = p->N;
    col0.resize((size_t)N);
    int64_t C = 0;
    for (int64_t j = 0; j < N; ++j) {
        const int64_t c = load_idx(colorvec, color_bytes, j);
        if (c > C) C = c;
        FD_REQUIRE(c <= std::numeric_limits<int32_t>::max(), FD_ERR_ARG, "colour %lld too large", (long long)c);
        col0[(size_t)j] = c >= 1 ? (int32_t)(c - 1) : -1;
    }
    p->C = C;
    p->color8 = C <= 253;  // 0xFF = no colour, 0xFE = padding
    return FD_OK;
}

static int upload_colors(fd_plan *p, const std::vector<int32_t> &col0, const std::vector<int32_t> &nzc)
{
    if (p->color8) {
        std::vector<uint8_t> a(col0.size()), b(nzc.size());
        for (size_t i = 0; i < col0.size(); ++i) a[i] = col0[i] < 0 ? 0xFF : (uint8_t)col0[i];
        for (size_t i = 0; i < nzc.size(); ++i) b[i] = nzc[i] == -2 ? 0xFE : nzc[i] < 0 ? 0xFF : (uint8_t)nzc[i];
        uint8_t *d = nullptr;
        int rc = dev_upload(&d, a);
        if (rc) return rc;
        p->d_color = d;
        rc = dev_upload(&d, b);
        if (rc) return rc;
        p->d_nzcolor = d;
    } else {
        int32_t *d = nullptr;
        int rc = dev_upload(&d, col0);
        if (rc) return rc;
        p->d_color = d;
        rc = dev_upload(&d, nzc);
        if (rc) return rc;
        p->d_nzcolor = d;
    }
    return FD_OK;
}

// "The plan being created IS the lowered real problem of FD_PLAN_COMPLEX_X": set by the lowering functions around their inner
// plan_create call (LoweredScope), read by apply_opts.  Internal state, deliberately NOT a bit of the public fd_plan_opts.flags.
static thread_local bool t_lowered_cx = false;
struct LoweredScope {
    LoweredScope() { t_lowered_cx = true; }
    ~LoweredScope() { t_lowered_cx = false; }
};
constexpr int32_t kPlanKnownFlags = FD_PLAN_EPS_CONTIGUOUS | FD_PLAN_COMPLEX_X | FD_PLAN_FINGERPRINT | FD_PLAN_STORE_CSC;

static int apply_opts(fd_plan *p, const fd_plan_opts *opts)
{
    FD_REQUIRE(opts != nullptr, FD_ERR_ARG, "opts is NULL");
    FD_REQUIRE(opts->fdtype == FD_FORWARD || opts->fdtype == FD_CENTRAL || opts->fdtype == FD_COMPLEX,
               FD_ERR_UNSUPPORTED,
               "Unrecognized fdtype: valid values are Val{:forward}, Val{:central} and Val{:complex}.");
    FD_REQUIRE((opts->flags & ~kPlanKnownFlags) == 0, FD_ERR_ARG, "unknown fd_plan_opts.flags bits 0x%x (zero the struct before filling it)",
               (unsigned)(opts->flags & ~kPlanKnownFlags));
    FD_REQUIRE(!(opts->flags & FD_PLAN_COMPLEX_X), FD_ERR_UNSUPPORTED,
               "complex-valued x (FD_PLAN_COMPLEX_X) reaches this plan kind through its lowering only");
    p->fdtype = opts->fdtype;
    p->col0 = 0;
    p->col1 = p->N;
    if (!(opts->col_begin == 0 && opts->col_end == 0)) {
        FD_REQUIRE(opts->col_begin >= 0 && opts->col_begin <= opts->col_end && opts->col_end <= p->N, FD_ERR_ARG,
                   "column window [%lld,%lld) outside [0,%lld)", (long long)opts->col_begin,
                   (long long)opts->col_end, (long long)p->N);
        p->col0 = opts->col_begin;
        p->col1 = opts->col_end;
    }
    p->x0 = 0;
    p->x1 = p->N;
    if (!(opts->x_begin == 0 && opts->x_end == 0)) {
        FD_REQUIRE(opts->x_begin >= 0 && opts->x_begin <= opts->x_end && opts->x_end <= p->N, FD_ERR_ARG,
                   "x window [%lld,%lld) outside [0,%lld)", (long long)opts->x_begin, (long long)opts->x_end,
                   (long long)p->N);
        p->x0 = opts->x_begin;
        p->x1 = opts->x_end;
    }
    p->scratch_bytes = opts->scratch_bytes > 0 ? opts->scratch_bytes : ((int64_t)64 << 30);
    // test / tuning switches select between bit-identical kernel variants; they are read HERE, once per plan, so a
    // process can build plans of both variants side by side and fd_plan_info reports which one a plan uses
    auto env_int = [](const char *name, int dflt) { const char *v = getenv(name); return (v && *v) ? atoi(v) : dflt; };
    p->small_ok = env_int("FDJAC_SMALL", 1) != 0;
    p->list_U = env_int("FDJAC_TILE", 2);
    if (p->list_U != 1 && p->list_U != 2) p->list_U = 4;
    // non-temporal loads of x in the step-size reduction: right when 240 MB of plain nzval stores are still draining (the
    // hand-over path, round 2); WRONG when f!'s storing launch follows (round 3): that launch re-reads x, which the reduction's
    // plain loads leave in the 256 MiB Infinity Cache -- N = 10^7: 75 instead of 82 us per Jacobian (profiles/r03_c_*).  Unless
    // FDJAC_EPS_NT forces one, a call decides by which path it takes.
    // the reduction's blocks sum contiguous ranges of x: the default since round 3 (measured equal to the grid-stride map -- N = 10^7:
    // 75.3 vs 76.0 us per Jacobian, profiles/r03_f_eps_contig_ab.txt -- and a sharded reduction then reads only the shard's own
    // range); FDJAC_EPS_CONTIG=0 restores the grid-stride map, FD_PLAN_EPS_CONTIGUOUS insists on the contiguous one
    p->eps_contig = (opts->flags & FD_PLAN_EPS_CONTIGUOUS) != 0 || env_int("FDJAC_EPS_CONTIG", 1) != 0;
    p->cx = t_lowered_cx;                                // (inside a LoweredScope only)
    if (p->cx) p->small_ok = false;                      // the fused small-problem launch has ONE colour rule for norm and perturbation
    p->eps_nt_forced = env_int("FDJAC_EPS_NT", -1);
    p->eps_nt = p->eps_nt_forced != 0;
    p->lazy_diff = env_int("FDJAC_LAZY_DIFF", 1) != 0;
    p->bd_allowed = env_int("FDJAC_BAND_DESC", 1) != 0;
    // a FD_LAZY_CAP_STORE launcher stores the Jacobian of a verified exact band itself (include/fdjac_device.h): default since
    // round 3 (N = 10^7 tridiagonal: 0.18 -> 0.09 ms per Jacobian, bit-identical); FDJAC_LAZY_STORE=0 keeps the hand-over
    p->store_allowed = env_int("FDJAC_LAZY_STORE", 1) != 0;
    p->want_store_csc = (opts->flags & FD_PLAN_STORE_CSC) != 0;   // a compact device copy of the pattern for column-centric storing launches
    p->own_c0 = 0;
    p->own_c1 = -1;
    if (!(opts->color_begin == 0 && opts->color_end == 0)) {
        FD_REQUIRE(opts->color_begin >= 0 && opts->color_begin <= opts->color_end, FD_ERR_ARG,
                   "colour range [%lld,%lld) is not a range", (long long)opts->color_begin, (long long)opts->color_end);
        p->own_c0 = opts->color_begin;
        p->own_c1 = opts->color_end;
    }
    return FD_OK;
}

// FD_PLAN_FINGERPRINT: after a successful build, record the content fingerprints of the CALLER's arrays (fdjac_match.hip) -- at
// the public entry points, so that a lowered (complex-valued x) plan is fingerprinted in the caller's units, not the lowered ones.
static int finish_fingerprint(int rc, fd_plan **out, const fd_plan_opts *opts, int idx_kind, const void *a, int64_t len_a, const void *b,
                              int64_t len_b, int idx_bytes, int idx_base, const void *colorvec, int color_bytes, int64_t len_color,
                              int memkind, int64_t N)
{
    if (rc || !opts || !(opts->flags & FD_PLAN_FINGERPRINT) || !out || !*out) return rc;
    fd_pattern_arrays src;
    memset(&src, 0, sizeof src);
    src.idx_a = a; src.len_a = len_a; src.idx_b = b; src.len_b = len_b; src.colorvec = colorvec; src.len_color = len_color;
    src.idx_bytes = idx_bytes ? idx_bytes : 8; src.idx_base = idx_base; src.color_bytes = color_bytes; src.memkind = memkind;
    const bool all = opts->col_begin == 0 && opts->col_end == 0;
    rc = plan_record_fingerprint(*out, idx_kind, &src, all ? 0 : opts->col_begin, all ? N : opts->col_end);
    if (rc) {
        fd_plan_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

// Scratch for the batched perturbed points / f! values and the epsilon reduction.
static int alloc_scratch(fd_plan *p, const std::vector<int32_t> &col0)
{
    p->cplx = p->fdtype == FD_COMPLEX ? 2 : 1;
    p->pts = p->fdtype == FD_CENTRAL ? 2 : 1;
    p->ldx = round_up(p->N, 32);
    p->ldf = round_up(p->M, 32);
    const int64_t per_color = (int64_t)p->pts * p->cplx * (int64_t)sizeof(real_t) * (p->ldx + p->ldf);
    int64_t B = p->C > 0 ? p->scratch_bytes / std::max<int64_t>(per_color, 1) : 1;
    B = std::max<int64_t>(1, std::min<int64_t>(B, std::max<int64_t>(p->C, 1)));
    B = std::min<int64_t>(B, 32768);  // keeps the f! batch within one grid dimension
    p->chunkB = B;
    p->nchunks = p->C > 0 ? (p->C + B - 1) / B : 0;
    int rc;
    // (+1 row: small problems evaluate f(x) as one more member of the perturbed batch)
    // (the materialised points d_X and the staging copies of x / f_in are allocated on first use -- ensure_points /
    //  ensure_stage: a lazy-point launcher with device inputs never needs them, 480 MB less at N = 10^7)
    if ((rc = dev_alloc(&p->d_FX, (B * p->pts + 1) * p->cplx * p->ldf))) return rc;
    if ((rc = dev_alloc(&p->d_fx, p->ldf))) return rc;
    if ((rc = dev_alloc(&p->d_eps, std::max<int64_t>(p->C, 1)))) return rc;

    if (p->fdtype == FD_COMPLEX) {
        // d_fx is not used by the complex step: keep it zero, it is the "fx" of the imag-only decompression
        FD_HIP_CHECK(hipMemset(p->d_fx, 0, sizeof(real_t) * (size_t)p->ldf));
        p->d_zero = p->d_fx;
        // eps(Float64) for every colour (src/epsilons.jl:104-107, src/jacobians.jl:624)
        FD_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
        int r2 = launch_fill(p->ctx, p->d_eps, std::max<int64_t>(p->C, 1), std::numeric_limits<real_t>::epsilon());
        if (r2) return r2;
    } else if (p->C > 0 && p->kind != K_DENSE) {
        if (p->C <= kRegColors) {
            // cyclic colours (mod1(j, C) and its rotations): the reduction computes them instead of reading them
            if (!p->built_on_device) {   // (the device builder ran the same test with wave ballots)
                const char *fc = getenv("FDJAC_EPS_CYCLIC");
                bool cyc = !(fc && *fc && atoi(fc) == 0) && p->N >= 1 && col0[0] >= 0;
                const int32_t sh = cyc ? col0[0] : 0;
                for (int64_t j = 0; j < p->N && cyc; ++j) cyc = col0[(size_t)j] == (int32_t)((j + sh) % p->C);
                p->cyc_C = cyc ? (int)p->C : 0;
                p->cyc_shift = cyc ? (int)sh : 0;
            }
            const char *cm = getenv("FDJAC_GRID_CAP");
            // 4 workgroups per CU: measured 31.0 us for partial + finalize at N = 10^7 (8: 34.5, 16: 33.7, 2: 36.6,
            // uncapped 36.1 -- fewer partials for the finalize, enough loads in flight for the reduction)
            const int64_t mult = (cm && *cm) ? atoll(cm) : 4;
            const int64_t tiles = (p->N + 2047) / 2048;  // k_eps_partial_reg: 4 x 512 elements per block round
            // (a function of N alone -- 256 CUs x 4, not of the device the plan happens to live on: every rank of a sharded
            //  reduction must cut the same blocks; fd_plan_set_comm verifies it)
            p->n_partial_blocks = balanced_grid(tiles, mult > 0 ? (int64_t)256 * mult : ((int64_t)1 << 30));
            p->eps_tpb = p->eps_contig ? (int)((tiles + p->n_partial_blocks - 1) / p->n_partial_blocks) : 0;
            // (+ kMaxEpsShards rows: a sharded reduction pads the grid to a whole number of blocks per shard)
            p->partial_cap = ((int64_t)p->n_partial_blocks + kMaxEpsShards) * kRegColors;
            if ((rc = dev_alloc(&p->d_partial, p->partial_cap))) return rc;
        } else {
            // counting sort of the columns by colour
            std::vector<int64_t> cptr((size_t)p->C + 1, 0);
            for (int64_t j = 0; j < p->N; ++j)
                if (col0[(size_t)j] >= 0) cptr[(size_t)col0[(size_t)j] + 1]++;
            for (int64_t c = 0; c < p->C; ++c) cptr[(size_t)c + 1] += cptr[(size_t)c];
            std::vector<int32_t> perm((size_t)cptr[(size_t)p->C]);
            std::vector<int64_t> fillp(cptr.begin(), cptr.end() - 1);
            for (int64_t j = 0; j < p->N; ++j)
                if (col0[(size_t)j] >= 0) perm[(size_t)fillp[(size_t)col0[(size_t)j]]++] = (int32_t)j;
            if ((rc = dev_upload(&p->d_perm, perm))) return rc;
            if ((rc = dev_upload(&p->d_cptr, cptr))) return rc;
            // ~512 columns per workgroup: two dependent-free rounds of (index load, gather) each -- with 4096 the
            // reduction of a many-coloured small problem ran at the latency of 16 rounds (96 colours, N = 320000: 21 us)
            int64_t chunks = (p->N / std::max<int64_t>(p->C, 1) + 511) / 512;
            p->seg_chunks = (int)std::max<int64_t>(1, std::min<int64_t>(chunks, 256));
            FD_REQUIRE((int64_t)p->C * p->seg_chunks < ((int64_t)1 << 31), FD_ERR_UNSUPPORTED, "too many colours");
            if ((rc = dev_alloc(&p->d_partial, (int64_t)p->seg_chunks * p->C))) return rc;
        }
    }
    return FD_OK;
}

static int ensure_points(fd_plan *p)
{
    if (p->d_X) return FD_OK;
    return dev_alloc(&p->d_X, (p->chunkB * p->pts + 1) * p->cplx * p->ldx);
}
static int ensure_stage(fd_plan *p, bool x, bool fin)
{
    int rc;
    if (x && !p->d_xstage && (rc = dev_alloc(&p->d_xstage, p->ldx))) return rc;
    if (fin && !p->d_finstage && (rc = dev_alloc(&p->d_finstage, p->ldf))) return rc;
    return FD_OK;
}

// the all-zero "fx" and the doubled step sizes of decompressions that receive differences (FD_LAZY_CAP_DIFF)
static int ensure_diff_scratch(fd_plan *p)
{
    int rc;
    if (!p->d_zero_own) {
        if ((rc = dev_alloc(&p->d_zero_own, p->ldf))) return rc;
        FD_HIP_CHECK(hipMemsetAsync(p->d_zero_own, 0, sizeof(real_t) * (size_t)p->ldf, p->ctx->stream));
        p->d_zero = p->d_zero_own;
    }
    if (p->fdtype == FD_CENTRAL && !p->d_eps2 && (rc = dev_alloc(&p->d_eps2, std::max<int64_t>(p->C, 1)))) return rc;
    return FD_OK;
}

static int new_plan(fd_ctx *ctx, int kind, int64_t M, int64_t N, fd_plan **out)
{
    FD_REQUIRE(ctx != nullptr && out != nullptr, FD_ERR_ARG, "ctx/out is NULL");
    FD_REQUIRE(M >= 0 && N >= 1, FD_ERR_SHAPE, "bad shape %lld x %lld", (long long)M, (long long)N);
    FD_REQUIRE(M < ((int64_t)1 << 31) && N < ((int64_t)1 << 31), FD_ERR_UNSUPPORTED,
               "dimensions >= 2^31 are not supported");
    FD_HIP_CHECK(hipSetDevice(ctx->device));
    fd_plan *p = new (std::nothrow) fd_plan();
    FD_REQUIRE(p != nullptr, FD_ERR_NOMEM, "out of host memory");
    p->ctx = ctx;
    p->kind = kind;
    p->M = M;
    p->N = N;
    *out = p;
    return FD_OK;
}

#define FD_TRY(expr)                 \
    do {                             \
        int _rc = (expr);            \
        if (_rc != FD_OK) {          \
            fd_plan_destroy(p);      \
            *out = nullptr;          \
            return _rc;              \
        }                            \
    } while (0)

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
        const char *fw = getenv("FDJAC_WINDOW"), *fs = getenv("FDJAC_SORTED");
        const int force_w = (fw && *fw) ? atoi(fw) : -1, force_s = (fs && *fs) ? atoi(fs) : -1;
        WinBuild best;
        if (force_w != 0 && force_s != 1) {
            const char *ft = getenv("FDJAC_WIN_TILE");   // test / tuning switch: force the tile size (2048, 1024 or 512)
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
                const char *fp = getenv("FDJAC_WIN_PERIODIC");
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
    const char *fw = getenv("FDJAC_WINDOW2D");
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

// Shared by the three index-list kinds: local entries [e0,e1) with rows, columns (0-based).
static int finish_list_plan(fd_plan *p, const std::vector<int32_t> &col0, std::vector<int32_t> &rows,
                            std::vector<int32_t> &nzc, std::vector<int64_t> &dest,
                            const std::vector<int64_t> *colstart = nullptr)
{
    int rc;
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

    if (!has_dest && p->nnz_local > 0) {
        const char *fw1 = getenv("FDJAC_WINDOW"), *fs1 = getenv("FDJAC_SORTED");
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
            const char *fs = getenv("FDJAC_SORTED");
            p->sorted_gather = scattered;
            if (fs && *fs) p->sorted_gather = atoi(fs) != 0 && p->nnz_local >= 4 * kSortTile;
        }
    }
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
        if ((rc = dev_upload(&p->d_spos, spos))) return rc;
        // f(x) through LDS (forward differences, k_decompress_sorted FXL): the runs of rows every tile touches
        {
            const char *fl = getenv("FDJAC_FX_LDS");
            if (p->fdtype == FD_FORWARD && !(fl && *fl && atoi(fl) == 0)) {
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
        // Tile ORDER for patterns with a far band (3-D stencils: offsets 0, +-1, +-nx, +-nx*ny).  A tile's gathers reach the rows a
        // whole "plane" D = max |row - column| away; walking the tiles in storage order the three planes in use are 3 * C * D * 8
        // bytes (6.7 MB for 200^3, 7 colours) against 4 MB of L2 per XCD, and the plane above / below is fetched through the
        // fabric a second and third time (rocprofv3: 2.2 x the distinct bytes).  Walk instead: for each in-plane region of Rg
        // columns, all planes in turn -- the three-plane working set of a region is 3 * C * Rg * 8 bytes <= 2 MiB.  Pure
        // scheduling: which workgroup takes which tile (results do not depend on it).
        const char *to = getenv("FDJAC_TILE_ORDER");
        const int want = (to && *to) ? atoi(to) : 1;
        if (colstart && want != 0 && ntiles >= (want == 2 ? 16u : 256u)) {
            const int64_t ncols = (int64_t)colstart->size() - 1;
            const int64_t D = reach;
            const int64_t cpt = std::max<int64_t>(1, ncols / (int64_t)ntiles);
            if (D >= (want == 2 ? 2 : 16) * cpt && D < ncols) {
                int64_t Rg = ((int64_t)2 << 20) / (3 * std::max<int64_t>(p->C, 1) * (int64_t)sizeof(real_t));
                Rg = std::max<int64_t>(4 * cpt, std::min<int64_t>(Rg, D / 2));
                // only the tiles the kernel walks: ceil(nnz_local / kSortTile) -- the lists are padded to kListPad (two tiles), and
                // an all-padding tile in the order would displace a real one (its values would never be written)
                const size_t nreal = (size_t)((p->nnz_local + kSortTile - 1) / kSortTile);
                std::vector<int32_t> order(nreal);
                for (size_t t = 0; t < nreal; ++t) order[t] = (int32_t)t;
                std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
                    const int64_t ua = tcol[(size_t)a] % D, ub = tcol[(size_t)b] % D;
                    const int64_t ka = ua / Rg, kb = ub / Rg;
                    if (ka != kb) return ka < kb;
                    const int64_t la = tcol[(size_t)a] / D, lb = tcol[(size_t)b] / D;
                    if (la != lb) return la < lb;
                    return ua < ub;
                });
                if ((rc = dev_upload(&p->d_tile_order, order))) return rc;
            }
        }
    }
    if ((rc = dev_upload(&p->d_rowval, rows))) return rc;
    if ((rc = upload_colors(p, col0, nzc))) return rc;
    if (has_dest && (rc = dev_upload(&p->d_dest, dest))) return rc;
    return alloc_scratch(p, col0);
}

}  // namespace fdjac

#include "fdjac_planbuild.hip"
#include "fdjac_storetable.hip"

using namespace fdjac;

extern "C" {

#ifndef FDJAC_F32   /* shared by both instantiations: defined once, by the Float64 build */
void fdjac_set_error_v(const char *fmt, va_list ap) { vsnprintf(g_err, sizeof(g_err), fmt, ap); }

int fd_version(void) { return FDJAC_VERSION; }
const char *fd_last_error(void) { return g_err; }

int fd_ctx_create(int device, void *stream, fd_ctx **out)
{
    FD_REQUIRE(out != nullptr, FD_ERR_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return FD_ERR_NODEVICE;
    }
    FD_REQUIRE(device >= 0 && device < ndev, FD_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    FD_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    FD_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    fd_ctx *c = new (std::nothrow) fd_ctx();
    FD_REQUIRE(c != nullptr, FD_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (stream == FD_STREAM_DEFAULT) {
        c->stream = nullptr;   // the legacy default stream
        c->own_stream = false;
    } else if (stream) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            delete c;
            set_error("hipStreamCreate failed: %s", hipGetErrorString(se));
            return FD_ERR_HIP;
        }
        c->own_stream = true;
    }
    *out = c;
    return FD_OK;
}

int fd_ctx_destroy(fd_ctx *ctx)
{
    if (!ctx) return FD_OK;
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return FD_OK;
}

void *fd_ctx_stream(fd_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int fd_ctx_synchronize(fd_ctx *ctx)
{
    FD_REQUIRE(ctx != nullptr, FD_ERR_ARG, "ctx is NULL");
    FD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return FD_OK;
}

#endif

int fd_plan_destroy(fd_plan *p)
{
    if (!p) return FD_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    void *ptrs[] = {p->d_color, p->d_rowval, p->d_nzcolor, p->d_dest, p->d_spos, p->d_wtiles, p->d_wcode, p->d_w2desc, p->d_cr_rlo, p->d_cr_cnt, p->d_cr_off,
                    p->d_perm, p->d_cptr, p->d_X, p->d_FX, p->d_fx, p->d_eps, p->d_partial, p->d_xstage,
                    p->d_finstage, p->d_outstage[0], p->d_outstage[1], p->d_outstage[2], p->d_zero_own, p->d_eps2, p->d_tile_order, p->d_fxwin, p->d_fp, p->d_sc_colptr, p->d_sc_rowval, p->d_split};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    for (auto &sp : p->spans) {
        p->event_pool.push_back(sp.a);
        p->event_pool.push_back(sp.b);
    }
    for (hipEvent_t ev : p->event_pool) (void)hipEventDestroy(ev);
    delete p;
    return FD_OK;
}

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
        const char *pd = getenv("FDJAC_PLAN_DEVICE");
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
            if (res == PBR_DONE && brc == FD_OK)   // (the raw arrays are still on the device: the compact copy comes from them)
                brc = build_store_csc(p, (const char *)d_cp - ib * (size_t)p->col0, (const char *)d_rv - ib * (size_t)e0, idx_bytes, idx_base);
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
    FD_TRY(finish_list_plan(p, col0, rows, nzc, dest, kind == K_CSC ? &colstart : nullptr));
    if (kind == K_CSC) FD_TRY(build_store_csc_host(p, colptr, rowval, idx_bytes, idx_base));
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
        hipError_t e = idx_bytes == 8 ? hipMemcpy(&v64, (const char *)colptr_dev + 8 * (size_t)j, 8, hipMemcpyDeviceToHost)
                                      : hipMemcpy(&v32, (const char *)colptr_dev + 4 * (size_t)j, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_error("reading colptr from the device failed: %s", hipGetErrorString(e)); fd_plan_destroy(p); *out = nullptr; return FD_ERR_HIP; }
        cp[k] = (idx_bytes == 8 ? v64 : (int64_t)v32) - idx_base;
    }
    if (!(cp[0] >= 0 && cp[1] >= cp[0])) { set_error("colptr is not monotone"); fd_plan_destroy(p); *out = nullptr; return FD_ERR_SHAPE; }
    p->entry_begin = cp[0];
    int brc = FD_OK;
    const char *pd = getenv("FDJAC_PLAN_DEVICE");
    int res = (pd && *pd && atoi(pd) == 0) ? (int)PBR_DECLINED
                                           : device_build_csc(p, colptr_dev, rowval_dev, idx_bytes, idx_base, colorvec_dev, color_bytes, cp[0], cp[1], &brc);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipGetLastError();
    if (res == PBR_DONE) {
        if (brc == FD_OK) brc = build_store_csc(p, colptr_dev, rowval_dev, idx_bytes, idx_base);
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
    auto mix_dev = [&](const void *d, size_t n) -> int {
        if (!d || !n) return FD_OK;
        std::vector<char> tmp(n);
        FD_HIP_CHECK(hipMemcpy(tmp.data(), d, n, hipMemcpyDeviceToHost));
        mix(tmp.data(), n);
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
        const char *pd = getenv("FDJAC_PLAN_DEVICE");
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
        const char *fw = getenv("FDJAC_WINDOW");
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
        const char *pd = getenv("FDJAC_PLAN_DEVICE");
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
    std::vector<int32_t> col0;
    FD_TRY(ingest_colors(p, colorvec, color_bytes, col0));
    FD_TRY(upload_colors(p, col0, {}));
    const int64_t nloc = p->col1 - p->col0;
    std::vector<int32_t> rlo((size_t)nloc), cnt((size_t)nloc);
    std::vector<int64_t> offs((size_t)nloc);
    const int64_t w = bl + bu + 1;
    int64_t r0 = N, r1 = 0, dmin = std::numeric_limits<int64_t>::max(), dmax = 0;
    int64_t J = 0;
    bool pairs_ok = true;   // every column: even first row, even row count, even destination -> 16-B work items
    for (int64_t j = p->col0; j < p->col1; ++j) {
        while (off[(size_t)J + 1] <= j) ++J;
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
        const size_t jj = (size_t)(j - p->col0);
        rlo[jj] = (int32_t)off[(size_t)K0];
        cnt[jj] = (int32_t)rows_n;
        offs[jj] = start0 + (j - off[(size_t)J]) * stride;
        pairs_ok = pairs_ok && (((off[(size_t)K0] | rows_n) & 1) == 0);
        r0 = std::min<int64_t>(r0, off[(size_t)K0]);
        r1 = std::max<int64_t>(r1, off[(size_t)K1 + 1]);
        dmin = std::min<int64_t>(dmin, offs[jj]);
        dmax = std::max<int64_t>(dmax, offs[jj] + rows_n);
    }
    if (nloc == 0) { r0 = r1 = 0; dmin = dmax = 0; }
    // outputs are relative to the first local stored value
    for (auto &o : offs) {
        o -= dmin;
        pairs_ok = pairs_ok && ((o & 1) == 0);
    }
    p->cr_pairs = pairs_ok && nloc > 0;
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
    FD_TRY(dev_upload(&p->d_cr_rlo, rlo));
    FD_TRY(dev_upload(&p->d_cr_cnt, cnt));
    FD_TRY(dev_upload(&p->d_cr_off, offs));
    FD_TRY(alloc_scratch(p, col0));
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

int fd_plan_info(const fd_plan *p, int key, int64_t *value)
{
    FD_REQUIRE(p && value, FD_ERR_ARG, "NULL argument");
    switch (key) {
    case FD_INFO_M: *value = p->M; break;
    case FD_INFO_N: *value = p->N; break;
    case FD_INFO_NCOLORS: *value = p->C; break;
    case FD_INFO_NOUTS: *value = p->split_n ? p->split_n : p->nouts; break;
    case FD_INFO_OUT0_LEN: *value = p->split_n ? p->split_len[0] : p->out_len[0]; break;
    case FD_INFO_OUT1_LEN: *value = p->split_n ? p->split_len[1] : p->out_len[1]; break;
    case FD_INFO_OUT2_LEN: *value = p->split_n ? p->split_len[2] : p->out_len[2]; break;
    case FD_INFO_ROW_BEGIN: *value = p->row0; break;
    case FD_INFO_ROW_END: *value = p->row1; break;
    case FD_INFO_NCHUNKS: *value = p->nchunks; break;
    case FD_INFO_SCRATCH_BYTES:
        *value = (int64_t)sizeof(real_t) * (p->chunkB * p->pts * p->cplx * (p->ldx + p->ldf) + 2 * p->ldf + p->ldx);
        break;
    case FD_INFO_NNZ_LOCAL: *value = p->nnz_local; break;
    case FD_INFO_FCALLS_LAST: *value = p->fcalls_last; break;
    case FD_INFO_ENTRY_BEGIN: *value = p->entry_begin; break;
    case FD_INFO_SORTED_GATHER: *value = p->sorted_gather ? 1 : 0; break;
    case FD_INFO_LINES_DIRECT_X100: *value = (int64_t)(p->lines_direct * 100); break;
    case FD_INFO_LINES_SORTED_X100: *value = (int64_t)(p->lines_sorted * 100); break;
    case FD_INFO_WINDOW: *value = (p->window || p->tri_window) ? 1 : 0; break;
    case FD_INFO_WIN_OVERREAD_X100: *value = (int64_t)(p->win_overread * 100); break;
    case FD_INFO_WINDOW2D: *value = p->window2d ? 1 : 0; break;
    case FD_INFO_WIN_PERIOD: *value = p->win_per_P; break;
    case FD_INFO_COLRANGE_WG: *value = p->kind == K_COLRANGE ? 1 : 0; break;
    case FD_INFO_SMALL_FUSED:
        *value = (p->small_ok && p->N <= kSmallN && p->C > 0 && p->C <= kRegColors && p->kind != K_DENSE &&
                  p->fdtype != FD_COMPLEX) ? 1 : 0;
        break;
    case FD_INFO_EPS_CYCLIC: *value = p->cyc_C; break;
    case FD_INFO_EPS_NT: *value = (p->eps_nt_forced >= 0 ? p->eps_nt_forced != 0 : !store_active(p)) ? 1 : 0; break;
    case FD_INFO_BAND_DESC: *value = p->bd_t1 - p->bd_t0; break;
    case FD_INFO_LAZY_STORE:
        *value = (store_active(p) || store_csc_active(p)) ? 1 : 0;
        break;
    case FD_INFO_STORE_CSC:
        *value = p->store_csc_ok ? p->sc_entries : 0;
        break;
    case FD_INFO_LAZY_DIFF:
        *value = (p->lazy_fn && (p->lazy_caps & FD_LAZY_CAP_DIFF) && p->lazy_diff && p->fdtype != FD_COMPLEX && p->kind != K_DENSE) ? 1 : 0;
        break;
    case FD_INFO_BUILT_ON_DEVICE: *value = p->built_on_device ? 1 : 0; break;
    default: set_error("unknown info key %d", key); return FD_ERR_ARG;
    }
    return FD_OK;
}

// ---- timing spans ---------------------------------------------------------------------------
static hipEvent_t take_event(fd_plan *p)
{
    if (!p->event_pool.empty()) {
        hipEvent_t e = p->event_pool.back();
        p->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct Span {
    fd_plan *p;
    int idx = -1;
    Span(fd_plan *pl, int stage) : p(pl)
    {
        // level 1: only the graded kernel (2 events per call); level 2: every stage + the whole call; level 3: the whole call only
        if (p->timing == 0) return;
        if (p->timing == 1 && stage != FD_STAGE_DECOMPRESS) return;
        if (p->timing == 3 && stage != FD_STAGE_TOTAL) return;
        fdjac::TimedSpan s{stage, take_event(p), take_event(p)};
        (void)hipEventRecord(s.a, p->ctx->stream);
        p->spans.push_back(s);
        idx = (int)p->spans.size() - 1;
    }
    void stop()
    {
        if (idx >= 0) (void)hipEventRecord(p->spans[(size_t)idx].b, p->ctx->stream);
        idx = -1;
    }
    ~Span() { stop(); }
};

// Reads the spans whose events have completed.  blocking = false (start of every call) never waits for the
// device: the stream keeps running ahead of the host; spans are recorded in stream order, so the first one still
// in flight ends the sweep.  The list is bounded by a blocking sweep once kMaxSpansInFlight are outstanding.
static constexpr size_t kMaxSpansInFlight = 4096;
static int collect_spans(fd_plan *p, bool blocking = true)
{
    if (p->spans.empty()) return FD_OK;
    if (!blocking && p->spans.size() >= kMaxSpansInFlight) blocking = true;
    if (blocking) FD_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    size_t done = 0;
    for (auto &s : p->spans) {
        if (!blocking && hipEventQuery(s.b) != hipSuccess) {
            (void)hipGetLastError();   // hipErrorNotReady is not an error: keep it out of the launch checks
            break;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            p->ms_sum[s.stage] += ms;
            p->launches[s.stage] += 1;
            if (p->samples[s.stage].size() < 65536) p->samples[s.stage].push_back(ms);
        }
        p->event_pool.push_back(s.a);
        p->event_pool.push_back(s.b);
        ++done;
    }
    p->spans.erase(p->spans.begin(), p->spans.begin() + (ptrdiff_t)done);
    return FD_OK;
}

int fd_plan_enable_timing(fd_plan *p, int on)
{
    FD_REQUIRE(p, FD_ERR_ARG, "plan is NULL");
    int rc = collect_spans(p);
    if (rc) return rc;
    p->timing = on < 0 ? 0 : (on > 3 ? 3 : on);
    for (int i = 0; i < FD_NSTAGES; ++i) {
        p->ms_sum[i] = 0;
        p->launches[i] = 0;
        p->samples[i].clear();
    }
    return FD_OK;
}

int fd_plan_get_timing_samples(fd_plan *p, int stage, double *ms_out, int64_t cap, int64_t *n_out)
{
    FD_REQUIRE(p && n_out && stage >= 0 && stage < FD_NSTAGES && (ms_out || cap == 0), FD_ERR_ARG, "bad argument");
    int rc = collect_spans(p);
    if (rc) return rc;
    const int64_t n = std::min<int64_t>((int64_t)p->samples[stage].size(), cap);
    for (int64_t k = 0; k < n; ++k) ms_out[k] = (double)p->samples[stage][(size_t)k];
    *n_out = (int64_t)p->samples[stage].size();
    return FD_OK;
}

int fd_plan_get_timings(fd_plan *p, double *ms_sum, int64_t *launches)
{
    FD_REQUIRE(p && ms_sum && launches, FD_ERR_ARG, "NULL argument");
    int rc = collect_spans(p);
    if (rc) return rc;
    for (int i = 0; i < FD_NSTAGES; ++i) {
        ms_sum[i] = p->ms_sum[i];
        launches[i] = p->launches[i];
    }
    return FD_OK;
}

// the register-path reduction over a global grid of blocks is the one that can be split across ranks / shards
static bool eps_shardable(const fd_plan *p)
{
    return p->fdtype != FD_COMPLEX && p->kind != K_DENSE && p->C > 0 && p->C <= kRegColors &&
           !(p->small_ok && p->N <= kSmallN) && p->d_partial != nullptr;
}

// The plain f! launcher.  For a plan of the lowered complex-valued-x problem the arrays hold (re, im) pairs: the launcher is
// called as the header promises for complex elements -- is_complex = 1, strides and rows counted in complex elements.
static inline int call_f(const fd_plan *p, fd_f_launch f, void *fctx, void *fx, const void *x, int64_t nbatch, int64_t x_stride,
                         int64_t fx_stride, int64_t row_begin, int64_t row_end, int is_complex, void *stream)
{
    if (!p->cx) return f(fctx, fx, x, nbatch, x_stride, fx_stride, row_begin, row_end, is_complex, stream);
    return f(fctx, fx, x, nbatch, x_stride / 2, fx_stride / 2, row_begin / 2, (row_end + 1) / 2, 1, stream);
}

// ---- the hot path ---------------------------------------------------------------------------
static int jacobian_enqueue(fd_plan *p, fd_f_launch f, void *fctx, const real_t *x_dev, const real_t *fin_dev,
                            double relstep, double absstep, double dir, real_t *const *outs)
{
    fd_ctx *ctx = p->ctx;
    hipStream_t s = ctx->stream;
    FD_REQUIRE(f != nullptr, FD_ERR_ARG, "f launcher is NULL");
    if (!(relstep > 0)) {
        // default_relstep, src/epsilons.jl:133-144
        const real_t e = std::numeric_limits<real_t>::epsilon();   // default_relstep(fdtype, eltype(x))
        relstep = p->fdtype == FD_FORWARD ? (double)std::sqrt(e) : p->fdtype == FD_CENTRAL ? (double)std::cbrt(e) : 1.0;
    }
    if (absstep < 0) absstep = relstep;
    p->relstep_last = relstep;
    p->absstep_last = absstep;
    p->fcalls_last = 0;
    if (collect_spans(p, false) != FD_OK) return FD_ERR_HIP;  // harvest finished spans, never wait for the device
    Span total(p, FD_STAGE_TOTAL);

    // x must be 16-B aligned for the vector loads; stage it otherwise
    if (((uintptr_t)x_dev) & kPairMask) {
        { const int rc = ensure_stage(p, true, false); if (rc) return rc; }
        FD_HIP_CHECK(hipMemcpyAsync(p->d_xstage, x_dev, sizeof(real_t) * (size_t)p->N, hipMemcpyDeviceToDevice, s));
        x_dev = p->d_xstage;
    }

    // Small problems are launch-latency bound: one single-workgroup launch computes the step sizes and, when the
    // points are materialised, writes them too -- with x itself as one more batch member so that f(x) needs no launch
    // of its own.  (Complex step: no step-size reduction to fuse with; large or many-coloured problems: the wide path.)
    const bool small_off = !p->small_ok;
    const bool full_colors = p->own_c0 == 0 && (p->own_c1 < 0 || p->own_c1 >= p->C);
    // (which reduction computes the step sizes depends on N and C only, so column windows, colour ownership and colour
    // chunks of the same problem all see bit-identical step sizes)
    const bool small = !small_off && p->N <= kSmallN && p->C > 0 && p->C <= kRegColors && p->kind != K_DENSE &&
                       p->fdtype != FD_COMPLEX;
    const bool small_points = small && !p->lazy_fn && p->nchunks == 1 && full_colors;   // points written by the fused launch
    const bool base_in_batch = small_points && p->fdtype == FD_FORWARD && !fin_dev;
    p->fx_batch_row = nullptr;
    if (small_points) { const int rc = ensure_points(p); if (rc) return rc; }

    // a launcher that hands over central differences: the doubled step sizes come out of the finalize launch
    if (p->fdtype == FD_CENTRAL && p->lazy_fn && (p->lazy_caps & FD_LAZY_CAP_DIFF) && p->lazy_diff && p->kind != K_DENSE) {
        const int rc = ensure_diff_scratch(p);
        if (rc) return rc;
    }
    if (p->eps_mode != FD_EPS_PRECOMPUTED) p->eps2_fresh = false;
    p->eps_nt = p->eps_nt_forced >= 0 ? p->eps_nt_forced != 0 : !(store_active(p) || store_csc_active(p));   // (see apply_opts)
    // step sizes for every colour (one pass over x), src/jacobians.jl:559-561 / 600-602
    if (p->fdtype != FD_COMPLEX && p->C > 0 && p->eps_mode == FD_EPS_PRECOMPUTED) {
        // the caller ran fd_plan_eps_partials / exchanged / fd_plan_eps_finalize: p->d_eps is current
        FD_REQUIRE(!small_points, FD_ERR_UNSUPPORTED, "FD_EPS_PRECOMPUTED needs a plan whose reduction can be sharded");
    } else if (p->fdtype != FD_COMPLEX && p->C > 0) {
        Span sp(p, FD_STAGE_EPS);
        int rc;
        if (small) {
            rc = launch_eps_perturb_small(p, x_dev, relstep, absstep, dir, small_points ? p->fdtype : -1,
                                          base_in_batch ? (int)(p->C * p->pts) : -1);
        } else if (p->comm && eps_shardable(p)) {   // (also with a single-rank communicator: same code path)
            // sharded reduction: this rank's blocks of the global grid, all-gather of the partial sums (in place, padded
            // slots), the same finalize as the unsharded call => bit-identical step sizes on every rank
            const int W = fdjac_comm_nranks(p->comm), r = fdjac_comm_rank(p->comm);
            const int S = (p->n_partial_blocks + W - 1) / W;
            const int b0 = std::min(r * S, p->n_partial_blocks), nb = std::min(S, p->n_partial_blocks - b0);
            rc = launch_eps_partial(p, x_dev, b0, nb);
            if (!rc) rc = fdjac_comm_allgather_f64(p->comm, p->d_partial, (int64_t)S * kRegColors);
            if (!rc) rc = launch_eps_finalize(p, p->n_partial_blocks, kRegColors, relstep, absstep, dir);
        } else {
            rc = launch_eps(p, x_dev, relstep, absstep, dir);
        }
        if (rc) return rc;
    }

    // f(x) for forward differences (src/jacobians.jl:540-545).  With a lazy-point launcher the base
    // evaluation rides along with the first perturbed batch (one launch fewer).
    const real_t *fx = nullptr;
    bool base_pending = false;
    if (p->fdtype == FD_FORWARD) {
        if (fin_dev) {
            fx = fin_dev;
        } else if (base_in_batch) {
            fx = p->fx_batch_row = p->d_FX + (int64_t)p->C * p->pts * p->ldf;
        } else if (p->lazy_fn && p->nchunks > 0) {
            base_pending = true;
            fx = p->d_fx;
        } else {
            Span sp(p, FD_STAGE_F);
            const int rc = call_f(p, f, fctx, p->d_fx, x_dev, 1, p->N, p->ldf, p->row0, p->row1, 0, (void *)s);
            FD_REQUIRE(rc == 0, FD_ERR_CALLBACK, "f! launcher returned %d", rc);
            p->fcalls_last += 1;
            fx = p->d_fx;
        }
    }

    // dense J / list kinds with several chunks or uncovered entries start from zero (fill_matrix!,
    // src/jacobians.jl:530-532).  Kinds that write every stored value in one chunk skip the fill.
    if (p->kind == K_CSC_DENSE || p->kind == K_COO_DENSE) {
        FD_HIP_CHECK(hipMemsetAsync(outs[0], 0, sizeof(real_t) * (size_t)p->out_len[0], s));
    } else if (p->C == 0) {
        for (int k = 0; k < p->nouts; ++k)
            FD_HIP_CHECK(hipMemsetAsync(outs[k], 0, sizeof(real_t) * (size_t)p->out_len[k], s));
    }

    // colours of this plan: all of them, or the owned range (fd_plan_opts.color_begin/end), in chunks of chunkB
    const int64_t oc0 = std::min<int64_t>(p->own_c0, p->C), oc1 = p->own_c1 < 0 ? p->C : std::min<int64_t>(p->own_c1, p->C);
    bool diff_base_counted = false;
    for (int64_t cl = oc0; cl < oc1; cl += p->chunkB) {
        const int c_lo = (int)cl;
        const int c_hi = (int)std::min<int64_t>(oc1, cl + p->chunkB);
        const int B = c_hi - c_lo;
        bool lazy_done = false, imag_only = false;
        // a launcher that can, hands over DIFFERENCES (f(point) - f(x), or f(plus) - f(minus)): no f(x) pass / half the f!
        // arrays, and the decompression reads one array per colour (the forward kernels with fx = 0 and, for central
        // differences, the doubled step sizes: (a - 0.0) / (2 eps) -- the bits of the plain path)
        bool want_diff = p->lazy_fn && (p->lazy_caps & FD_LAZY_CAP_DIFF) && p->lazy_diff && p->fdtype != FD_COMPLEX &&
                               !(p->fdtype == FD_FORWARD && !base_pending) && p->kind != K_DENSE;
        if (want_diff) { const int rc = ensure_diff_scratch(p); if (rc) return rc; }
        // the launcher stores the finished quotients into the Jacobian itself (include/fdjac_device.h) -- the exact band was
        // verified at plan time -- and nothing is launched after f!: this launch IS the difference + decompression
        // (src/jacobians.jl:565-568), so its span is recorded as the graded stage
        if (store_active(p) && !(p->fdtype == FD_FORWARD && !base_pending)) {
            Span sp(p, FD_STAGE_DECOMPRESS);
            fd_band_store bs;
            fd_stencil5_store s5;
            memset(&bs, 0, sizeof bs);
            memset(&s5, 0, sizeof s5);
            s5.out = outs[0];
            s5.nx = p->store5_nx; s5.ny = p->store5_ny; s5.entry_begin = p->entry_begin; s5.col_begin = p->col0; s5.col_end = p->col1;
            s5.color = p->d_color; s5.color_bytes = p->color8 ? 1 : 4; s5.C = (int)p->C; s5.elem_bytes = (int)sizeof(real_t);
            const bool stencil = !p->store_ok && p->store5_ok && p->kind == K_CSC;
            fd_colrange_store scr;
            memset(&scr, 0, sizeof scr);
            if (p->kind == K_COLRANGE) {
                scr.out = outs[0]; scr.M = p->M; scr.N = p->N; scr.col_begin = p->col0; scr.col_end = p->col1;
                scr.row_first = p->d_cr_rlo; scr.row_count = p->d_cr_cnt; scr.dest = (const long long *)p->d_cr_off;
                scr.color = p->d_color; scr.color_bytes = p->color8 ? 1 : 4; scr.C = (int)p->C; scr.elem_bytes = (int)sizeof(real_t);
                scr.pairs = p->cr_pairs ? 1 : 0; scr.nblk = p->cr_nblk; scr.block_size = p->cr_bs; scr.bl = p->cr_bl; scr.bu = p->cr_bu;
            }
            bs.M = p->M; bs.N = p->N; bs.entry_begin = p->entry_begin; bs.col_begin = p->col0; bs.col_end = p->col1;
            bs.l = p->store_l; bs.u = p->store_u; bs.C = p->store_C; bs.shift = p->store_shift;
            bs.elem_bytes = (int)sizeof(real_t);
            if (p->kind == K_TRIDIAG) {
                bs.layout = FD_BAND_TRIDIAGONAL;
                bs.out_dl = outs[0]; bs.out = outs[1]; bs.out_du = outs[2];
            } else {
                bs.layout = p->kind == K_BANDED ? FD_BAND_BANDED : FD_BAND_CSC;
                bs.out = outs[0];
            }
            fd_lazy_points lp = {};
            lp.x = x_dev;
            lp.color = p->d_color;
            lp.eps = p->d_eps;
            lp.color_bytes = p->color8 ? 1 : 4;
            lp.c_lo = c_lo;
            lp.ncolors = B;
            lp.pts = p->pts;
            lp.nparts = 1;
            lp.diff = p->fdtype == FD_COMPLEX ? 0 : ((p->fdtype == FD_FORWARD && !diff_base_counted) ? 2 : 1);
            lp.store = p->kind == K_COLRANGE ? (const void *)&scr : stencil ? (const void *)&s5 : (const void *)&bs;
            lp.store_kind = p->kind == K_COLRANGE ? FD_STORE_COLRANGE : stencil ? FD_STORE_STENCIL5 : FD_STORE_BAND;
            lp.is_complex = p->fdtype == FD_COMPLEX ? 1 : 0;
            lp.imag_only = lp.is_complex;
            const int rc = p->lazy_fn(fctx, p->d_FX, &lp, p->ldf, p->row0, p->row1, (void *)s);
            FD_REQUIRE(rc == 0 || rc == FD_LAZY_DECLINED, FD_ERR_CALLBACK, "lazy f! launcher (store) returned %d", rc);
            if (rc == 0) {
                p->fcalls_last += (int64_t)B * p->pts + (lp.diff == 2 ? 1 : 0);
                diff_base_counted = true;
                continue;
            }
        }
        // ... or, for ANY pattern and colouring, column by column through the compact copy of the pattern (FD_PLAN_STORE_CSC,
        // fd_csc_store): a launcher registered with FD_LAZY_CAP_STORE_CSC evaluates the row of every stored entry at its column's
        // colour point and stores the quotient.  Forward differences subtract f(x): the caller's f_in, or ONE plain evaluation
        // (src/jacobians.jl:540-545) -- M + nnz row evaluations instead of (1 + C) M.
        if (store_csc_active(p)) {
            if (base_pending) {
                Span sp(p, FD_STAGE_F);
                const int rc = call_f(p, f, fctx, p->d_fx, x_dev, 1, p->N, p->ldf, p->row0, p->row1, 0, (void *)s);
                FD_REQUIRE(rc == 0, FD_ERR_CALLBACK, "f! launcher returned %d", rc);
                p->fcalls_last += 1;
                base_pending = false;
            }
            Span sp(p, FD_STAGE_DECOMPRESS);
            fd_csc_store sc;
            memset(&sc, 0, sizeof sc);
            sc.out = outs[0]; sc.M = p->M; sc.N = p->N; sc.col_begin = p->col0; sc.col_end = p->col1;
            sc.colptr = p->d_sc_colptr; sc.rowval = p->d_sc_rowval; sc.color = p->d_color; sc.fx_base = p->fdtype == FD_FORWARD ? fx : nullptr;
            sc.color_bytes = p->color8 ? 1 : 4; sc.C = (int)p->C; sc.elem_bytes = (int)sizeof(real_t); sc.valid_coloring = p->sc_valid ? 1 : 0;
            fd_lazy_points lp = {};
            lp.x = x_dev;
            lp.color = p->d_color;
            lp.eps = p->d_eps;
            lp.color_bytes = sc.color_bytes;
            lp.c_lo = c_lo;
            lp.ncolors = B;
            lp.pts = p->pts;
            lp.nparts = 1;
            lp.diff = 1;
            lp.store = &sc;
            lp.store_kind = FD_STORE_CSC;
            const int rc = p->lazy_fn(fctx, p->d_FX, &lp, p->ldf, p->row0, p->row1, (void *)s);
            FD_REQUIRE(rc == 0 || rc == FD_LAZY_DECLINED, FD_ERR_CALLBACK, "lazy f! launcher (column store) returned %d", rc);
            if (rc == 0) {
                p->fcalls_last += (int64_t)B * p->pts;
                continue;
            }
            if (p->fdtype == FD_FORWARD) want_diff = false;      // (declined: f(x) exists already -- plain values are handed over below)
        }
        bool diff_done = false;
        if (p->lazy_fn) {
            Span sp(p, FD_STAGE_F);
            fd_lazy_points lp = {};
            lp.x = x_dev;
            lp.color = p->d_color;
            lp.eps = p->d_eps;
            lp.base_out = (base_pending && !want_diff) ? p->d_fx : nullptr;
            lp.diff = want_diff ? ((p->fdtype == FD_FORWARD && !diff_base_counted) ? 2 : 1) : 0;
            lp.store_kind = 0;
            lp.color_bytes = p->color8 ? 1 : 4;
            lp.c_lo = c_lo;
            lp.ncolors = B;
            lp.pts = p->pts;
            lp.is_complex = p->fdtype == FD_COMPLEX ? 1 : 0;
            // complex step: only imag(f) is ever used (src/jacobians.jl:635) -- a launcher that can, writes just that
            lp.imag_only = (p->fdtype == FD_COMPLEX && (p->lazy_caps & FD_LAZY_CAP_IMAG_ONLY)) ? 1 : 0;
            lp.part = 0;
            lp.nparts = 1;
            imag_only = lp.imag_only != 0;
            const int rc = p->lazy_fn(fctx, p->d_FX, &lp, p->ldf, p->row0, p->row1, (void *)s);
            FD_REQUIRE(rc == 0 || rc == FD_LAZY_DECLINED, FD_ERR_CALLBACK, "lazy f! launcher returned %d", rc);
            if (rc == 0 && want_diff) {
                lazy_done = diff_done = true;          // (base_pending stays: a later batch the launcher declines needs f(x))
                p->fcalls_last += (int64_t)B * p->pts + (lp.diff == 2 ? 1 : 0);
                diff_base_counted = true;
            } else if (rc == 0) {
                lazy_done = true;
                p->fcalls_last += (int64_t)B * p->pts + (base_pending ? 1 : 0);
                base_pending = false;
            }
        }
        if (!lazy_done) {
            if (base_pending) {   // the lazy launcher declined the batch that would have carried f(x)
                Span sp(p, FD_STAGE_F);
                const int rc = call_f(p, f, fctx, p->d_fx, x_dev, 1, p->N, p->ldf, p->row0, p->row1, 0, (void *)s);
                FD_REQUIRE(rc == 0, FD_ERR_CALLBACK, "f! launcher returned %d", rc);
                p->fcalls_last += 1;
                base_pending = false;
            }
            { const int rc = ensure_points(p); if (rc) return rc; }
            if (!small_points) {
                Span sp(p, FD_STAGE_PERTURB);
                int rc = launch_perturb(p, x_dev, c_lo, B);
                if (rc) return rc;
            }
            Span sp(p, FD_STAGE_F);
            const int64_t npts = (int64_t)B * p->pts + (base_in_batch ? 1 : 0);   // (+ x itself: f(x) of the forward arm)
            const int rc = call_f(p, f, fctx, p->d_FX, p->d_X, npts, p->ldx, p->ldf, p->row0, p->row1,
                             p->fdtype == FD_COMPLEX ? 1 : 0, (void *)s);
            FD_REQUIRE(rc == 0, FD_ERR_CALLBACK, "f! launcher returned %d", rc);
            p->fcalls_last += npts;
        }
        {
            Span sp(p, FD_STAGE_DECOMPRESS);
            // imaginary parts as a real array: the forward kernels with fx = 0 compute (a - 0.0)/eps == a/eps bit for bit
            const bool io = lazy_done && imag_only;
            int rc;
            if (diff_done) {
                real_t *eps_plain = p->d_eps;
                if (p->fdtype == FD_CENTRAL) {
                    if (!p->eps2_fresh && (rc = launch_scale(p->ctx, p->d_eps2 + c_lo, p->d_eps + c_lo, B, (real_t)2))) return rc;
                    p->d_eps = p->d_eps2;
                }
                rc = launch_decompress(p, p->d_zero, c_lo, c_hi, outs, (int)FD_FORWARD);
                p->d_eps = eps_plain;
            } else {
                rc = launch_decompress(p, io ? p->d_fx : fx, c_lo, c_hi, outs, io ? (int)FD_FORWARD : p->fdtype);
            }
            if (rc) return rc;
        }
    }
    return FD_OK;
}

// complex-valued x on Tridiagonal storage: the lowered plan fills ONE concatenated array (dl | d | du), copied into the caller's three
static int jacobian_split(fd_plan *p, bool async, fd_f_launch f, void *fctx, const void *x, int x_kind, const void *f_in, int f_in_kind, double relstep,
                          double absstep, double dir, void *const *outs, int out_kind)
{
    for (int k = 0; k < p->split_n; ++k) FD_REQUIRE(outs[k] || p->split_len[k] == 0, FD_ERR_ARG, "outs[%d] is NULL", k);
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    if (!p->d_split) { const int rc = dev_alloc(&p->d_split, p->out_len[0]); if (rc) return rc; }
    void *one[3] = {p->d_split, nullptr, nullptr};
    const int n = p->split_n;
    p->split_n = 0;                 // (the inner call sees the plan's one output)
    const int rc = async ? fd_jacobian_async(p, f, fctx, x, f_in, relstep, absstep, dir, one)
                         : fd_jacobian(p, f, fctx, x, x_kind, f_in, f_in_kind, relstep, absstep, dir, one, FD_DEVICE);
    p->split_n = n;
    if (rc) return rc;
    int64_t off = 0;
    for (int k = 0; k < n; ++k) {
        if (p->split_len[k] > 0)
            FD_HIP_CHECK(hipMemcpyAsync(outs[k], p->d_split + off, sizeof(real_t) * (size_t)p->split_len[k],
                                        out_kind == FD_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, p->ctx->stream));
        off += p->split_len[k];
    }
    if (!async) FD_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    return FD_OK;
}

int fd_jacobian_async(fd_plan *p, fd_f_launch f, void *fctx, const void *x, const void *f_in, double relstep,
                      double absstep, double dir, void *const *outs)
{
    FD_REQUIRE(p && x && outs, FD_ERR_ARG, "NULL argument");
    if (p->split_n) return jacobian_split(p, true, f, fctx, x, FD_DEVICE, f_in, FD_DEVICE, relstep, absstep, dir, outs, FD_DEVICE);
    for (int k = 0; k < p->nouts; ++k) FD_REQUIRE(outs[k] || p->out_len[k] == 0, FD_ERR_ARG, "outs[%d] is NULL", k);
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    real_t *o[3] = {(real_t *)outs[0], p->nouts > 1 ? (real_t *)outs[1] : nullptr,
                    p->nouts > 2 ? (real_t *)outs[2] : nullptr};
    return jacobian_enqueue(p, f, fctx, (const real_t *)x, (const real_t *)f_in, relstep, absstep, dir, o);
}

int fd_jacobian(fd_plan *p, fd_f_launch f, void *fctx, const void *x, int x_kind, const void *f_in, int f_in_kind,
                double relstep, double absstep, double dir, void *const *outs, int out_kind)
{
    FD_REQUIRE(p && x && outs, FD_ERR_ARG, "NULL argument");
    if (p->split_n) return jacobian_split(p, false, f, fctx, x, x_kind, f_in, f_in_kind, relstep, absstep, dir, outs, out_kind);
    for (int k = 0; k < p->nouts; ++k) FD_REQUIRE(outs[k] || p->out_len[k] == 0, FD_ERR_ARG, "outs[%d] is NULL", k);
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const real_t *x_dev = (const real_t *)x;
    { const int rc = ensure_stage(p, x_kind == FD_HOST, f_in && f_in_kind == FD_HOST); if (rc) return rc; }
    if (x_kind == FD_HOST) {
        FD_HIP_CHECK(hipMemcpyAsync(p->d_xstage, x, sizeof(real_t) * (size_t)p->N, hipMemcpyHostToDevice, s));
        x_dev = p->d_xstage;
    }
    const real_t *fin_dev = (const real_t *)f_in;
    if (f_in && f_in_kind == FD_HOST) {
        FD_HIP_CHECK(hipMemcpyAsync(p->d_finstage, f_in, sizeof(real_t) * (size_t)p->M, hipMemcpyHostToDevice, s));
        fin_dev = p->d_finstage;
    }
    real_t *o[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < p->nouts; ++k) {
        if (out_kind == FD_DEVICE) {
            o[k] = (real_t *)outs[k];
        } else {
            if (!p->d_outstage[k]) {
                int rc = dev_alloc(&p->d_outstage[k], p->out_len[k]);
                if (rc) return rc;
            }
            o[k] = p->d_outstage[k];
        }
    }
    int rc = jacobian_enqueue(p, f, fctx, x_dev, fin_dev, relstep, absstep, dir, o);
    if (rc) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    if (out_kind == FD_HOST)
        for (int k = 0; k < p->nouts; ++k)
            if (p->out_len[k] > 0)
                FD_HIP_CHECK(hipMemcpyAsync(outs[k], o[k], sizeof(real_t) * (size_t)p->out_len[k], hipMemcpyDeviceToHost, s));
    FD_HIP_CHECK(hipStreamSynchronize(s));
    return FD_OK;
}

int fd_plan_set_lazy_f(fd_plan *p, fd_f_launch_lazy lazy)
{
    FD_REQUIRE(p != nullptr, FD_ERR_ARG, "plan is NULL");
    p->lazy_fn = p->cx ? nullptr : lazy;   // (complex-valued x: materialised points only -- the lazy protocol describes real points)
    p->lazy_caps = 0;
    return FD_OK;
}

int fd_plan_set_lazy_caps(fd_plan *p, int caps)
{
    FD_REQUIRE(p != nullptr, FD_ERR_ARG, "plan is NULL");
    p->lazy_caps = caps;
    return FD_OK;
}

int fd_plan_set_comm(fd_plan *p, fd_comm *comm)
{
    FD_REQUIRE(p != nullptr, FD_ERR_ARG, "plan is NULL");
    FD_REQUIRE(comm == nullptr || fdjac_comm_ctx(comm) == p->ctx, FD_ERR_ARG, "the communicator belongs to another context");
    FD_REQUIRE(comm == nullptr || fdjac_comm_nranks(comm) <= kMaxEpsShards, FD_ERR_UNSUPPORTED, "more than %d ranks", kMaxEpsShards);
    if (comm) {
        // every rank must cut the SAME global grid of blocks (it is a function of N, the colour count and the map -- but a rank
        // with another FDJAC_GRID_CAP / FDJAC_EPS_CONTIG would use other slots: mismatched all-gather counts hang, matched
        // ones finalize garbage), and every rank must agree on WHETHER its reduction is sharded at all (a rank whose plan keeps
        // the replicated reduction would skip the per-call all-gather the others enter).  One tiny all-reduce at attach time
        // settles both; it is unconditional -- every rank enters it whatever its own plan looks like.
        const bool sh = eps_shardable(p);
        const double nb = sh ? (double)p->n_partial_blocks : -1.0, tp = sh ? (double)p->eps_tpb : -1.0;
        const double mine[4] = {nb, -nb, tp, -tp};
        double got[4] = {0, 0, 0, 0};
        const int rc = fdjac_comm_allreduce_max4(comm, mine, got);
        if (rc) return rc;
        FD_REQUIRE(got[0] == mine[0] && got[1] == mine[1] && got[2] == mine[2] && got[3] == mine[3], FD_ERR_COMM,
                   "the ranks disagree on the step-size reduction (this rank: %s, %d blocks of %d tiles): same N, colours, fdtype, "
                   "FDJAC_SMALL, FDJAC_GRID_CAP and FD_PLAN_EPS_CONTIGUOUS everywhere?", sh ? "sharded" : "replicated",
                   p->n_partial_blocks, p->eps_tpb);
    }
    p->comm = comm;
    return FD_OK;
}

int fd_plan_eps_partials(fd_plan *p, const void *x_dev, int shard, int nshards, void **partials_out,
                         int64_t *slot_doubles_out)
{
    FD_REQUIRE(p && x_dev, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(nshards >= 1 && nshards <= kMaxEpsShards && shard >= 0 && shard < nshards, FD_ERR_ARG,
               "shard %d of %d", shard, nshards);
    FD_REQUIRE(eps_shardable(p), FD_ERR_UNSUPPORTED,
               "this plan's step-size reduction cannot be sharded (needs 1..8 colours, N > 16384, forward / central)");
    FD_REQUIRE((((uintptr_t)x_dev) & kPairMask) == 0, FD_ERR_ARG, "x must be 16-byte aligned");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    const int S = (p->n_partial_blocks + nshards - 1) / nshards;
    const int b0 = std::min(shard * S, p->n_partial_blocks), nb = std::min(S, p->n_partial_blocks - b0);
    if (partials_out) *partials_out = p->d_partial;
    if (slot_doubles_out) *slot_doubles_out = (int64_t)S * kRegColors;
    return launch_eps_partial(p, (const real_t *)x_dev, b0, nb);
}

int fd_plan_eps_finalize(fd_plan *p, double relstep, double absstep, double dir)
{
    FD_REQUIRE(p != nullptr, FD_ERR_ARG, "plan is NULL");
    FD_REQUIRE(eps_shardable(p), FD_ERR_UNSUPPORTED, "this plan's step-size reduction cannot be sharded");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    if (!(relstep > 0)) {
        const real_t e = std::numeric_limits<real_t>::epsilon();
        relstep = p->fdtype == FD_FORWARD ? (double)std::sqrt(e) : (double)std::cbrt(e);
    }
    if (absstep < 0) absstep = relstep;
    return launch_eps_finalize(p, p->n_partial_blocks, kRegColors, relstep, absstep, dir);
}

int fd_plan_eps_shard_range(fd_plan *p, int shard, int nshards, int64_t *x_begin, int64_t *x_end)
{
    FD_REQUIRE(p && x_begin && x_end, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(nshards >= 1 && nshards <= kMaxEpsShards && shard >= 0 && shard < nshards, FD_ERR_ARG, "shard %d of %d", shard, nshards);
    FD_REQUIRE(eps_shardable(p), FD_ERR_UNSUPPORTED, "this plan's step-size reduction cannot be sharded");
    if (p->eps_tpb <= 0) { *x_begin = 0; *x_end = p->N; return FD_OK; }      // grid-stride: every block reads all over x
    const int S = (p->n_partial_blocks + nshards - 1) / nshards;
    const int64_t per_block = (int64_t)p->eps_tpb * 2048;                     // k_eps_partial_reg's tile: 4 x 512 elements
    const int64_t b0 = std::min<int64_t>((int64_t)shard * S, p->n_partial_blocks), b1 = std::min<int64_t>(b0 + S, p->n_partial_blocks);
    *x_begin = std::min<int64_t>(b0 * per_block, p->N);
    *x_end = shard == nshards - 1 ? p->N : std::min<int64_t>(b1 * per_block, p->N);
    return FD_OK;
}

int fd_plan_set_eps_mode(fd_plan *p, int mode)
{
    FD_REQUIRE(p != nullptr, FD_ERR_ARG, "plan is NULL");
    FD_REQUIRE(mode == FD_EPS_COMPUTE || mode == FD_EPS_PRECOMPUTED, FD_ERR_ARG, "unknown eps mode %d", mode);
    FD_REQUIRE(mode == FD_EPS_COMPUTE || eps_shardable(p), FD_ERR_UNSUPPORTED,
               "this plan's step-size reduction cannot be sharded");
    p->eps_mode = mode;
    return FD_OK;
}

int fd_plan_get_epsilons(fd_plan *p, double *eps_out)
{
    FD_REQUIRE(p && eps_out, FD_ERR_ARG, "NULL argument");
    if (p->C == 0) return FD_OK;
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    FD_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    std::vector<real_t> tmp((size_t)p->C);
    FD_HIP_CHECK(hipMemcpy(tmp.data(), p->d_eps, sizeof(real_t) * (size_t)p->C, hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < p->C; ++c) eps_out[c] = (double)tmp[(size_t)c];
    return FD_OK;
}

#ifndef FDJAC_F32   /* colouring and the copy probe do not depend on the element type */
int fd_color_banded(int64_t N, int64_t l, int64_t u, int64_t *colorvec_out, int64_t *ncolors_out)
{
    FD_REQUIRE(colorvec_out && N >= 1 && l + u + 1 >= 1, FD_ERR_ARG, "bad argument");
    const int64_t w = l + u + 1;
    for (int64_t j = 0; j < N; ++j) colorvec_out[j] = j % w + 1;
    if (ncolors_out) *ncolors_out = std::min<int64_t>(w, N);
    return FD_OK;
}

int fd_color_columns_greedy(int64_t M, int64_t N, const void *colptr, const void *rowval, int idx_bytes, int idx_base,
                            int64_t *colorvec_out, int64_t *ncolors_out)
{
    FD_REQUIRE(colptr && rowval && colorvec_out, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(idx_bytes == 4 || idx_bytes == 8, FD_ERR_ARG, "idx_bytes must be 4 or 8");
    FD_REQUIRE(M >= 0 && N >= 1, FD_ERR_SHAPE, "bad shape");
    const int64_t nnz = load_idx(colptr, idx_bytes, N) - idx_base;
    FD_REQUIRE(nnz >= 0, FD_ERR_SHAPE, "colptr is not monotone");
    // CSR view: for every row the columns that touch it
    std::vector<int64_t> rptr((size_t)M + 1, 0);
    for (int64_t q = 0; q < nnz; ++q) {
        const int64_t r = load_idx(rowval, idx_bytes, q) - idx_base;
        FD_REQUIRE(r >= 0 && r < M, FD_ERR_SHAPE, "rowval[%lld] outside 1..%lld", (long long)q, (long long)M);
        rptr[(size_t)r + 1]++;
    }
    for (int64_t r = 0; r < M; ++r) rptr[(size_t)r + 1] += rptr[(size_t)r];
    std::vector<int32_t> rcols((size_t)nnz);
    {
        std::vector<int64_t> fill(rptr.begin(), rptr.end() - 1);
        for (int64_t j = 0; j < N; ++j) {
            const int64_t a = load_idx(colptr, idx_bytes, j) - idx_base, b = load_idx(colptr, idx_bytes, j + 1) - idx_base;
            FD_REQUIRE(a >= 0 && a <= b && b <= nnz, FD_ERR_SHAPE, "colptr is not monotone at column %lld", (long long)j);
            for (int64_t q = a; q < b; ++q) rcols[(size_t)fill[(size_t)(load_idx(rowval, idx_bytes, q) - idx_base)]++] = (int32_t)j;
        }
    }
    std::vector<int64_t> stamp;  // stamp[c] == j+1  <=> colour c is forbidden for column j
    int64_t C = 0;
    for (int64_t j = 0; j < N; ++j) {
        const int64_t a = load_idx(colptr, idx_bytes, j) - idx_base, b = load_idx(colptr, idx_bytes, j + 1) - idx_base;
        for (int64_t q = a; q < b; ++q) {
            const int64_t r = load_idx(rowval, idx_bytes, q) - idx_base;
            for (int64_t t = rptr[(size_t)r]; t < rptr[(size_t)r + 1]; ++t) {
                const int32_t k = rcols[(size_t)t];
                if (k < j) {  // already coloured neighbour
                    const int64_t ck = colorvec_out[k];
                    if ((int64_t)stamp.size() <= ck) stamp.resize((size_t)ck + 1, 0);
                    stamp[(size_t)ck] = j + 1;
                }
            }
        }
        int64_t c = 1;
        while (c < (int64_t)stamp.size() && stamp[(size_t)c] == j + 1) ++c;
        colorvec_out[j] = c;
        if (c > C) C = c;
    }
    if (ncolors_out) *ncolors_out = C;
    return FD_OK;
}

int fd_stream_copy_gbps(fd_ctx *ctx, int64_t bytes, int iters, double *gbps_out)
{
    FD_REQUIRE(ctx && gbps_out && bytes >= 16 && iters >= 1, FD_ERR_ARG, "bad argument");
    FD_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t n16 = bytes / 16;
    void *a = nullptr, *b = nullptr;
    FD_HIP_CHECK(hipMalloc(&a, (size_t)n16 * 16));
    if (hipMalloc(&b, (size_t)n16 * 16) != hipSuccess) {
        (void)hipFree(a);
        set_error("hipMalloc failed");
        return FD_ERR_NOMEM;
    }
    (void)hipMemsetAsync(a, 1, (size_t)n16 * 16, ctx->stream);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch_stream_copy(ctx, a, b, n16);  // warm-up
    (void)hipEventRecord(e0, ctx->stream);
    for (int i = 0; i < iters; ++i) launch_stream_copy(ctx, a, b, n16);
    (void)hipEventRecord(e1, ctx->stream);
    hipError_t se = hipStreamSynchronize(ctx->stream);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    FD_REQUIRE(se == hipSuccess && ms > 0, FD_ERR_HIP, "stream copy probe failed: %s", hipGetErrorString(se));
    *gbps_out = 2.0 * (double)n16 * 16.0 * iters / (ms * 1e-3) / 1e9;
    return FD_OK;
}

#endif

}  // extern "C"
