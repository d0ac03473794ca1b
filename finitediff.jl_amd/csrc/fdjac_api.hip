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
#include <string>

#include "fdjac_internal.h"
#include "fdjac_eps_dev.h"

namespace fdjac {

#ifndef FDJAC_F32
static thread_local char g_err[512] = "";
#endif

int launch_eps(fd_plan *p, const real_t *x, double relstep, double absstep, double dir);
int launch_eps_groups(fd_plan *p, const real_t *x, int g0, int ng, bool final, double relstep, double absstep, double dir);
int launch_eps_final(fd_plan *p, double relstep, double absstep, double dir);
int launch_eps_flags(fd_plan *p, const real_t *x, const FusedEps &fz);
constexpr int kMaxEpsShards = kEpsGroups;   // a shard of the reduction is a whole number of groups
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
constexpr int32_t kPlanKnownFlags = FD_PLAN_EPS_CONTIGUOUS | FD_PLAN_COMPLEX_X | FD_PLAN_FINGERPRINT | FD_PLAN_STORE_CSC | FD_PLAN_STORE_CSC_ALWAYS | FD_PLAN_STORE_CSC_ROWS;

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
    auto env_int = [](const char *name, int dflt) { const char *v = fdjac::test_switch(name); return (v && *v) ? atoi(v) : dflt; };
    p->small_ok = env_int("FDJAC_SMALL", 1) != 0;
    // non-temporal loads of x in the step-size reduction: right when 240 MB of plain nzval stores are still draining (the
    // hand-over path, round 2); WRONG when f!'s storing launch follows (round 3): that launch re-reads x, which the reduction's
    // plain loads leave in the 256 MiB Infinity Cache -- N = 10^7: 75 instead of 82 us per Jacobian (profiles/r03_c_*).  A call
    // decides by which path it takes.
    p->cx = t_lowered_cx;                                // (inside a LoweredScope only)
    if (p->cx) p->small_ok = false;                      // the fused small-problem launch has ONE colour rule for norm and perturbation
    p->eps_nt = true;
    { const char *v = fdjac::test_switch("FDJAC_FUSED_MAX_N"); if (v && *v) p->fz_max_n = atoll(v); }
    p->fz_shared_ok = env_int("FDJAC_FUSED_SHARED", 0) != 0;
    p->fz_flags_ok = env_int("FDJAC_EPS_FLAGS", 1) != 0;
    p->lazy_diff = env_int("FDJAC_LAZY_DIFF", 1) != 0;
    p->bd_allowed = env_int("FDJAC_BAND_DESC", 1) != 0;
    // a FD_LAZY_CAP_STORE launcher stores the Jacobian of a verified exact band itself (include/fdjac_device.h): default since
    // round 3 (N = 10^7 tridiagonal: 0.18 -> 0.09 ms per Jacobian, bit-identical); FDJAC_LAZY_STORE=0 keeps the hand-over
    p->store_allowed = env_int("FDJAC_LAZY_STORE", 1) != 0;
    p->want_store_csc = (opts->flags & FD_PLAN_STORE_CSC) != 0;
    p->want_store_rows = (opts->flags & FD_PLAN_STORE_CSC_ROWS) != 0;
    p->store_csc_always = (opts->flags & FD_PLAN_STORE_CSC_ALWAYS) != 0;   // a compact device copy of the pattern for column-centric storing launches
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
    // (the batched f! values d_FX, the materialised points d_X and the staging copies of x / f_in are allocated on first use --
    //  ensure_values / ensure_points / ensure_stage: a storing launch needs none of them (config 5: 0.5 GB less, and a plan that
    //  is ready milliseconds earlier), a lazy-point launcher with device inputs no points)
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
                const char *fc = fdjac::test_switch("FDJAC_EPS_CYCLIC");
                bool cyc = !(fc && *fc && atoi(fc) == 0) && p->N >= 1 && col0[0] >= 0;
                const int32_t sh = cyc ? col0[0] : 0;
                for (int64_t j = 0; j < p->N && cyc; ++j) cyc = col0[(size_t)j] == (int32_t)((j + sh) % p->C);
                p->cyc_C = cyc ? (int)p->C : 0;
                p->cyc_shift = cyc ? (int)sh : 0;
            }
            // the two-level grid (k_eps_partial_reg): 64 groups of tpg tiles, each summed by bpg <= 16 blocks of tpb tiles -- about
            // 4 workgroups per CU at large N (measured best in round 2: fewer partials, enough loads in flight).  A function of N
            // alone, not of the device: every rank of a sharded reduction must cut the same groups (fd_plan_set_comm verifies it)
            const int64_t tiles = (p->N + 2047) / 2048;  // 4 x 512 elements per block round
            const int64_t tpg = (tiles + kEpsGroups - 1) / kEpsGroups;
            const int64_t tpb = (tpg + kEpsBlocksPerGroup - 1) / kEpsBlocksPerGroup;
            FD_REQUIRE(tpg < ((int64_t)1 << 24), FD_ERR_UNSUPPORTED, "N too large for the step-size reduction's grid");
            p->eps_tpg = (int)tpg;
            p->eps_tpb = (int)tpb;
            p->eps_bpg = (int)((tpg + tpb - 1) / tpb);
            p->n_partial_blocks = kEpsGroups * p->eps_bpg;
            p->partial_cap = (int64_t)p->n_partial_blocks * kRegColors;
            if ((rc = dev_alloc(&p->d_partial, p->partial_cap))) return rc;
            if ((rc = dev_alloc(&p->d_gsum, (int64_t)2 * kEpsGroups * kRegColors))) return rc;
            if ((rc = dev_alloc(&p->d_tick, (int64_t)kEpsGroups + 1))) return rc;
            FD_HIP_CHECK(hipMemsetAsync(p->d_gsum, 0, sizeof(double) * 2 * kEpsGroups * kRegColors, p->ctx->stream));
            FD_HIP_CHECK(hipMemsetAsync(p->d_tick, 0, sizeof(unsigned) * (kEpsGroups + 1), p->ctx->stream));
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

static int ensure_values(fd_plan *p)
{
    if (p->d_FX) return FD_OK;
    // (+1 row: small problems evaluate f(x) as one more member of the perturbed batch)
    return dev_alloc(&p->d_FX, (p->chunkB * p->pts + 1) * p->cplx * p->ldf);
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

#include "fdjac_plan_list.hip"

}  // namespace fdjac

#include "fdjac_planbuild.hip"
#include "fdjac_planbuild_lists.hip"
#include "fdjac_store_csc.hip"

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
    if (ctx->check_stream) { (void)hipStreamSynchronize(ctx->check_stream); (void)hipStreamDestroy(ctx->check_stream); }
    if (ctx->check_event) (void)hipEventDestroy(ctx->check_event);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->h_stale) (void)hipHostFree(ctx->h_stale);
    delete ctx;
    return FD_OK;
}

void *fd_ctx_stream(fd_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int fd_ctx_synchronize(fd_ctx *ctx)
{
    FD_REQUIRE(ctx != nullptr, FD_ERR_ARG, "ctx is NULL");
    FD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->check_stream) FD_HIP_CHECK(hipStreamSynchronize(ctx->check_stream));      // (the deferred content checks run beside the main stream)
    if (ctx->h_stale && *(volatile int *)ctx->h_stale) {       // a deferred content check (fd_plan_matches_async) found a stale plan
        *ctx->h_stale = 0;
        set_error("a deferred content check found that a plan no longer matches the caller's pattern / colour arrays (fd_plan_stale tells "
                  "which): results enqueued since that check are those of the old arrays");
        return FD_ERR_STALE;
    }
    return FD_OK;
}

#endif

int fd_plan_destroy(fd_plan *p)
{
    if (!p) return FD_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    if (p->ctx->check_stream && p->d_fpx) (void)hipStreamSynchronize(p->ctx->check_stream);      // (a deferred check of this plan may still be reading d_fpx)
    void *ptrs[] = {p->d_color, p->d_rowval, p->d_nzcolor, p->d_dest, p->d_spos, p->d_wtiles, p->d_wcode, p->d_w2desc, p->d_cr_rlo, p->d_cr_cnt, p->d_cr_off,
                    p->d_perm, p->d_cptr, p->d_X, p->d_FX, p->d_fx, p->d_eps, p->d_partial, p->d_gsum, p->d_tick, p->d_fpx, p->d_xstage,
                    p->d_finstage, p->d_outstage[0], p->d_outstage[1], p->d_outstage[2], p->d_zero_own, p->d_eps2, p->d_tile_order, p->d_fxwin, p->d_fp, p->d_sc_colptr, p->d_sc_rowval, p->d_sc_note, p->d_sr_ptr, p->d_sr_col, p->d_sr_slot, p->d_sr_order, p->d_sr_tile, p->d_se_col, p->d_se_slot, p->d_se_info, p->d_split, p->d_bbb_off, p->d_bbb_blk, p->d_bbb_start, p->d_bbb_stride};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    if (p->h_pstale) (void)hipHostFree(p->h_pstale);
    if (p->h_fz_err) (void)hipHostFree(p->h_fz_err);
    if (p->d_fz_part) (void)hipFree(p->d_fz_part);
    if (p->d_fz_eps) (void)hipFree(p->d_fz_eps);
    if (p->d_fz_trace) (void)hipFree(p->d_fz_trace);
    for (auto &sp : p->spans) {
        p->event_pool.push_back(sp.a);
        p->event_pool.push_back(sp.b);
    }
    for (hipEvent_t ev : p->event_pool) (void)hipEventDestroy(ev);
    delete p;
    return FD_OK;
}

#include "fdjac_plan_create.hip"

int fd_plan_row_lists(const fd_plan *p, const void **row_ptr_dev, const void **row_col_dev, const void **row_slot_dev, int64_t *entries, uint64_t *plan_serial)
{
    FD_REQUIRE(p != nullptr, FD_ERR_ARG, "plan is NULL");
    FD_REQUIRE(p->store_csc_ok && p->d_sr_ptr != nullptr, FD_ERR_UNSUPPORTED,
               "this plan keeps no row lists: create it with FD_PLAN_STORE_CSC | FD_PLAN_STORE_CSC_ROWS (a SparseMatrixCSC common-pattern plan that holds every column)");
    if (row_ptr_dev) *row_ptr_dev = p->d_sr_ptr;
    if (row_col_dev) *row_col_dev = p->d_sr_col;
    if (row_slot_dev) *row_slot_dev = p->d_sr_slot;
    if (entries) *entries = p->sc_entries;
    if (plan_serial) *plan_serial = p->sc_serial;
    return FD_OK;
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
    case FD_INFO_EPS_NT: *value = !(store_active(p) || store_csc_active(p)) ? 1 : 0; break;
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
        if (p->timing == 1 && p->timing_stride > 1 && ((p->timing_calls - 1) % p->timing_stride) != 0) return;      // (sampled: see fd_plan_set_timing_stride)
        if (p->timing == 4 && stage == FD_STAGE_TOTAL) return;       // level 4: every stage, not the whole call (its markers would sit between the stages)
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
    p->timing = on < 0 ? 0 : (on > 4 ? 4 : on);
    for (int i = 0; i < FD_NSTAGES; ++i) {
        p->ms_sum[i] = 0;
        p->launches[i] = 0;
        p->samples[i].clear();
    }
    return FD_OK;
}

int fd_plan_set_timing_stride(fd_plan *p, int stride)
{
    FD_REQUIRE(p && stride >= 1, FD_ERR_ARG, "bad argument");
    p->timing_stride = stride;
    p->timing_calls = 0;
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

// which shard of the sharded reduction this rank runs (a bare mailbox, or the communicator's ranks)
static void eps_shard_of(const fd_plan *p, int *W, int *r)
{
    if (p->p2p) { *W = fdjac_p2p_nranks(p->p2p); *r = fdjac_p2p_rank(p->p2p); }
    else { *W = fdjac_comm_nranks(p->comm); *r = fdjac_comm_rank(p->comm); }
}

// the per-step exchange of a sharded call: every rank's group sums (+ the halo of a sharded x), then level 2 of the reduction
static int eps_exchange(fd_plan *p, real_t *x_dev, int S, double relstep, double absstep, double dir)
{
    fd_p2p *mb = p->p2p ? p->p2p : (p->comm ? fdjac_comm_p2p(p->comm) : nullptr);
    if (mb && fdjac_p2p_nranks(mb) > 1) {
        fdjac_eps_final fin;
        fin.gsum = p->d_gsum; fin.ldp = kRegColors; fin.ngroups = kEpsGroups; fin.C = (int)p->C;
        fin.relstep = relstep; fin.absstep = absstep; fin.dir = dir; fin.is_forward = p->fdtype == FD_FORWARD ? 1 : 0;
        fin.eps = p->d_eps; fin.eps2 = p->d_eps2; fin.elem_bytes = (int)sizeof(real_t);
        const int rc = fdjac_p2p_step(mb, p->halo > 0 ? (void *)x_dev : nullptr, p->halo_own0, p->halo_own1, p->halo, (int)sizeof(real_t),
                                      p->d_gsum, (int64_t)S * kRegColors * (int64_t)sizeof(double), &fin);
        if (rc != FD_ERR_UNSUPPORTED) {        // (UNSUPPORTED: the payload does not fit the mailbox slot -- RCCL below)
            if (!rc) p->eps2_fresh = p->d_eps2 != nullptr;
            return rc;
        }
    }
    int rc = FD_OK;
    if (p->comm) {
        if (p->halo > 0) rc = fd_comm_halo_exchange(p->comm, x_dev, p->halo_own0, p->halo_own1, p->halo, (int)sizeof(real_t));
        if (!rc) rc = fdjac_comm_allgather_f64(p->comm, p->d_gsum, (int64_t)S * kRegColors);
    }
    if (!rc) rc = launch_eps_final(p, relstep, absstep, dir);
    return rc;
}

// The plain f! launcher.  For a plan of the lowered complex-valued-x problem the arrays hold (re, im) pairs: the launcher is
// called as the header promises for complex elements -- is_complex = 1, strides and rows counted in complex elements.
static inline int call_f(const fd_plan *p, fd_f_launch f, void *fctx, void *fx, const void *x, int64_t nbatch, int64_t x_stride,
                         int64_t fx_stride, int64_t row_begin, int64_t row_end, int is_complex, void *stream)
{
    set_error("%s", "");      // (a launcher of this library says WHY it failed in the error text: FD_F_CHECK passes it on)
    if (!p->cx) return f(fctx, fx, x, nbatch, x_stride, fx_stride, row_begin, row_end, is_complex, stream);
    return f(fctx, fx, x, nbatch, x_stride / 2, fx_stride / 2, row_begin / 2, (row_end + 1) / 2, 1, stream);
}
// a failed f! launch: its return code and, if the launcher left one (the runtime-compiled functors do), its own message
#define FD_F_CHECK(rc)                                                                                                      \
    do {                                                                                                                    \
        if ((rc) != 0) {                                                                                                    \
            const std::string _why = fd_last_error();                                                                       \
            set_error("f! launcher returned %d%s%s", (int)(rc), _why.empty() ? "" : ": ", _why.c_str());                    \
            return FD_ERR_CALLBACK;                                                                                         \
        }                                                                                                                   \
    } while (0)

// the step sizes of a sharded call in separate launches (what a call does when its storing launch cannot carry the reduction):
// this rank's GROUPS of the two-level sum (the part of x it owns), then ONE exchange -- the group sums of every rank (64 / W x 8
// doubles each) and, for a sharded x (fd_plan_set_halo), the halo of x ride in the same launch, whose last workgroup adds the 64 group
// sums in order and writes the step sizes (mailbox: fdjac_p2p_step); without a mailbox: halo by RCCL send / recv, in-place all-gather of
// the group sums, k_eps_final.  Same bits as the unsharded call on every rank.
static int eps_sharded_classic(fd_plan *p, const real_t *x_dev, double relstep, double absstep, double dir)
{
    int W, r;
    eps_shard_of(p, &W, &r);
    const int S = (kEpsGroups + W - 1) / W;
    const int g0 = std::min(r * S, kEpsGroups), ng = std::min(S, kEpsGroups - g0);
    int rc = launch_eps_groups(p, x_dev, g0, ng, false, relstep, absstep, dir);
    if (!rc) {
        Span se(p, FD_STAGE_EXCHANGE);
        rc = eps_exchange(p, const_cast<real_t *>(x_dev), S, relstep, absstep, dir);
    }
    return rc;
}

// ---- the fused step: buffers (allocated on first use; every slot starts as the sentinel) ----
static int fused_reset(fd_plan *p)
{
    const size_t np = 2 * (size_t)p->n_partial_blocks * kRegColors, ne = 2 * (size_t)kFzReplicas * kFzPitch;
    std::vector<unsigned long long> hp(np, kFzSentinel64);
    std::vector<rbits_t> he(ne, FzBits<real_t>::sentinel);
    FD_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    FD_HIP_CHECK(hipMemcpy(p->d_fz_part, hp.data(), np * 8, hipMemcpyHostToDevice));
    FD_HIP_CHECK(hipMemcpy(p->d_fz_eps, he.data(), ne * sizeof(rbits_t), hipMemcpyHostToDevice));
    FD_HIP_CHECK(hipDeviceSynchronize());
    *p->h_fz_err = 0;
    p->fz_parity = 0;
    return FD_OK;
}
static int ensure_fused(fd_plan *p)
{
    if (p->d_fz_part) return FD_OK;
    FD_HIP_CHECK(hipHostMalloc((void **)&p->h_fz_err, sizeof(int), hipHostMallocMapped));
    *p->h_fz_err = 0;
    FD_HIP_CHECK(hipHostGetDevicePointer((void **)&p->d_fz_err, p->h_fz_err, 0));
    FD_HIP_CHECK(hipMalloc((void **)&p->d_fz_eps, 2 * (size_t)kFzReplicas * kFzPitch * sizeof(rbits_t)));
    FD_HIP_CHECK(hipMalloc((void **)&p->d_fz_part, 2 * (size_t)p->n_partial_blocks * kRegColors * sizeof(double)));
    { const char *v = fdjac::test_switch("FDJAC_FUSED_TRACE"); if (v && atoi(v) != 0) FD_HIP_CHECK(hipMalloc((void **)&p->d_fz_trace, 16 * sizeof(long long))); }
    return fused_reset(p);
}
static int64_t fused_timeout_ticks()
{
    // wall_clock64() counts at 100 MHz on gfx9; FDJAC_P2P_TIMEOUT_MS bounds every device-side wait (default 2 s)
    const char *v = getenv("FDJAC_P2P_TIMEOUT_MS");
    const int64_t ms = (v && *v) ? atoll(v) : 2000;
    return (ms < 1 ? 1 : ms) * 100000;
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
    FD_REQUIRE(!(p->h_pstale && *(volatile int *)p->h_pstale), FD_ERR_STALE,
               "a deferred content check (fd_plan_matches_async) found that this plan no longer matches the caller's pattern / colour arrays");
    if (p->h_fz_err && *(volatile int *)p->h_fz_err) {
        const int what = *(volatile int *)p->h_fz_err;
        (void)fused_reset(p);
        FD_REQUIRE(false, FD_ERR_COMM, "the previous fused step of this plan timed out (%s never arrived; FDJAC_P2P_TIMEOUT_MS): its "
                   "output holds NaNs", what == 1 ? "a block sum of the step-size reduction" : "the step sizes");
    }
    {
        fd_p2p *mbq = p->p2p ? p->p2p : (p->comm ? fdjac_comm_p2p(p->comm) : nullptr);
        const int st = mbq ? fdjac_p2p_failed(mbq) : 0;
        FD_REQUIRE(st == 0, FD_ERR_COMM, "an earlier exchange through this plan's mailbox timed out (code %d: a peer's data never arrived; "
                   "fd_p2p_status / fd_comm_p2p_status; FDJAC_P2P_TIMEOUT_MS): step sizes and halo of that call are not to be trusted", st);
    }
    ++p->timing_calls;
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
    if (small_points) { int rc = ensure_points(p); if (!rc) rc = ensure_values(p); if (rc) return rc; }

    // a launcher that hands over central differences: the doubled step sizes come out of the finalize launch
    if (p->fdtype == FD_CENTRAL && p->lazy_fn && (p->lazy_caps & FD_LAZY_CAP_DIFF) && p->lazy_diff && p->kind != K_DENSE) {
        const int rc = ensure_diff_scratch(p);
        if (rc) return rc;
    }
    // the fused step: the storing launch of a FD_LAZY_CAP_FUSED_EPS launcher runs the reduction itself (one launch per Jacobian) -- on one
    // GPU up to fz_max_n columns; in a sharded call (mailbox attached) always: its finishers exchange the group sums and the halo too
    fd_p2p *fz_mb = p->p2p ? p->p2p : (p->comm ? fdjac_comm_p2p(p->comm) : nullptr);
    const bool shard_ctx = p->comm != nullptr || p->p2p != nullptr;
    bool fuse = !small && p->fdtype != FD_COMPLEX && p->C > 0 && p->C <= kRegColors && p->cyc_C > 0 && p->eps_mode == FD_EPS_COMPUTE &&
                p->lazy_fn && (p->lazy_caps & FD_LAZY_CAP_FUSED_EPS) && store_active(p) && p->store_ok &&
                (p->kind == K_CSC || p->kind == K_BANDED || p->kind == K_TRIDIAG) && p->nchunks == 1 && full_colors &&
                !(p->fdtype == FD_FORWARD && fin_dev) && p->d_partial != nullptr;
    if (shard_ctx && p->halo > 0) {      // (fd_plan_set_halo may run before the mailbox is attached: the rank is known only now)
        int W = 1, r = 0;
        eps_shard_of(p, &W, &r);
        FD_REQUIRE(r == 0 || p->halo_own0 >= p->halo, FD_ERR_ARG, "no room for the lower halo: x[%lld - %lld, ...) starts before 0", (long long)p->halo_own0, (long long)p->halo);
        FD_REQUIRE(r + 1 >= W || p->halo_own1 + p->halo <= p->N, FD_ERR_ARG, "no room for the upper halo: x[..., %lld + %lld) ends behind N = %lld",
                   (long long)p->halo_own1, (long long)p->halo, (long long)p->N);
    }
    if (fuse) {
        // (a call being CAPTURED into a HIP graph is replayed with the arguments of the capture: the fused step's buffer parity / mailbox
        //  epoch are host state that must advance per launch -- such a call takes the separate launches, which carry no such state)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) fuse = false;
    }
    bool fuse_sharded = false;
    int fzW = 1, fzr = 0;
    if (fuse && shard_ctx) {
        if (fz_mb) { fzW = fdjac_p2p_nranks(fz_mb); fzr = fdjac_p2p_rank(fz_mb); }
        // the halo cells are polled by the lanes that own the first / last pair (quad) of the rank's columns
        const bool halo_ok = p->halo == 0 || (p->halo == 2 && p->halo_own0 == p->col0 && p->halo_own1 == p->col1 && p->halo_own0 % 4 == 0 &&
                                              (p->halo_own1 % 4 == 0 || fzr == fzW - 1));
        // (ranks sharing ONE device -- tests, dry runs: a launch full of wavefronts that wait for a peer would keep the peer from running)
        // (and only for shards up to fz_max_n columns -- measured with the loop-back rank share at N = 10^7: 8 ranks 18.3 us fused against
        //  22.4 in three launches, 4 ranks 28.7 / 30.2, 2 ranks 82 / 45: a launch whose storing wavefronts do not all fit on the device
        //  keeps the rest queued behind wavefronts that wait)
        fuse = fuse_sharded = fz_mb != nullptr && fzW > 1 && p->fz_sharded_ok && eps_shardable(p) && halo_ok &&
                              (int64_t)((kEpsGroups + fzW - 1) / fzW) * p->eps_tpg * 2048 <= p->fz_max_n &&      // (the LARGEST shard: the same verdict on every rank)
                              (!fdjac_p2p_shared_device(fz_mb) || p->fz_shared_ok);
    } else if (fuse) {
        fuse = p->N <= p->fz_max_n;
    }
    // the reduction as its own launch takes the fused step's hand-offs too (k_eps_flags: finishers instead of tickets) -- one GPU, <= 8 colours
    bool eps_flags = !fuse && p->fz_flags_ok && !small && !shard_ctx && p->fdtype != FD_COMPLEX && p->C > 0 && p->C <= kRegColors &&
                     p->kind != K_DENSE && p->eps_mode == FD_EPS_COMPUTE && p->d_partial != nullptr &&
                     p->n_partial_blocks == kEpsGroups * p->eps_bpg && p->n_partial_blocks <= kFzMaxBlocks;
    if (eps_flags) {      // (a captured call is replayed with the capture's buffer parity: the ticket form carries no host state)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) eps_flags = false;
    }
    FusedEps fz_job;
    if (fuse || eps_flags) {
        const int rc = ensure_fused(p);
        if (rc) return rc;
        memset(&fz_job, 0, sizeof fz_job);
        fz_job.eg.tpg = p->eps_tpg; fz_job.eg.bpg = p->eps_bpg; fz_job.eg.tpb = p->eps_tpb; fz_job.eg.final_groups = kEpsGroups;
        fz_job.eg.C = (int)p->C; fz_job.eg.is_forward = p->fdtype == FD_FORWARD ? 1 : 0;
        fz_job.eg.relstep = relstep; fz_job.eg.absstep = absstep; fz_job.eg.dir = dir;
        fz_job.nranks = 1; fz_job.rank = 0; fz_job.g0 = 0; fz_job.ng = kEpsGroups;
        if (fuse_sharded) {
            const int S = (kEpsGroups + fzW - 1) / fzW;
            fz_job.nranks = fzW; fz_job.rank = fzr;
            fz_job.g0 = std::min(fzr * S, kEpsGroups); fz_job.ng = std::min(S, kEpsGroups - fz_job.g0);
            fdjac_p2p_fused mbv;
            const int rm = fdjac_p2p_fused_begin(fz_mb, &mbv);
            if (rm) return rm;
            fz_job.peer = mbv.peer; fz_job.local = mbv.local; fz_job.fz_off = mbv.fz_off; fz_job.buf = mbv.buf; fz_job.buf_reset = mbv.buf_reset;
            if (p->halo > 0) { fz_job.xw = const_cast<real_t *>(x_dev); fz_job.own_begin = p->halo_own0; fz_job.own_end = p->halo_own1; fz_job.halo = (int)p->halo; }
        }
        fz_job.nblocks = fz_job.ng * p->eps_bpg;
        fz_job.cyc_C = p->cyc_C; fz_job.cyc_shift = p->cyc_shift; fz_job.pair = p->cx ? 1 : 0;
        const size_t hp = (size_t)p->n_partial_blocks * kRegColors, he = (size_t)kFzReplicas * kFzPitch;
        fz_job.part = p->d_fz_part + (p->fz_parity ? hp : 0);
        fz_job.part_next = p->d_fz_part + (p->fz_parity ? 0 : hp);
        fz_job.epsr = (rbits_t *)p->d_fz_eps + (p->fz_parity ? he : 0);
        fz_job.epsr_next = (rbits_t *)p->d_fz_eps + (p->fz_parity ? 0 : he);
        fz_job.eps = p->d_eps; fz_job.eps2 = p->d_eps2;
        fz_job.err = fuse_sharded ? fdjac_p2p_err_word(fz_mb) : p->d_fz_err;
        fz_job.timeout_ticks = fused_timeout_ticks();
        fz_job.trace = p->d_fz_trace;
        if (p->d_fz_trace) {      // (diagnostic runs only: minima start at all-ones, maxima at zero)
            static const long long init[16] = {-1, 0, 0, 0, 0, -1, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            FD_HIP_CHECK(hipMemcpyAsync(p->d_fz_trace, init, sizeof init, hipMemcpyHostToDevice, s));
        }
    }
    if (p->eps_mode != FD_EPS_PRECOMPUTED) p->eps2_fresh = false;
    p->eps_nt = !(store_active(p) || store_csc_active(p));   // (see apply_opts)
    // step sizes for every colour (one pass over x), src/jacobians.jl:559-561 / 600-602
    if (p->fdtype != FD_COMPLEX && p->C > 0 && p->eps_mode == FD_EPS_PRECOMPUTED) {
        // the caller ran fd_plan_eps_partials / exchanged / fd_plan_eps_finalize: p->d_eps is current
        FD_REQUIRE(!small_points, FD_ERR_UNSUPPORTED, "FD_EPS_PRECOMPUTED needs a plan whose reduction can be sharded");
    } else if (fuse) {
        // (nothing here: the storing launch below computes the step sizes)
    } else if (eps_flags) {
        Span sp(p, FD_STAGE_EPS);
        const int rc = launch_eps_flags(p, x_dev, fz_job);
        if (rc) return rc;
        p->fz_parity ^= 1u;
    } else if (p->fdtype != FD_COMPLEX && p->C > 0) {
        Span sp(p, FD_STAGE_EPS);
        int rc;
        if (small) {
            rc = launch_eps_perturb_small(p, x_dev, relstep, absstep, dir, small_points ? p->fdtype : -1,
                                          base_in_batch ? (int)(p->C * p->pts) : -1);
        } else if ((p->comm || p->p2p) && eps_shardable(p)) {   // (also with a single-rank communicator: same code path)
            rc = eps_sharded_classic(p, x_dev, relstep, absstep, dir);
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
            FD_F_CHECK(rc);
            p->fcalls_last += 1;
            fx = p->d_fx;
        }
    }

    // dense J / list kinds with several chunks or uncovered entries start from zero (fill_matrix!,
    // src/jacobians.jl:530-532).  Kinds that write every stored value in one chunk skip the fill.
    if (p->kind == K_CSC_DENSE || p->kind == K_COO_DENSE || (p->kind == K_BBB && p->bbb_fill)) {
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
        if (p->kind == K_BBB && store_active(p) && !(p->fdtype == FD_FORWARD && !base_pending)) {
            // BandedBlockBandedMatrix storage: the launcher fills every slab itself (fd_bbb_store)
            Span sp(p, FD_STAGE_DECOMPRESS);
            fd_bbb_store bb;
            memset(&bb, 0, sizeof bb);
            bb.out = outs[0]; bb.N = p->N; bb.nblk = p->bbb_nb; bb.block_size = p->bbb_bs;
            bb.bl = p->bbb_bl; bb.bu = p->bbb_bu; bb.lam = p->bbb_lam; bb.mu = p->bbb_mu;
            bb.start = (const long long *)p->d_bbb_start; bb.stride = (const long long *)p->d_bbb_stride;
            bb.color = p->d_color; bb.color_bytes = p->color8 ? 1 : 4; bb.C = (int)p->C; bb.elem_bytes = (int)sizeof(real_t);
            fd_lazy_points lp = {};
            lp.x = x_dev; lp.color = p->d_color; lp.eps = p->d_eps; lp.color_bytes = bb.color_bytes;
            lp.c_lo = c_lo; lp.ncolors = B; lp.pts = p->pts; lp.nparts = 1;
            lp.diff = (p->fdtype == FD_FORWARD && !diff_base_counted) ? 2 : 1;
            lp.store = &bb; lp.store_kind = FD_STORE_BBB;
            const int rc = p->lazy_fn(fctx, p->d_FX, &lp, p->ldf, p->row0, p->row1, (void *)s);
            FD_REQUIRE(rc == 0 || rc == FD_LAZY_DECLINED, FD_ERR_CALLBACK, "lazy f! launcher (BBB store) returned %d", rc);
            if (rc == 0) {
                p->fcalls_last += (int64_t)B * p->pts + (lp.diff == 2 ? 1 : 0);
                diff_base_counted = true;
                continue;
            }
        } else if (store_active(p) && !(p->fdtype == FD_FORWARD && !base_pending)) {
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
            lp.eps_job = fuse ? &fz_job : nullptr;
            int rc = p->lazy_fn(fctx, p->d_FX, &lp, p->ldf, p->row0, p->row1, (void *)s);
            if (fuse && rc == 0) {
                if (fuse_sharded) fdjac_p2p_fused_commit(fz_mb);
                p->fz_parity ^= 1u;
                p->eps2_fresh = p->d_eps2 != nullptr;
            } else if (fuse) {      // (declined with the reduction attached: the library's own launch(es), then the plain storing launch)
                fuse = false;
                sp.stop();
                { Span se(p, FD_STAGE_EPS); const int re = fuse_sharded ? eps_sharded_classic(p, x_dev, relstep, absstep, dir) : launch_eps(p, x_dev, relstep, absstep, dir); if (re) return re; }
                Span sp2(p, FD_STAGE_DECOMPRESS);
                lp.eps_job = nullptr;
                if (rc == FD_LAZY_DECLINED) rc = p->lazy_fn(fctx, p->d_FX, &lp, p->ldf, p->row0, p->row1, (void *)s);
            }
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
            // (a launcher with FD_LAZY_CAP_STORE_CSC_BASE evaluates the unperturbed rows itself: no plain evaluation, fx_base = NULL)
            const bool own_base = base_pending && (p->lazy_caps & FD_LAZY_CAP_STORE_CSC_BASE) != 0;
            if (base_pending && !own_base) {
                Span sp(p, FD_STAGE_F);
                const int rc = call_f(p, f, fctx, p->d_fx, x_dev, 1, p->N, p->ldf, p->row0, p->row1, 0, (void *)s);
                FD_F_CHECK(rc);
                p->fcalls_last += 1;
                base_pending = false;
            }
            Span sp(p, FD_STAGE_DECOMPRESS);
            fd_csc_store sc;
            memset(&sc, 0, sizeof sc);
            sc.out = outs[0]; sc.M = p->M; sc.N = p->N; sc.col_begin = p->col0; sc.col_end = p->col1;
            sc.colptr = p->d_sc_colptr; sc.rowval = p->d_sc_rowval; sc.note = p->d_sc_note; sc.color = p->d_color; sc.fx_base = (p->fdtype == FD_FORWARD && !own_base) ? fx : nullptr;
            sc.color_bytes = p->color8 ? 1 : 4; sc.C = (int)p->C; sc.elem_bytes = (int)sizeof(real_t); sc.valid_coloring = p->sc_valid ? 1 : 0; sc.reach = p->sc_reach; sc.plan_serial = p->sc_serial;
            sc.row_ptr = p->d_sr_ptr; sc.row_col = p->d_sr_col; sc.row_slot = p->d_sr_slot; sc.row_pack = p->d_sr_order; sc.row_tile = p->d_sr_tile; sc.ent_col = p->d_se_col; sc.ent_slot = p->d_se_slot; sc.ent_info = p->d_se_info; sc.ent_tile_max = p->se_tile_max;
            fd_lazy_points lp = {};
            lp.x = x_dev;
            lp.color = p->d_color;
            lp.eps = p->d_eps;
            lp.color_bytes = sc.color_bytes;
            lp.c_lo = c_lo;
            lp.ncolors = B;
            lp.pts = p->pts;
            lp.nparts = 1;
            lp.diff = p->fdtype == FD_COMPLEX ? 0 : 1;
            lp.is_complex = p->fdtype == FD_COMPLEX ? 1 : 0;      // (the complex step: imag(f(x + i eps e_j)) / eps stored -- FD_LAZY_CAP_STORE_CSC_COMPLEX)
            lp.imag_only = lp.is_complex;
            lp.store = &sc;
            lp.store_kind = FD_STORE_CSC;
            const int rc = p->lazy_fn(fctx, p->d_FX, &lp, p->ldf, p->row0, p->row1, (void *)s);
            FD_REQUIRE(rc == 0 || rc == FD_LAZY_DECLINED, FD_ERR_CALLBACK, "lazy f! launcher (column store) returned %d", rc);
            if (rc == 0) {
                p->fcalls_last += (int64_t)B * p->pts + ((own_base && !diff_base_counted) ? 1 : 0);      // (f(x): once, inside the launch)
                if (own_base) diff_base_counted = true;
                continue;
            }
            if (p->fdtype == FD_FORWARD && !own_base) want_diff = false;      // (declined: f(x) exists already -- plain values are handed over below)
        }
        if (fuse) {      // (the storing launch did not happen: the hand-over path needs the step sizes first)
            fuse = false;
            Span se(p, FD_STAGE_EPS);
            const int re = fuse_sharded ? eps_sharded_classic(p, x_dev, relstep, absstep, dir) : launch_eps(p, x_dev, relstep, absstep, dir);
            if (re) return re;
        }
        { const int rc = ensure_values(p); if (rc) return rc; }      // (from here on the f! values are handed over through d_FX)
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
                FD_F_CHECK(rc);
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
            FD_F_CHECK(rc);
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
    if (p->h_fz_err && *(volatile int *)p->h_fz_err) {
        (void)fused_reset(p);
        FD_REQUIRE(false, FD_ERR_COMM, "the fused step timed out (FDJAC_P2P_TIMEOUT_MS): the output holds NaNs");
    }
    return FD_OK;
}

// Colour ownership, assembled (fd_plan_opts.color_begin / color_end): a plan that owns a range of colours writes only the stored values
// of its colours' columns and leaves everything else in outs untouched -- summing the ranks' outputs is an exact assembly only if
// the untouched entries are ZERO, which is true for a fresh buffer and false from the second call on (they hold the previous sum).
// This entry point owns the whole sequence: zero-fill, the call, ONE in-place all-reduce per output (comm = NULL: the caller sums).
int fd_jacobian_owned_async(fd_plan *p, fd_comm *comm, fd_f_launch f, void *fctx, const void *x, const void *f_in, double relstep,
                            double absstep, double dir, void *const *outs)
{
    FD_REQUIRE(p && x && outs, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(comm == nullptr || fdjac_comm_ctx(comm) == p->ctx, FD_ERR_ARG, "the communicator belongs to another context");
    FD_REQUIRE(p->split_n == 0, FD_ERR_UNSUPPORTED, "colour ownership of a lowered complex-valued Tridiagonal plan");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    for (int k = 0; k < p->nouts; ++k) {
        FD_REQUIRE(outs[k] || p->out_len[k] == 0, FD_ERR_ARG, "outs[%d] is NULL", k);
        if (p->out_len[k] > 0) FD_HIP_CHECK(hipMemsetAsync(outs[k], 0, sizeof(real_t) * (size_t)p->out_len[k], p->ctx->stream));
    }
    int rc = fd_jacobian_async(p, f, fctx, x, f_in, relstep, absstep, dir, outs);
    for (int k = 0; k < p->nouts && !rc && comm; ++k)
        if (p->out_len[k] > 0) rc = fd_comm_allreduce_sum(comm, outs[k], p->out_len[k], (int)sizeof(real_t));
    return rc;
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
        // with another FDJAC_EPS_CONTIG would use other slots: mismatched all-gather counts hang, matched
        // ones finalize garbage), and every rank must agree on WHETHER its reduction is sharded at all (a rank whose plan keeps
        // the replicated reduction would skip the per-call all-gather the others enter).  One tiny all-reduce at attach time
        // settles both; it is unconditional -- every rank enters it whatever its own plan looks like.
        const bool sh = eps_shardable(p);
        const double nb = sh ? (double)p->eps_tpg : -1.0, tp = sh ? (double)p->eps_bpg * 65536.0 + (double)p->eps_tpb : -1.0;
        const double mine[4] = {nb, -nb, tp, -tp};
        double got[4] = {0, 0, 0, 0};
        const int rc = fdjac_comm_allreduce_max4(comm, mine, got);
        if (rc) return rc;
        FD_REQUIRE(got[0] == mine[0] && got[1] == mine[1] && got[2] == mine[2] && got[3] == mine[3], FD_ERR_COMM,
                   "the ranks disagree on the step-size reduction (this rank: %s, 64 groups of %d tiles in %d blocks): same N, colours, "
                   "fdtype and FDJAC_SMALL everywhere?", sh ? "sharded" : "replicated", p->eps_tpg, p->eps_bpg);
    }
    p->comm = comm;
    return FD_OK;
}

int fd_plan_set_p2p(fd_plan *p, fd_p2p *p2p)
{
    FD_REQUIRE(p != nullptr, FD_ERR_ARG, "plan is NULL");
    FD_REQUIRE(p2p == nullptr || fdjac_p2p_ctx(p2p) == p->ctx, FD_ERR_ARG, "the mailbox belongs to another context");
    FD_REQUIRE(p2p == nullptr || fdjac_p2p_nranks(p2p) <= kMaxEpsShards, FD_ERR_UNSUPPORTED, "more than %d ranks", kMaxEpsShards);
    if (p2p) {
        // as fd_plan_set_comm: every rank must cut the SAME global grid and agree on WHETHER its reduction is sharded at all, and -- new in
        // round 6 -- on whether the step is the fused ONE-launch form (a rank that ran three launches against peers that run one would
        // wait for data that never comes: its peers write other mailbox cells).  Collective: every rank attaches, in the same order.
        const bool sh = eps_shardable(p);
        double mine[8] = {sh ? (double)p->eps_tpg : -1.0, sh ? (double)p->eps_bpg * 65536.0 + (double)p->eps_tpb : -1.0, (double)p->N, (double)p->C,
                          (double)p->fdtype, (double)sizeof(real_t), p->fz_sharded_ok ? (double)p->fz_max_n : -1.0, 0.0};
        std::vector<double> all((size_t)8 * (size_t)fdjac_p2p_nranks(p2p));
        const int rc = fdjac_p2p_agree8(p2p, mine, all.data());
        if (rc) return rc;
        for (int r = 0; r < fdjac_p2p_nranks(p2p); ++r)
            for (int k = 0; k < 7; ++k)
                FD_REQUIRE(all[(size_t)(8 * r + k)] == mine[k], FD_ERR_COMM,
                           "rank %d disagrees on the step-size reduction (this rank: %s, N = %lld, %lld colours, 64 groups of %d tiles in %d blocks): "
                           "same N, colours, fdtype, element type and FDJAC_SMALL everywhere?", r, sh ? "sharded" : "replicated", (long long)p->N,
                           (long long)p->C, p->eps_tpg, p->eps_bpg);
    }
    p->p2p = p2p;
    return FD_OK;
}

int fd_plan_set_halo(fd_plan *p, int64_t own_begin, int64_t own_end, int64_t halo)
{
    FD_REQUIRE(p != nullptr, FD_ERR_ARG, "plan is NULL");
    FD_REQUIRE(halo >= 0 && own_begin >= 0 && own_end >= own_begin && own_end <= p->N, FD_ERR_ARG, "bad range [%lld,%lld) / halo %lld",
               (long long)own_begin, (long long)own_end, (long long)halo);
    FD_REQUIRE(halo == 0 || own_end - own_begin >= halo, FD_ERR_ARG, "this rank owns fewer than `halo` = %lld elements", (long long)halo);
    FD_REQUIRE(halo == 0 || (halo * (int64_t)sizeof(real_t)) % 8 == 0, FD_ERR_ARG, "the halo must be a multiple of 8 bytes");
    FD_REQUIRE(halo == 0 || eps_shardable(p), FD_ERR_UNSUPPORTED,
               "the halo rides with the sharded step-size reduction: this plan's reduction cannot be sharded (needs 1..8 colours, N > 16384, "
               "forward / central) -- exchange the halo with fd_comm_halo_exchange before the call");
    p->halo = halo;
    p->halo_own0 = own_begin;
    p->halo_own1 = own_end;
    return FD_OK;
}

int fd_plan_eps_partials(fd_plan *p, const void *x_dev, int shard, int nshards, void **partials_out,
                         int64_t *slot_doubles_out)
{
    FD_REQUIRE(p && x_dev, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(nshards >= 1 && nshards <= kMaxEpsShards && shard >= 0 && shard < nshards, FD_ERR_ARG,
               "shard %d of %d (at most %d shards)", shard, nshards, kMaxEpsShards);
    FD_REQUIRE(eps_shardable(p), FD_ERR_UNSUPPORTED,
               "this plan's step-size reduction cannot be sharded (needs 1..8 colours, N > 16384, forward / central)");
    FD_REQUIRE((((uintptr_t)x_dev) & kPairMask) == 0, FD_ERR_ARG, "x must be 16-byte aligned");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    const int S = (kEpsGroups + nshards - 1) / nshards;
    const int g0 = std::min(shard * S, kEpsGroups), ng = std::min(S, kEpsGroups - g0);
    if (partials_out) *partials_out = p->d_gsum;
    if (slot_doubles_out) *slot_doubles_out = (int64_t)S * kRegColors;
    return launch_eps_groups(p, (const real_t *)x_dev, g0, ng, false, 0.0, 0.0, 1.0);
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
    return launch_eps_final(p, relstep, absstep, dir);
}

int fd_plan_eps_shard_range(fd_plan *p, int shard, int nshards, int64_t *x_begin, int64_t *x_end)
{
    FD_REQUIRE(p && x_begin && x_end, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(nshards >= 1 && nshards <= kMaxEpsShards && shard >= 0 && shard < nshards, FD_ERR_ARG, "shard %d of %d", shard, nshards);
    FD_REQUIRE(eps_shardable(p), FD_ERR_UNSUPPORTED, "this plan's step-size reduction cannot be sharded");
    const int S = (kEpsGroups + nshards - 1) / nshards;
    const int64_t per_group = (int64_t)p->eps_tpg * 2048;                     // k_eps_partial_reg's tile: 4 x 512 elements
    const int64_t g0 = std::min<int64_t>((int64_t)shard * S, kEpsGroups), g1 = std::min<int64_t>(g0 + S, kEpsGroups);
    *x_begin = std::min<int64_t>(g0 * per_group, p->N);
    *x_end = shard == nshards - 1 ? p->N : std::min<int64_t>(g1 * per_group, p->N);
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

// diagnostic (FDJAC_TEST_SWITCHES=1 FDJAC_FUSED_TRACE=1): the wall_clock64 marks (100 MHz) of the plan's last fused launch
int fd_plan_fused_trace(fd_plan *p, long long *marks16)
{
    FD_REQUIRE(p && marks16, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(p->d_fz_trace != nullptr, FD_ERR_UNSUPPORTED, "no trace: FDJAC_FUSED_TRACE=1 (with FDJAC_TEST_SWITCHES=1) before the plan's first fused call");
    FD_HIP_CHECK(hipSetDevice(p->ctx->device));
    FD_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));
    FD_HIP_CHECK(hipMemcpy(marks16, p->d_fz_trace, 16 * sizeof(long long), hipMemcpyDeviceToHost));
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
