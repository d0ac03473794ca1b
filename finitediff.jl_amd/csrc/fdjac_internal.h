// Internal declarations shared by the libfdjac translation units (not part of the C ABI).
//
// The library is built twice from the same sources: once with real_t = double (symbols fd_*) and once, through the
// unity translation unit fdjac_f32.hip, with real_t = float (symbols fd32_*, include/fdjac.h "Float32 instantiation").
// The second build renames everything that depends on the element type and leaves the shared pieces (contexts,
// error text, colouring, copy probe) to the first.
#pragma once
#include <hip/hip_runtime.h>

#ifdef FDJAC_F32
#define FDJAC_REAL float
#define fdjac fdjac32
#define fd_plan fd32_plan
#define fd_jvp_plan fd32_jvp_plan
#define fd_plan_create_csc fd32_plan_create_csc
#define fd_plan_create_csc_device fd32_plan_create_csc_device
#define fd_plan_checksum fd32_plan_checksum
#define fd_plan_create_csc_dense fd32_plan_create_csc_dense
#define fd_plan_create_coo_dense fd32_plan_create_coo_dense
#define fd_plan_create_entries fd32_plan_create_entries
#define fd_plan_create_dense fd32_plan_create_dense
#define fd_plan_create_tridiagonal fd32_plan_create_tridiagonal
#define fd_plan_create_banded fd32_plan_create_banded
#define fd_plan_create_blockbanded fd32_plan_create_blockbanded
#define fd_plan_create_bandedblockbanded fd32_plan_create_bandedblockbanded
#define fd_plan_destroy fd32_plan_destroy
#define fd_plan_matches fd32_plan_matches
#define fd_plan_matches_async fd32_plan_matches_async
#define fd_plan_stale fd32_plan_stale
#define fd_plan_info fd32_plan_info
#define fd_plan_row_lists fd32_plan_row_lists
#define fd_jacobian fd32_jacobian
#define fd_jacobian_async fd32_jacobian_async
#define fd_jacobian_owned_async fd32_jacobian_owned_async
#define fd_plan_set_lazy_f fd32_plan_set_lazy_f
#define fd_plan_set_lazy_caps fd32_plan_set_lazy_caps
#define fd_plan_get_epsilons fd32_plan_get_epsilons
#define fd_plan_fused_trace fd32_plan_fused_trace
#define fd_plan_set_comm fd32_plan_set_comm
#define fd_plan_set_p2p fd32_plan_set_p2p
#define fd_plan_set_halo fd32_plan_set_halo
#define fd_plan_eps_partials fd32_plan_eps_partials
#define fd_plan_eps_finalize fd32_plan_eps_finalize
#define fd_plan_set_eps_mode fd32_plan_set_eps_mode
#define fd_plan_eps_shard_range fd32_plan_eps_shard_range
#define fd_plan_enable_timing fd32_plan_enable_timing
#define fd_plan_set_timing_stride fd32_plan_set_timing_stride
#define fd_plan_get_timings fd32_plan_get_timings
#define fd_plan_get_timing_samples fd32_plan_get_timing_samples
#define fd_builtin_f_create fd32_builtin_f_create
#define fd_builtin_f_create_sparse fd32_builtin_f_create_sparse
#define fd_builtin_f_destroy fd32_builtin_f_destroy
#define fd_builtin_f_counts fd32_builtin_f_counts
#define fd_builtin_f_info fd32_builtin_f_info
#define fd_builtin_f_lazy fd32_builtin_f_lazy
#define fd_builtin_f_lazy_caps fd32_builtin_f_lazy_caps
#define fd_jvp_plan_set_lazy_caps fd32_jvp_plan_set_lazy_caps
#define fd_builtin_f_lazy_jvp_caps fd32_builtin_f_lazy_jvp_caps
#define fd_blocktridiag_solver fd32_blocktridiag_solver
#define fd_blocktridiag_solver_create fd32_blocktridiag_solver_create
#define fd_blocktridiag_solver_destroy fd32_blocktridiag_solver_destroy
#define fd_blocktridiag_solver_set_policy fd32_blocktridiag_solver_set_policy
#define fd_blocktridiag_solver_status fd32_blocktridiag_solver_status
#define fd_blocktridiag_solve_async fd32_blocktridiag_solve_async
#define fd_banded_solver fd32_banded_solver
#define fd_banded_solver_create fd32_banded_solver_create
#define fd_banded_solver_destroy fd32_banded_solver_destroy
#define fd_banded_solver_set_policy fd32_banded_solver_set_policy
#define fd_banded_solver_status fd32_banded_solver_status
#define fd_banded_solve_async fd32_banded_solve_async
#define fd_jvp_plan_create fd32_jvp_plan_create
#define fd_jvp_plan_destroy fd32_jvp_plan_destroy
#define fd_jvp fd32_jvp
#define fd_jvp_async fd32_jvp_async
#define fd_jvp_get_epsilon fd32_jvp_get_epsilon
#define fd_jvp_plan_set_lazy_f fd32_jvp_plan_set_lazy_f
#define fd_builtin_f_lazy_jvp fd32_builtin_f_lazy_jvp
#else
#define FDJAC_REAL double

#endif

#include <time.h>
#include <cstdarg>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "fdjac.h"
#include "fdjac_device.h"

// communicator internals shared by both element-type builds (fdjac_comm.hip)
extern "C" int fdjac_comm_allgather_f64(fd_comm *c, double *buf, int64_t slot_elems);
extern "C" int fdjac_comm_allreduce_max4(fd_comm *c, const double *mine, double *out);   // host values, blocking (attach time)
extern "C" int fdjac_comm_nranks(const fd_comm *c);
extern "C" int fdjac_comm_rank(const fd_comm *c);
extern "C" const fd_ctx *fdjac_comm_ctx(const fd_comm *c);
extern "C" fd_p2p *fdjac_comm_p2p(const fd_comm *c);       // the communicator's own mailbox (fd_comm_enable_p2p), or NULL
extern "C" const fd_ctx *fdjac_p2p_ctx(const fd_p2p *p);
extern "C" int fdjac_p2p_nranks(const fd_p2p *p);
extern "C" int fdjac_p2p_rank(const fd_p2p *p);
// level 2 of the step-size reduction, run by the last workgroup of the step exchange (fdjac_p2p.hip: element-type independent)
struct fdjac_eps_final {
    double *gsum;
    int ldp, ngroups, C, is_forward, elem_bytes;
    double relstep, absstep, dir;
    void *eps, *eps2;
};
// ONE launch per step of a sharded call: this rank's slot of `gsum` (slot_bytes bytes at rank * slot_bytes) to every peer, the
// halo of x to the neighbours (x = NULL / halo = 0: none), every peer's slot and the neighbours' halos received, then fin.
// FD_ERR_UNSUPPORTED (nothing enqueued) when the payload does not fit the mailbox slot.
extern "C" int fdjac_p2p_step(fd_p2p *p, void *x, int64_t own_begin, int64_t own_end, int64_t halo, int elem_bytes, double *gsum,
                              int64_t slot_bytes, const fdjac_eps_final *fin);

// the fused step's cells in a mailbox (csrc/fdjac_eps_dev.h, fdjac_p2p.hip): a cell holds a sentinel until its ONE writer stores the
// value; three buffers by epoch, each [8 colours x 64 groups group sums][lower halo][upper halo]
constexpr int kFzBufs = 3;
constexpr int64_t kFzGsumBytes = 64 * 8 * 8, kFzHaloBytes = 64, kFzBufBytes = kFzGsumBytes + 2 * kFzHaloBytes;
constexpr unsigned long long kFzSentinel64 = 0x7FF85EEDFD1AC0DEull;   // quiet NaNs with a payload: never the result of arithmetic on
constexpr unsigned kFzSentinel32 = 0x7FC5EED1u;                       //   ordinary inputs (an input NaN of exactly this payload: timeout)
constexpr unsigned long long kFzSentinelHalo64 = 0x7FC5EED17FC5EED1ull;      // halo cells: two Float32 sentinels = the Float64 halo sentinel

// the mailbox as the fused step of a sharded call sees it (fdjac_p2p_fused_begin advances the mailbox's epoch)
struct fdjac_p2p_fused {
    char *const *peer;       // device array of the peers' mailbox bases
    char *local;             // this rank's mailbox
    int64_t fz_off;          // offset of the fused step's cells in a mailbox
    int buf, buf_reset;      // this step's buffer; the buffer it resets (-1: loop-back, none)
    int *err;
    int nranks, rank;
};
extern "C" int fdjac_p2p_fused_begin(fd_p2p *p, fdjac_p2p_fused *out);      // (the step's buffer; the epoch advances with fdjac_p2p_fused_commit)
extern "C" void fdjac_p2p_fused_commit(fd_p2p *p);
extern "C" int fdjac_p2p_shared_device(const fd_p2p *p);      // 1: some peer lives on this rank's device (no fused sharded step then)
extern "C" int fdjac_p2p_failed(const fd_p2p *p);
extern "C" int *fdjac_p2p_err_word(const fd_p2p *p);    // (device address)
extern "C" int fdjac_p2p_agree8(fd_p2p *p, const double *mine, double *out);      // blocking all-gather of 8 doubles per rank (host arrays)       // the mailbox's sticky error word (a wait timed out)

// error text: one thread-local buffer for both instantiations (defined by the Float64 build)
extern "C" void fdjac_set_error_v(const char *fmt, va_list ap);

namespace fdjac {

typedef FDJAC_REAL real_t;                                        // element type of x, f!, J
typedef real_t r2_t __attribute__((ext_vector_type(2)));          // a pair of elements: the unit of vector access
constexpr uintptr_t kPairMask = 2 * sizeof(real_t) - 1;           // alignment of a pair (16 B for Float64)

inline void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    fdjac_set_error_v(fmt, ap);
    va_end(ap);
}

#define FD_HIP_CHECK(expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            ::fdjac::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                               __LINE__);                                                      \
            return FD_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)

#define FD_REQUIRE(cond, code, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            ::fdjac::set_error(__VA_ARGS__); \
            return (code);                   \
        }                                    \
    } while (0)

enum PlanKind { K_CSC = 0, K_CSC_DENSE, K_COO_DENSE, K_TRIDIAG, K_BANDED, K_COLRANGE, K_DENSE, K_BBB };

constexpr int kBlock = 256;          // 4 wave64 per workgroup
constexpr int kRegColors = 8;        // epsilon reduction keeps <= this many colour sums in registers
constexpr int kEpsGroups = 64;       // ... and is defined over this many contiguous groups of x (the unit a rank of a sharded reduction owns)
constexpr int kEpsBlocksPerGroup = 16;   //   each summed by at most this many workgroups (64 x 16 = 4 per CU; 64 per group was measured in round 6: a shard's
                                         //   pass 6.2 -> 4.7 us, but the unsharded N = 10^7 pass 24 -> 33 us -- 2496 workgroups and 39 arrivals per ticket)
constexpr int64_t kListPad = 4096;   // index lists are padded to a multiple of this many entries (>= largest tile)
constexpr int kEpsLdsMax = 2048;     // stage eps[] in LDS up to this many colours per chunk
constexpr int64_t kSmallN = 16384;   // below this a single-workgroup launch does step sizes (+ perturbation): launch-bound regime
constexpr int kSortTile = 2048;      // entries per workgroup of the sorted-gather (LDS-transposed) decompression
constexpr int kFxWin = 8;            // ... its tiles' runs of rows whose f(x) is staged in LDS (forward differences): at most this many,
constexpr int kFxRows = kSortTile;   //     this many rows in total (the staging area is the tile's value area)
constexpr int kWinMaxCol = 8;        // row-window decompression: at most this many consecutive colours per tile,
constexpr int kWinMaxWin = 4;        //   this many row windows per tile (a 5-point stencil needs 3),
constexpr int kWinPeriodMax = 64;    //   longest period (in entries) of a regular tile's codes; its head is staged in LDS:
constexpr int kWinHeadBytes = 128;   //   kWinPeriodMax 16-bit codes at the start of the 1-D kernel's dynamic LDS
constexpr int kWinGap = 64;          //   a new window starts after a gap of more than this many rows,
constexpr int kWinMaxLds = 52 * 1024;  // this much LDS per workgroup (3 workgroups per CU),
constexpr double kWinMaxOverread = 4.0;  // and this many f! values loaded per stored entry for scattered patterns
// 2-D (strided) tiles of the row-window kernel, for 2-D stencil patterns in natural ordering: a tile is kW2MaxRun
// column runs one stencil stride apart, so the row windows of neighbouring grid rows are shared inside the tile
constexpr int kW2Desc = 64;          // ints per tile descriptor
constexpr int kW2MaxWin = 12;        // row windows per tile
constexpr int kW2MaxRun = 8;         // column runs per tile

// Exact floor(n / d) for every n < 2^31 by one 64-bit multiply (Granlund-Montgomery: m = ceil(2^(31+l) / d), l = ceil(log2 d),
// so m < 2^32 and n * m < 2^63): the multiplier and the shift travel packed in one 64-bit kernel argument.
inline uint64_t fd_magic31(uint32_t d)
{
    int l = 0;
    while (((uint64_t)1 << l) < d) ++l;
    const uint64_t m = ((((uint64_t)1) << (31 + l)) + d - 1) / d;
    return (m << 8) | (uint64_t)(31 + l);
}
__host__ __device__ inline uint32_t fd_div31(uint32_t n, uint64_t packed)
{
    return (uint32_t)(((uint64_t)n * (packed >> 8)) >> (packed & 0xFFu));
}

// XCD-aware tile mapping.  MI355X dispatches workgroup b to XCD b % 8 and each XCD has a private
// 4 MiB L2.  Patterns whose gathers revisit a row from several places of the storage order
// (5-point stencils: rows k-nx, k, k+nx) would otherwise fetch every line into up to 3 different
// L2s.  Giving XCD x the contiguous tile range [x*q, (x+1)*q) keeps each revisit in the L2 that
// already holds the line.  Launch 8*q workgroups (q = ceil(ntiles/8)); tiles >= ntiles exit.
// Placement is a performance assumption only -- results never depend on it.
__host__ __device__ inline int64_t xcd_chunks(int64_t ntiles) { return (ntiles + 7) / 8; }
__device__ inline int64_t xcd_tile(int64_t block, int64_t ntiles)
{
    return (block & 7) * xcd_chunks(ntiles) + (block >> 3);
}

// the step rule applied to a masked sum of squares t = sum_{color[j] == c} x[j]^2 (every step-size kernel ends in this)
template <typename T> __device__ __forceinline__ T eps_rule(double t, double relstep, double absstep, double dir, int is_forward)
{
    // norm(x2), sqrt and the step rule in the element type, as the reference computes them:
    //   forward: max(relstep*abs(sqrt(norm)), absstep)*dir   (src/epsilons.jl:26-29; the sqrt of the 2-norm is src/jacobians.jl:561)
    //   central: max(relstep*abs(sqrt(norm)), absstep)       (src/epsilons.jl:50-53; jacobians.jl:602)
    const T nrm = (T)sqrt(t);                  // norm(x2)
    const T xs = fabs(sqrt(nrm));              // abs(sqrt(tmp))
    const T a = (T)relstep * xs;
    T e = (a > (T)absstep) ? a : (T)absstep;
    if (is_forward) e = e * (T)dir;
    return e;
}

// TEST SWITCHES.  A handful of FDJAC_* environment variables select between bit-identical kernel variants / plan builders so that the
// tests can run both sides of every "same bits" claim (tests/test_gpu_parity.py, test_gpu_planbuild.py, ...):
//   FDJAC_SMALL (fused single-workgroup launches of small problems), FDJAC_LAZY_DIFF / FDJAC_LAZY_STORE (hand-over forms of the lazy
//   launchers), FDJAC_EPS_CYCLIC (the step-size reduction's computed colours), FDJAC_BAND_DESC (computed tile
//   descriptors), FDJAC_PLAN_DEVICE (host vs device plan builder), FDJAC_WINDOW / FDJAC_WINDOW2D / FDJAC_SORTED / FDJAC_WIN_TILE /
//   FDJAC_WIN_PERIODIC / FDJAC_TILE_ORDER (which decompression kernel a hand-over plan compiles to).
// They are read through this ONE function, and only in a process that opted in with FDJAC_TEST_SWITCHES=1 (tests/conftest.py sets it):
// a production process ignores them altogether -- there every choice is the plan builder's.  (Operational variables are not gated:
// FDJAC_RCCL_LIB, FDJAC_HIPRTC_LIB, FDJAC_P2P_TIMEOUT_MS, FDJAC_PLAN_THREADS, FDJAC_PLAN_TIMING.)
inline const char *test_switch(const char *name)
{
    static const bool on = [] { const char *v = getenv("FDJAC_TEST_SWITCHES"); return v && *v && atoi(v) != 0; }();
    return on ? getenv(name) : nullptr;
}

struct TimedSpan {
    int stage;
    hipEvent_t a, b;
};

}  // namespace fdjac

// content fingerprints of the arrays a plan was compiled from (FD_PLAN_FINGERPRINT, fdjac_match.hip), in the caller's units
struct fd_fingerprint {
    bool valid = false;
    int idx_kind = 0;                 // 0 structural (colours only), 1 CSC (a = colptr, b = rowval), 2 index lists (a = rows, b = cols)
    uint64_t h_a = 0, h_b = 0, h_color = 0;
    int64_t len_a = 0, len_b = 0, len_color = 0;   // the arrays' lengths at plan creation
    int64_t a0 = 0, an = 0, b0 = 0, bn = 0;        // the ranges that were hashed
};

struct fd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    // fd_plan_matches_async: a sticky "some plan was found stale" word in pinned host memory mapped to the device -- a kernel raises
    // it, the host reads it without a synchronisation (fd_ctx_synchronize / the next fd_jacobian* of that plan report FD_ERR_STALE)
    int *h_stale = nullptr, *d_stale = nullptr;
    // ... and the side stream those checks run on (round 6): the fused fingerprint kernel reads the caller's arrays BESIDE the Jacobian
    // instead of ahead of it; it starts behind an event of the main stream (the arrays' producers) and nothing waits for it but
    // fd_ctx_synchronize / the plan's destruction
    hipStream_t check_stream = nullptr;
    hipEvent_t check_event = nullptr;
};

struct fd_plan {
    fd_ctx *ctx = nullptr;
    int kind = 0;
    int fdtype = 0;
    int64_t M = 0, N = 0;
    int64_t C = 0;        // maximum(colorvec)
    bool color8 = true;   // colours stored as uint8 (C <= 254) else int32
    int64_t col0 = 0, col1 = 0, x0 = 0, x1 = 0, row0 = 0, row1 = 0;
    int64_t own_c0 = 0, own_c1 = -1;   // owned colours [own_c0, own_c1), -1 = up to C (fd_plan_opts.color_begin/end)
    // kernel variants, fixed at plan creation (environment switches are read there, never per process):
    bool tri_window = false;       //   K_TRIDIAG: row-window kernel (FDJAC_WINDOW != 0, C <= 4, even first column)
    bool small_ok = true;          //   fused single-workgroup launches of small problems allowed (FDJAC_SMALL != 0)
    bool eps_nt = true;            //   step-size reduction reads x with non-temporal loads: decided per call
    int eps_tpg = 0, eps_bpg = 0, eps_tpb = 0;   // the two-level reduction's grid (k_eps_partial_reg): tiles per group, blocks per group, tiles
                                   //   per block -- functions of N alone, so every rank / window / shard of one problem cuts x the same way
    bool cx = false;               // complex-valued x (FD_PLAN_COMPLEX_X): this plan is the lowered REAL problem -- element 2j / 2j+1 =
                                   //   re / im of x_j, only the real parts carry colours (are perturbed), a coloured element's masked
                                   //   norm includes its imaginary partner (|x_j|^2), f! is called with is_complex = 1
    int cyc_C = 0, cyc_shift = 0;  // cyclic colours: color[j] == (j + cyc_shift) mod cyc_C for every column (0 = not cyclic);
                                   //   the reduction then computes the colours instead of reading them (FDJAC_EPS_CYCLIC=0: off)

    // pattern (device)
    void *d_color = nullptr;       // per column, 0-based colour, "none" = all-ones
    int32_t *d_rowval = nullptr;   // per local stored entry, 0-based row
    void *d_nzcolor = nullptr;     // per local stored entry
    int64_t *d_dest = nullptr;     // per local stored entry destination offset (dense-J kinds)
    // sorted-gather variant (scattered patterns): per tile of kSortTile entries, entries ordered by colour
    bool sorted_gather = false;
    uint16_t *d_spos = nullptr;    //   local output position (within the tile) of each sorted entry
    bool has_none = false;         //   some column has no colour (its entries are written as 0)
    bool built_on_device = false;  // the pattern was compiled by the device builder (fdjac_planbuild.hip), not the host loops
    double lines_direct = 0, lines_sorted = 0;  // plan-time estimate: distinct 128-B lines per wave gather
    // row-window variant (locally banded patterns): per tile of kSortTile entries the rows fall into a
    // short window and few colours, so the f! values are loaded DENSELY into LDS and gathered from there
    bool window = false;
    int4 *d_wtiles = nullptr;      //   3 x int4 per tile: {first colour, colours, row pairs, windows}, 4 x {first row, end pair}
    int win_tile = 0;              //   entries per tile (2048 or 1024)
    bool window2d = false;         //   2-D (strided) tiles: d_w2desc[kW2Desc * ntiles], codes in tile order
    int *d_w2desc = nullptr;
    // uniform band with cyclic colours: local entry p <-> Q = p + band_off = band_w * j + k, row j - band_u + k, colour
    // (j + band_shift) mod band_C; band_mw / band_mc = fd_magic31(band_w), fd_magic31(band_C): exact dividers for n < 2^31
    int64_t band_off = 0, band_C = 0;
    int band_w = 0, band_u = 0, band_shift = 0;
    uint64_t band_mw = 0, band_mc = 0;
    // the row-window kernel COMPUTES the descriptors of the tiles [bd_t0, bd_t1)
    // from the same band parameters instead of loading them (verified against the stored descriptors when the plan is
    // built) -- the descriptor load is one of two dependent global round trips of a workgroup's lifetime
    // the pattern is verified to be the exact band include/fdjac_device.h describes (CSC: corners included; BandedMatrix /
    // Tridiagonal storage: by construction) with cyclic colours: a FD_LAZY_CAP_STORE launcher then stores the quotients
    // itself and no decompression is launched (FDJAC_LAZY_STORE=0: never)
    bool store_allowed = true, store_ok = false;
    int store_l = 0, store_u = 0, store_C = 0, store_shift = 0;
    // ... and the 5-point stencil on an nx x ny grid (fd_stencil5_store): exact pattern + valid colouring verified
    bool store5_ok = false;
    int64_t store5_nx = 0, store5_ny = 0;
    // ... and ANY pattern, column by column, through a compact device copy of the local pattern (fd_csc_store; FD_PLAN_STORE_CSC)
    bool want_store_csc = false, store_csc_always = false, store_csc_ok = false;
    // BandedBlockBandedMatrix (K_BBB): block structure and the banded-data slab of every in-band block
    int64_t bbb_nb = 0;
    int bbb_bl = 0, bbb_bu = 0, bbb_lam = 0, bbb_mu = 0;
    bool bbb_fill = false;             // some slots of data belong to no slab: zero-fill before the launch
    bool store_bbb_ok = false;         // uniform blocks + a colouring verified VALID for the BBB pattern: a FD_LAZY_CAP_STORE launcher may store
    int64_t bbb_bs = 0;                //   (fd_bbb_store); the uniform block size
    int32_t *d_bbb_off = nullptr;      // [nb + 1] first row / column of every block
    int32_t *d_bbb_blk = nullptr;      // [N] block of every column
    int64_t *d_bbb_start = nullptr;    // [(bl + bu + 1) * nb] 0-based start of block (K, J)'s slab in data, -1: not in the band
    int64_t *d_bbb_stride = nullptr;   // [nb] column stride of the slabs of block-column J
    int32_t *d_sc_colptr = nullptr, *d_sc_rowval = nullptr;
    bool want_store_rows = false;                // FD_PLAN_STORE_CSC_ROWS: the same pattern by rows (fd_csc_store.row_ptr / row_col / row_slot), full column range only
    int32_t *d_sr_ptr = nullptr, *d_sr_col = nullptr, *d_sr_slot = nullptr, *d_sr_order = nullptr, *d_sr_tile = nullptr;
    int se_tile_max = 0;
    int32_t *d_se_col = nullptr, *d_se_slot = nullptr, *d_se_info = nullptr;      // the entries in the tiles' row order (fd_csc_store.ent_*), or none
    unsigned long long *d_sc_note = nullptr;     // fd_csc_store.note: four words of launcher memory about this pattern, zero at creation
    unsigned long long sc_serial = 0;            // fd_csc_store.plan_serial
    int64_t sc_entries = 0;
    int64_t sc_reach = -1;         //   max |row - column| over the local entries (fd_csc_store.reach), -1: not computed
    bool sc_valid = false;         //   colorvec verified to be a valid colouring of the local pattern (kernels may perturb one coordinate)
    // ... and block-banded storage (fd_colrange_store): valid colouring verified; uniform block structure recorded
    bool store_cr_ok = false;
    int64_t cr_nblk = 0, cr_bs = 0;
    int cr_bl = 0, cr_bu = 0;
    bool bd_allowed = true;        //   FDJAC_BAND_DESC=0: always load
    int64_t bd_t0 = 0, bd_t1 = 0;
    int64_t w2_ntiles = 0;
    int64_t w2_codes = 0;          //   number of 16-bit codes in d_wcode (2-D tiles)
    uint16_t *d_wcode = nullptr;   //   per entry: row - first row | (colour - first colour) << 11 | none << 14 | pad << 15
    int win_pairs = 0;             //   max row pairs of any tile (LDS pitch = 2*win_pairs doubles)
    int win_ncol = 0;              //   max colours of any tile
    int win_per_P = 0, win_per_S = 0, win_per_magic = 0;   // periodic entry codes of regular tiles (0 = none)
    double win_overread = 0;       //   dense window elements loaded per stored entry (1 = no waste)
    int64_t nnz_local = 0;
    int64_t entry_begin = 0;       // global index of the first local stored entry
    int64_t l = 0, u = 0;          // banded
    int32_t *d_cr_rlo = nullptr;   // colrange (block-banded): first row of each local column
    int32_t *d_cr_cnt = nullptr;   //   number of contiguous rows
    int64_t *d_cr_off = nullptr;   //   destination offset of the first row
    bool cr_pairs = false;         //   all three even for every column: the kernel works on row pairs (16 B)
    // segmented epsilon reduction (C > kRegColors)
    int32_t *d_perm = nullptr;     // columns sorted by colour
    int64_t *d_cptr = nullptr;     // C+1 offsets into perm
    int seg_chunks = 1;

    // scratch (device)
    int64_t ldx = 0, ldf = 0;      // leading dimensions (elements) of the batched point / value arrays
    int64_t chunkB = 0, nchunks = 0;
    int pts = 1;                   // f! points per colour (2 for central)
    int cplx = 0;                  // elements are (re,im) pairs
    fdjac::real_t *d_X = nullptr, *d_FX = nullptr, *d_fx = nullptr, *d_eps = nullptr;
    int32_t *d_fxwin = nullptr;        // sorted-gather plans: per tile kFxWin x (first row, rows) of the runs its entries' rows lie in (first < 0: none)
    int32_t *d_tile_order = nullptr;   // sorted-gather plans with a far band: the order in which the tiles are walked (else storage order)
    double *d_partial = nullptr;   // masked sums of squares are accumulated in Float64 for either element type
    double *d_gsum = nullptr;      //   [2 * kEpsGroups][kRegColors] group sums of the two-level reduction (the second half: padding of the
                                   //   last rank's slot when the groups do not divide by the rank count)
    unsigned *d_tick = nullptr;    //   [kEpsGroups + 1] arrival tickets (zero between launches)
    int64_t halo = 0, halo_own0 = 0, halo_own1 = 0;   // fd_plan_set_halo: x is sharded -- the call exchanges `halo` elements with the neighbour ranks
    fdjac::real_t *d_xstage = nullptr, *d_finstage = nullptr;
    int n_partial_blocks = 0;
    int64_t scratch_bytes = 0;

    int nouts = 1;
    int64_t out_len[3] = {0, 0, 0};
    // complex-valued x on Tridiagonal storage: the lowered plan writes ONE concatenated array (dl | d | du) into d_split, which
    // the call then copies into the caller's three arrays (split_n = 3, split_len in reals)
    int split_n = 0;
    int64_t split_len[3] = {0, 0, 0};
    fdjac::real_t *d_split = nullptr;
    fdjac::real_t *d_outstage[3] = {nullptr, nullptr, nullptr};

    const fdjac::real_t *fx_batch_row = nullptr;   // f(x) evaluated as one more member of the perturbed batch (small problems)
    fd_f_launch_lazy lazy_fn = nullptr;
    int lazy_caps = 0;             // FD_LAZY_CAP_* of lazy_fn
    bool lazy_diff = true;         // ask a FD_LAZY_CAP_DIFF launcher for differences (FDJAC_LAZY_DIFF=0: never)
    fdjac::real_t *d_zero_own = nullptr;      // an all-zero vector of ldf elements (allocated on first use by forward / central plans)
    const fdjac::real_t *d_zero = nullptr;    // the "fx" of decompressions whose f! arrays already hold differences or imaginary
                                              //   parts: d_fx for the complex step (never written), d_zero_own otherwise
    fdjac::real_t *d_eps2 = nullptr;          // 2 * eps per colour (central differences handed over as f(+) - f(-))
    bool eps2_fresh = false;                  //   d_eps2 matches d_eps (written by the finalize launch; else launch_scale)
    // the fused step (fdjac_eps_dev.h): block sums and published step sizes, each double-buffered by call parity and holding a
    // sentinel until written; an error word in pinned host memory (a wait timed out: FD_ERR_COMM from the next call)
    double *d_fz_part = nullptr;              // [2][n_partial_blocks][kRegColors]
    void *d_fz_eps = nullptr;                 // [2][kFzReplicas][kFzPitch] elements
    int *h_fz_err = nullptr, *d_fz_err = nullptr;
    unsigned fz_parity = 0;
    long long *d_fz_trace = nullptr;          // FDJAC_FUSED_TRACE=1: wall_clock64 marks of the last fused launch (fd_plan_fused_trace)
    bool fz_sharded_ok = true;                // sharded calls with a mailbox take the fused step (shards up to fz_max_n columns)
    bool fz_flags_ok = true;                  // the reduction's own launch in the flag form (k_eps_flags; FDJAC_EPS_FLAGS=0: tickets, k_eps_partial_reg)
    bool fz_shared_ok = false;                //   ... even when ranks share this device (FDJAC_FUSED_SHARED=1: small test problems only)
    int64_t fz_max_n = (int64_t)1 << 21;      // single GPU: problems up to this size take the fused step (FDJAC_FUSED_MAX_N; 0 = never)
    fd_comm *comm = nullptr;       // sharded step-size reduction (fd_plan_set_comm); nullptr = every rank reduces all of x
    fd_p2p *p2p = nullptr;         //   ... its group sums (and the halo of x) travel through this mailbox in ONE launch (fd_plan_set_p2p, or the
                                   //   communicator's own mailbox)
    int eps_mode = 0;              // FD_EPS_COMPUTE / FD_EPS_PRECOMPUTED
    int64_t partial_cap = 0;       // doubles allocated behind d_partial
    int64_t fcalls_last = 0;
    double relstep_last = 0, absstep_last = 0;

    fd_fingerprint fp;                          // FD_PLAN_FINGERPRINT: what fd_plan_matches compares against
    unsigned long long *d_fp = nullptr;         //   three device words for the fingerprint kernels (allocated on first use)
    unsigned long long *d_fpx = nullptr;        //   fd_plan_matches_async: [0..2] accumulators, [3..5] the plan's fingerprints, [6] ticket
    int *h_pstale = nullptr, *d_pstale = nullptr;   //   ... this plan's own sticky stale word (pinned host memory mapped to the device)

    int timing = 0;   // 0 off, 1 decompress + total, 2 all stages
    int timing_stride = 1;      // level 1: only every timing_stride-th call carries the two events (fd_plan_set_timing_stride)
    int64_t timing_calls = 0;
    std::vector<fdjac::TimedSpan> spans;       // recorded, not yet collected
    std::vector<hipEvent_t> event_pool;
    double ms_sum[FD_NSTAGES] = {};
    int64_t launches[FD_NSTAGES] = {};
    std::vector<float> samples[FD_NSTAGES];    // the individual spans (fd_plan_get_timing_samples), capped
};

namespace fdjac {
int plan_record_fingerprint(fd_plan *p, int idx_kind, const fd_pattern_arrays *src, int64_t col0, int64_t col1);   // fdjac_match.hip
}

// does this plan let a FD_LAZY_CAP_STORE launcher store the Jacobian itself (fd_lazy_points.store)?
// ... column by column through the compact pattern copy of a general pattern (any colouring, column window, colour chunk, ownership;
// columns without colour too; forward differences take f(x) from f_in or from one plain evaluation)
static inline bool store_csc_active(const fd_plan *p)
{
    return p->lazy_fn && (p->lazy_caps & FD_LAZY_CAP_STORE_CSC) && p->store_csc_ok && p->kind == fdjac::K_CSC &&
           (p->fdtype != FD_COMPLEX || (p->lazy_caps & FD_LAZY_CAP_STORE_CSC_COMPLEX)) && p->store_allowed;
}
static inline bool store_active(const fd_plan *p)
{
    if (!(p->lazy_fn && (p->lazy_caps & FD_LAZY_CAP_STORE)) || p->has_none) return false;
    // (the block-coupled launcher's storing kernel is the complex step's; a launcher with FD_LAZY_CAP_STORE_COLRANGE -- a compiled functor -- serves all three)
    if (p->kind == fdjac::K_COLRANGE) return p->store_cr_ok && (p->fdtype == FD_COMPLEX || (p->lazy_caps & FD_LAZY_CAP_STORE_COLRANGE) != 0);
    if (p->kind == fdjac::K_BBB) return p->store_bbb_ok && p->fdtype != FD_COMPLEX && p->store_allowed && p->nchunks == 1 && p->own_c0 == 0 &&
                                        (p->own_c1 < 0 || p->own_c1 >= p->C);
    return (p->store_ok || (p->store5_ok && p->kind == fdjac::K_CSC)) && p->fdtype != FD_COMPLEX &&
           (p->kind == fdjac::K_CSC || p->kind == fdjac::K_BANDED || p->kind == fdjac::K_TRIDIAG);
}

// FDJAC_PLAN_TIMING=1: wall-clock of the builder's sections on stderr (each mark synchronises the stream)
struct PbTimer {
    bool on;
    hipStream_t s;
    double t0;
    static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    PbTimer(hipStream_t st) : s(st) { const char *v = getenv("FDJAC_PLAN_TIMING"); on = v && *v && atoi(v) != 0; t0 = on ? now() : 0; }
    void mark(const char *what) { if (!on) return; (void)hipStreamSynchronize(s); const double t = now(); fprintf(stderr, "[fdjac plan] %-28s %8.3f ms\n", what, t - t0); t0 = t; }
};
