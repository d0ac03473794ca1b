// The consumer of the path: a device tridiagonal solve that takes the Jacobian WHERE
// fd_jacobian_async LEFT IT -- the three diagonals of a Tridiagonal J (outs of fd_plan_create_tridiagonal) or the nzval
// of a tridiagonal SparseMatrixCSC, whole or one rank's column range -- and solves
//        (alpha*I + beta*J) y = b
// the linear system of an implicit / Rosenbrock step with a `Tridiagonal` jac_prototype (downstream use of the
// reference: test/downstream/ordinarydiffeq_tridiagonal_solve.jl:18-30, W = I - gamma*J).  No reference counterpart
// inside FiniteDiff.jl itself: there the factorisation is LinearAlgebra's; here it is what keeps the multi-GPU Jacobian
// sharded (no 240 MB gather over xGMI before the solve: one all-gather of 8 numbers per rank instead).
//
// Algorithm: Wang's partition method, applied recursively.  A level's rows are cut into chunks of kChunk rows, one per
// thread.  A downward and an upward elimination sweep inside the chunk (registers only) leave one equation per chunk
// that couples the LAST unknown of three neighbouring chunks: a tridiagonal system 8 times smaller.  Levels are
// reduced until <= kTop unknowns remain (parallel cyclic reduction in LDS, one workgroup), then every level is
// back-substituted: a thread re-reads its chunk's original rows and solves them with the two now-known boundary
// values.  No pivoting: the method is for diagonally dominant systems (I - gamma*J of a diffusion-type J), like the
// partitioned / cyclic-reduction tridiagonal solvers of the vendor libraries.  Level 0 reads the user's J and b once
// in the reduction and once in the back-substitution (9 values per row in total + 1 written).  Measured at N = 10^7 on
// MI355X: 0.19 ms per solve, 5 launches -- two levels per launch (k_tri_reduce2 / k_tri_top2 / k_tri_backsub2 below),
// profiles/r05_p_solve.md.
//
// Multi-GPU (rows = the rank's column range): the rank's block T_r of the matrix is complete in its own slice; the two
// couplings to the neighbouring ranks are not (they live in the neighbours' columns).  SPIKE form:
//   phase A  T_r [g v w] = [b  e_first  e_last] (3 right-hand sides through the same reduction; only the 6 tip values
//            are back-substituted) -> packet of 8 numbers: the tips and the two couplings this rank knows FOR its
//            neighbours; ONE all-gather of the packets (fd_comm_allgather);
//   phase B  every rank solves the 2W x 2W interface system (one thread) and picks y just outside its rows;
//   phase C  T_r y = b - (coupling * neighbour value) at the first / last row: a plain local solve.
#include <new>

#include "fdjac_internal.h"

#ifdef FDJAC_F32
#define fd_tridiag_solver fd32_tridiag_solver
#define fd_tridiag_solver_create fd32_tridiag_solver_create
#define fd_tridiag_solver_destroy fd32_tridiag_solver_destroy
#define fd_tridiag_solver_status fd32_tridiag_solver_status
#define fd_tridiag_solver_set_policy fd32_tridiag_solver_set_policy
#define fd_tridiag_solve_async fd32_tridiag_solve_async
#define fd_tridiag_solve_interface fd32_tridiag_solve_interface
#define fd_tridiag_solve_finish fd32_tridiag_solve_finish
#endif

namespace fdjac {

constexpr int kChunk = 8;        // rows per thread
constexpr int kTop = 512;        // a level with at most this many rows is solved by one workgroup (PCR in LDS)
constexpr int kSumVals = 12;     // per chunk: f_e b_e c_e d_e[3]  f_s b_s g_s d_s[3]
constexpr int kMaxLevels = 12;
constexpr int kPacket = 8;       // doubles per rank in the interface exchange

struct TriRow {
    double a, b, c, d[3];
};

// level 0: the user's matrix, shifted and scaled, restricted to the local rows [g0, g0 + n)
struct SrcUser {
    int layout;                 // FD_TRI_DIAGONALS / FD_TRI_CSC
    const real_t *p0, *p1, *p2; // dl, d, du (window-relative, as the plan's outs)  |  nzval slice, -, -
    const real_t *rhs;          // local rows
    const double *adj;          // nullptr, or {subtract from the first row's rhs, subtract from the last row's}
    int64_t g0, n, N;           // first global row, local rows, global size
    int64_t e0;                 // CSC: global index of the slice's first stored value
    double alpha, beta;
    int *nd_flag;               // raised when a row is not diagonally dominant (|b| < |a| + |c|): the elimination does not pivot
    int refuse;                 // 1 (default): a solve that raised the flag writes NaN instead of its unreliable solution
    __device__ __forceinline__ void coef(int64_t i, double &a, double &b, double &c) const
    {
        const int64_t gi = g0 + i;
        double Aa = 0, Ab, Ac = 0;
        if (layout == FD_TRI_DIAGONALS) {
            const int64_t du0 = g0 > 0 ? g0 - 1 : 0;
            Ab = (double)p1[i];
            if (i > 0) Aa = (double)p0[i - 1];                 // A[gi, gi-1] = dl[gi-1-g0]
            if (i + 1 < n) Ac = (double)p2[gi - du0];          // A[gi, gi+1] = du[(gi+1)-1-du0]
        } else {
            Ab = (double)p0[3 * gi - e0];
            if (i > 0) Aa = (double)p0[3 * gi - 2 - e0];
            if (i + 1 < n) Ac = (double)p0[3 * gi + 2 - e0];
        }
        a = beta * Aa;
        b = alpha + beta * Ab;
        c = beta * Ac;
    }
    template <int NRHS> __device__ __forceinline__ TriRow load(int64_t i) const
    {
        TriRow r;
        coef(i, r.a, r.b, r.c);
        double d0 = (double)rhs[i];
        if (adj) {
            if (i == 0) d0 -= adj[0];
            if (i == n - 1) d0 -= adj[1];
        }
        r.d[0] = d0;
        r.d[1] = (NRHS > 1 && i == 0) ? 1.0 : 0.0;           // e_first
        r.d[2] = (NRHS > 1 && i == n - 1) ? 1.0 : 0.0;       // e_last
        return r;
    }
    template <int NRHS> __device__ __forceinline__ TriRow loadl(int, int64_t i) const { return load<NRHS>(i); }
    static constexpr bool kUnitRhs = true;    // right-hand sides 1, 2 are the unit vectors e_first, e_last: not fetched
    // the right-hand side of row i once its stored value d0 is there (the interface corrections of a sharded solve)
    __device__ __forceinline__ double rhs_fix(int64_t i, double d0) const
    {
        if (adj) {
            const double a0 = adj[0], a1 = adj[1];
            d0 = i == 0 ? d0 - a0 : d0;
            d0 = i == n - 1 ? d0 - a1 : d0;
        }
        return d0;
    }
    // (tri_fetch_rows issues every raw() of a tile before the first fix(): nothing consumes a load while others can still be issued)
    __device__ __forceinline__ double raw(int which, int64_t i) const { return which == 3 ? (double)rhs[i] : value(which, i); }
    __device__ __forceinline__ double fix(int which, int64_t i, double v) const { return which == 3 ? rhs_fix(i, v) : v; }
    // one coefficient of local row i: which = 0 a, 1 b, 2 c, 3 right-hand side.  Every load is unconditional -- a coefficient that
    // does not exist (a of the first row, c of the last) is fetched from the row's own diagonal entry and selected away: a load
    // inside a per-lane conditional is waited for at the join, and tri_fetch_rows wants ALL of a tile's loads in flight at once.
    __device__ __forceinline__ double value(int which, int64_t i) const
    {
        const int64_t gi = g0 + i;
        if (which == 3) return rhs_fix(i, (double)rhs[i]);
        const bool has = which == 0 ? i > 0 : i + 1 < n;
        const real_t *at;
        if (layout == FD_TRI_DIAGONALS) {
            const int64_t du0 = g0 > 0 ? g0 - 1 : 0;
            at = which == 1 ? p1 + i : which == 0 ? (has ? p0 + (i - 1) : p1 + i) : (has ? p2 + (gi - du0) : p1 + i);
        } else {
            at = p0 + (3 * gi - e0 + (which == 1 || !has ? 0 : which == 0 ? -2 : 2));
        }
        const double v = (double)*at;
        if (which == 1) return alpha + beta * v;
        return has ? beta * v : 0.0;
    }
};

// level >= 1: the reduced system defined by the chunk summaries of the level below
struct SrcLevel {
    const double *sum;   // [kSumVals][nc]
    int64_t nc;          // chunks below == rows here
    template <int NRHS> __device__ __forceinline__ TriRow load(int64_t k) const
    {
        TriRow r;
        r.a = sum[0 * nc + k];
        const double be = sum[1 * nc + k], ce = sum[2 * nc + k];
        if (k + 1 < nc) {
            const double t = ce / sum[7 * nc + k + 1];
            r.b = be - t * sum[6 * nc + k + 1];
            r.c = -t * sum[8 * nc + k + 1];
#pragma unroll
            for (int q = 0; q < 3; ++q) r.d[q] = q < NRHS ? sum[(3 + q) * nc + k] - t * sum[(9 + q) * nc + k + 1] : 0.0;
        } else {
            r.b = be;
            r.c = 0.0;
#pragma unroll
            for (int q = 0; q < 3; ++q) r.d[q] = q < NRHS ? sum[(3 + q) * nc + k] : 0.0;
        }
        return r;
    }
    template <int NRHS> __device__ __forceinline__ TriRow loadl(int, int64_t k) const { return load<NRHS>(k); }
    static constexpr bool kUnitRhs = false;
    __device__ __forceinline__ double raw(int which, int64_t k) const { return value(which, k); }
    __device__ __forceinline__ double fix(int, int64_t, double v) const { return v; }
    // one coefficient of row k: which = 0 a, 1 b, 2 c, 3 + q right-hand side q (lane-consecutive k: dense loads; all of them
    // unconditional -- the last row reads its own summary in place of the next one's and selects the result away)
    __device__ __forceinline__ double value(int which, int64_t k) const
    {
        if (which == 0) return sum[0 * nc + k];
        const bool nxt = k + 1 < nc;
        const int64_t kn = nxt ? k + 1 : k;
        const double t = sum[2 * nc + k] / sum[7 * nc + kn];
        if (which == 1) { const double s1 = sum[1 * nc + k], s6 = sum[6 * nc + kn]; return s1 - (nxt ? t * s6 : 0.0); }
        if (which == 2) { const double s8 = sum[8 * nc + kn]; return nxt ? -t * s8 : 0.0; }
        const int q = which - 3;
        const double sq = sum[(3 + q) * nc + k], s9 = sum[(9 + q) * nc + kn];
        return sq - (nxt ? t * s9 : 0.0);
    }
};

// where a chunk's summary goes: the level's array in memory, or (two levels per launch, below) a table in LDS
struct SumGlobal {
    double *__restrict__ sum;
    int64_t nc;
    __device__ __forceinline__ void put(int v, int64_t k, double x) const { sum[v * nc + k] = x; }
};
// One chunk's two elimination sweeps -> its summary (the row it contributes to the next level).
template <typename Src, int NRHS, typename Sink>
__device__ __forceinline__ void tri_reduce_chunk(const Src &src, int64_t n, const Sink &sink, int64_t k)
{
    const int64_t s = k * kChunk;
    const int m = (int)((n - s < kChunk) ? n - s : kChunk);
    double f[kChunk], b[kChunk], c[kChunk], d[kChunk][NRHS];
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
        if (i < m) {
            const TriRow r = src.template loadl<NRHS>(i, s + i);
            f[i] = r.a; b[i] = r.b; c[i] = r.c;
#pragma unroll
            for (int q = 0; q < NRHS; ++q) d[i][q] = r.d[q];
        } else {   // padding rows of the last chunk: identity rows, never used
            f[i] = 0; b[i] = 1; c[i] = 0;
#pragma unroll
            for (int q = 0; q < NRHS; ++q) d[i][q] = 0;
        }
    }
    // downward: row i -= (a_i / b_{i-1}) * row(i-1); the sub-diagonal turns into a coupling f_i to the unknown before the chunk
#pragma unroll
    for (int i = 1; i < kChunk; ++i)
        if (i < m) {
            const double mult = f[i] / b[i - 1];
            f[i] = -mult * f[i - 1];
            b[i] -= mult * c[i - 1];
#pragma unroll
            for (int q = 0; q < NRHS; ++q) d[i][q] -= mult * d[i - 1][q];
        }
    // (selects over the unrolled rows, never a runtime index: the rows live in registers)
    double fe = 0, be = 1, ce = 0, de[NRHS];
    double cf = 0, cb = 1, cg = -1, cd[NRHS];   // m == 1: the chunk's only row is its last unknown: y_first = z
#pragma unroll
    for (int q = 0; q < NRHS; ++q) { de[q] = 0; cd[q] = 0; }
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
        if (i == m - 1) {
            fe = f[i]; be = b[i]; ce = c[i];
#pragma unroll
            for (int q = 0; q < NRHS; ++q) de[q] = d[i][q];
        }
        if (i == m - 2) {   // start of the upward sweep: the row above the last one
            cf = f[i]; cb = b[i]; cg = c[i];
#pragma unroll
            for (int q = 0; q < NRHS; ++q) cd[q] = d[i][q];
        }
    }
    sink.put(0, k, fe);
    sink.put(1, k, be);
    sink.put(2, k, ce);
#pragma unroll
    for (int q = 0; q < NRHS; ++q) sink.put(3 + q, k, de[q]);      // (right-hand sides beyond NRHS are never read)
    // upward: the first row expressed through the unknown before the chunk (f) and the chunk's last unknown (g)
#pragma unroll
    for (int i = kChunk - 3; i >= 0; --i)
        if (i <= m - 3) {
            const double mult = c[i] / cb;
            cf = f[i] - mult * cf;
            cg = -mult * cg;
#pragma unroll
            for (int q = 0; q < NRHS; ++q) cd[q] = d[i][q] - mult * cd[q];
            cb = b[i];
        }
    sink.put(6, k, cf);
    sink.put(7, k, cb);
    sink.put(8, k, cg);
#pragma unroll
    for (int q = 0; q < NRHS; ++q) sink.put(9 + q, k, cd[q]);
}
// Rows through LDS, one coefficient array at a time.  A thread walks kChunk CONSECUTIVE rows; loading them directly makes
// every wave instruction touch 64 different cache lines and re-fetch each line up to 8 times from the L2 (measured: 1.7
// TB/s at level 0).  Instead the workgroup fetches one coefficient array of its tile of kBlock * kChunk rows with
// lane-consecutive loads, parks it in LDS at index r + r/8 (a thread's rows then start 9 doubles apart: conflict-free
// 8-byte reads), every thread copies its kChunk values to registers, and the same 18 KB of LDS serve the next array --
// 3 + NRHS rounds (a, b, c, right-hand sides), full occupancy.  Tridiagonal diagonals and the chunk summaries of the
// inner levels are read densely; the nzval of a tridiagonal CSC with stride 3 (each line three times).
constexpr int kTriTileRows = kBlock * kChunk;                    // 2048 rows
constexpr int kTriPitch = kTriTileRows + kTriTileRows / 8;       // padded
template <int NRHS> struct SrcRegs {
    double a[kChunk], b[kChunk], c[kChunk], d[NRHS][kChunk];
    template <int N2> __device__ __forceinline__ TriRow loadl(int li, int64_t) const
    {
        TriRow t;
        t.a = a[li]; t.b = b[li]; t.c = c[li];
#pragma unroll
        for (int q = 0; q < 3; ++q) t.d[q] = q < NRHS ? d[q < NRHS ? q : 0][li] : 0.0;
        return t;
    }
};
// The elimination does not pivot (LinearAlgebra's Tridiagonal \ does): it is accurate for diagonally dominant systems -- W = I - gamma J
// of diffusion-type problems.  Level 0 checks every row it fetches anyway and raises the solver's status word otherwise
// (fd_tridiag_solver_status): three absolute values and a compare per row, no extra traffic.
template <int NRHS>
__device__ __forceinline__ void tri_guard_rows(const SrcRegs<NRHS> &R, int64_t n, int64_t row0, int *nd_flag)
{
    if (!nd_flag) return;
    bool nd = false;
#pragma unroll
    for (int q = 0; q < kChunk; ++q) {
        const int64_t i = row0 + (int64_t)threadIdx.x * kChunk + q;
        nd = nd || (i < n && !(fabs(R.b[q]) >= fabs(R.a[q]) + fabs(R.c[q])));
    }
    if (nd) atomicOr(nd_flag, 1);
}
// (in two halves, issue and land: the two-level kernels below issue a tile's loads together with its halo's and land them later)
template <typename Src, int NRHS> struct TriLoads {
    static constexpr int kRounds = Src::kUnitRhs ? 4 : 3 + NRHS;
    double g[kRounds][kChunk];
};
template <typename Src, int NRHS>
__device__ __forceinline__ void tri_issue_rows(const Src &src, int64_t n, int64_t row0, TriLoads<Src, NRHS> &L)
{
    const int64_t rows = (n - row0 < kTriTileRows) ? n - row0 : kTriTileRows;
    constexpr int kRounds = TriLoads<Src, NRHS>::kRounds;
    // every array's global loads are issued before the first transposition: the memory latency is paid once, not once per
    // round (the barriers below would otherwise keep round w+1's loads behind round w's LDS traffic)
#pragma unroll
    for (int which = 0; which < kRounds; ++which)
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            const int r = j * kBlock + (int)threadIdx.x;  // lane-consecutive rows
            L.g[which][j] = src.raw(which, row0 + (r < rows ? r : (int)rows - 1));      // (unconditional: the tile's last row again)
        }
}
template <typename Src, int NRHS>
__device__ __forceinline__ void tri_land_rows(const Src &src, int64_t n, int64_t row0, TriLoads<Src, NRHS> &L, double *lds, SrcRegs<NRHS> &R)
{
    const int64_t rows = (n - row0 < kTriTileRows) ? n - row0 : kTriTileRows;
    constexpr int kRounds = TriLoads<Src, NRHS>::kRounds;
#pragma unroll
    for (int which = 0; which < kRounds; ++which)
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            const int r = j * kBlock + (int)threadIdx.x;
            L.g[which][j] = r < rows ? src.fix(which, row0 + r, L.g[which][j]) : 0.0;
        }
#pragma unroll
    for (int which = 0; which < kRounds; ++which) {
        if (which) __syncthreads();                       // the previous array has been copied out
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            const int r = j * kBlock + (int)threadIdx.x;
            if (r < rows) lds[r + (r >> 3)] = L.g[which][j];
        }
        __syncthreads();
        double *dst = which == 0 ? R.a : which == 1 ? R.b : which == 2 ? R.c : R.d[which - 3 < NRHS ? which - 3 : 0];
#pragma unroll
        for (int q = 0; q < kChunk; ++q) dst[q] = lds[9 * (int)threadIdx.x + q];   // rows 8t .. 8t+7 sit at 9t + q
    }
    if (Src::kUnitRhs && NRHS > 1) {                      // level 0: right-hand sides 1, 2 are e_first, e_last
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            const int64_t i = row0 + (int64_t)threadIdx.x * kChunk + q;
            R.d[NRHS > 1 ? 1 : 0][q] = i == 0 ? 1.0 : 0.0;
            R.d[NRHS > 2 ? 2 : 0][q] = i == n - 1 ? 1.0 : 0.0;
        }
    }
}
template <typename Src, int NRHS>
__device__ __forceinline__ void tri_fetch_rows(const Src &src, int64_t n, int64_t row0, double *lds, SrcRegs<NRHS> &R)
{
    TriLoads<Src, NRHS> L;
    tri_issue_rows<Src, NRHS>(src, n, row0, L);
    tri_land_rows<Src, NRHS>(src, n, row0, L, lds, R);
}
template <typename Src, int NRHS>
__global__ void __launch_bounds__(kBlock) k_tri_reduce(Src src, int64_t n, double *__restrict__ sum, int64_t nc)
{
    __shared__ double lds[kTriPitch];
    SrcRegs<NRHS> R;
    tri_fetch_rows<Src, NRHS>(src, n, (int64_t)blockIdx.x * kTriTileRows, lds, R);
    if constexpr (Src::kUnitRhs) tri_guard_rows<NRHS>(R, n, (int64_t)blockIdx.x * kTriTileRows, src.nd_flag);      // (level 0 only)
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= nc) return;
    tri_reduce_chunk<SrcRegs<NRHS>, NRHS>(R, n, SumGlobal{sum, nc}, k);
}

// Level 0 on the nzval of a tridiagonal CSC.  Column j stores (du_j, d_j, dl_j) = A[j-1,j], A[j,j], A[j+1,j] at 3j-1, 3j, 3j+1, so
// row i's coefficients sit at 3i-2, 3i, 3i+2: fetched coefficient by coefficient (tri_fetch_rows) every line is requested by three
// different instructions and comes from memory three times (rocprofv3: 780 MB fetched for 320 MB of J and b at N = 10^7).  Here the
// tile's span of nzval -- 3 * 2048 + 2 consecutive values -- is loaded ONCE with lane-consecutive loads into LDS (slot q + q/24: a
// thread's 26 values then start 25 doubles apart, conflict-free) and every thread picks its 8 rows' a, b, c out of it, with coef()'s
// arithmetic; the right-hand side follows through the same LDS as before.
constexpr int kTriRawVals = 3 * kTriTileRows + 2;
constexpr int kTriRawSlots = kTriRawVals + kTriRawVals / 24 + 2;
constexpr int kTriRawPer = (kTriRawVals + kBlock - 1) / kBlock;
struct TriLoadsCsc {
    real_t g[kTriRawPer];
    double gr[kChunk];
};
__device__ __forceinline__ void tri_issue_rows_csc(const SrcUser &src, int64_t n, int64_t row0, TriLoadsCsc &L)
{
    const int64_t rows = (n - row0 < kTriTileRows) ? n - row0 : kTriTileRows;
    // raw value q of the tile <-> nzval[lo + q]; row r (tile-local): a at 3r, b at 3r + 2, c at 3r + 4
    const int64_t lo = 3 * (src.g0 + row0) - src.e0 - 2;
    const int64_t qmin = lo < 0 ? -lo : 0;                                        // (the slice starts at its first row's diagonal or du)
    const int64_t qmax = 3 * rows + 1 - ((row0 + rows == n) ? 2 : 0);             // (the last local row's c is not in the slice)
    // (a uniform base + a 32-bit lane offset: one address register per load, not two)
    const real_t *__restrict__ base = src.p0 + (lo + qmin);
    const unsigned qspan = (unsigned)(qmax - qmin);
#pragma unroll
    for (int j = 0; j < kTriRawPer; ++j) {
        const int q = j * kBlock + (int)threadIdx.x - (int)qmin;
        L.g[j] = base[q < 0 ? 0u : (unsigned)q > qspan ? qspan : (unsigned)q];      // (unconditional, from a clamped position)
    }
    const real_t *__restrict__ rbase = src.rhs + row0;
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {                    // the right-hand side's loads are in flight behind them
        const unsigned r = (unsigned)(j * kBlock) + threadIdx.x;
        L.gr[j] = (double)rbase[r < (unsigned)rows ? r : (unsigned)rows - 1u];
    }
}
template <int NRHS>
__device__ __forceinline__ void tri_land_rows_csc(const SrcUser &src, int64_t n, int64_t row0, TriLoadsCsc &L, double *lds, SrcRegs<NRHS> &R)
{
    const int64_t rows = (n - row0 < kTriTileRows) ? n - row0 : kTriTileRows;
    const int64_t lo = 3 * (src.g0 + row0) - src.e0 - 2;
    const int64_t qmin = lo < 0 ? -lo : 0;
    const int64_t qmax = 3 * rows + 1 - ((row0 + rows == n) ? 2 : 0);
    // (the selects only after every load has been issued)
#pragma unroll
    for (int j = 0; j < kTriRawPer; ++j) {
        const int64_t q = (int64_t)j * kBlock + threadIdx.x;
        if (!(q >= qmin && q <= qmax)) L.g[j] = (real_t)0;
    }
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
        const int r = j * kBlock + (int)threadIdx.x;
        L.gr[j] = r < rows ? src.rhs_fix(row0 + r, L.gr[j]) : 0.0;
    }
#pragma unroll
    for (int j = 0; j < kTriRawPer; ++j) {
        const int q = j * kBlock + (int)threadIdx.x;
        if (q < kTriRawVals) lds[q + q / 24] = (double)L.g[j];
    }
    __syncthreads();
    const int base = 25 * (int)threadIdx.x;               // slot of raw value 24 t
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
        const int64_t i = row0 + (int64_t)threadIdx.x * kChunk + j;
        const double Aa = lds[base + 3 * j + (3 * j >= 24 ? 1 : 0)];
        const double Ab = lds[base + 3 * j + 2 + (3 * j + 2 >= 24 ? 1 : 0)];
        const double Ac = lds[base + 3 * j + 4 + (3 * j + 4 >= 24 ? 1 : 0)];
        R.a[j] = src.beta * (i > 0 ? Aa : 0.0);
        R.b[j] = src.alpha + src.beta * Ab;
        R.c[j] = src.beta * (i + 1 < n ? Ac : 0.0);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
        const int r = j * kBlock + (int)threadIdx.x;
        if (r < rows) lds[r + (r >> 3)] = L.gr[j];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kChunk; ++q) R.d[0][q] = lds[9 * (int)threadIdx.x + q];
    if (NRHS > 1) {
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
            const int64_t i = row0 + (int64_t)threadIdx.x * kChunk + q;
            R.d[NRHS > 1 ? 1 : 0][q] = i == 0 ? 1.0 : 0.0;
            R.d[NRHS > 2 ? 2 : 0][q] = i == n - 1 ? 1.0 : 0.0;
        }
    }
}
template <int NRHS>
__device__ __forceinline__ void tri_fetch_rows_csc(const SrcUser &src, int64_t n, int64_t row0, double *lds, SrcRegs<NRHS> &R)
{
    TriLoadsCsc L;
    tri_issue_rows_csc(src, n, row0, L);
    tri_land_rows_csc<NRHS>(src, n, row0, L, lds, R);
}
template <int NRHS>
__global__ void __launch_bounds__(kBlock) k_tri_reduce_csc(SrcUser src, int64_t n, double *__restrict__ sum, int64_t nc)
{
    __shared__ double lds[kTriRawSlots];
    SrcRegs<NRHS> R;
    tri_fetch_rows_csc<NRHS>(src, n, (int64_t)blockIdx.x * kTriTileRows, lds, R);
    tri_guard_rows<NRHS>(R, n, (int64_t)blockIdx.x * kTriTileRows, src.nd_flag);
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= nc) return;
    tri_reduce_chunk<SrcRegs<NRHS>, NRHS>(R, n, SumGlobal{sum, nc}, k);
}

// The top level (n <= kTop): parallel cyclic reduction in LDS, NRHS right-hand sides.  sol[q * n + i].
template <typename Src, int NRHS>
__global__ void __launch_bounds__(kBlock) k_tri_top(Src src, int n, double *__restrict__ sol)
{
    __shared__ double A[2][kTop], B[2][kTop], Cc[2][kTop], D[2][NRHS][kTop];
    for (int i = threadIdx.x; i < n; i += kBlock) {
        const TriRow r = src.template load<NRHS>(i);
        if constexpr (Src::kUnitRhs)       // (a system small enough to start at the top level: the dominance guard of level 0 sits here)
            if (src.nd_flag && !(fabs(r.b) >= fabs(r.a) + fabs(r.c))) atomicOr(src.nd_flag, 1);
        A[0][i] = r.a; B[0][i] = r.b; Cc[0][i] = r.c;
#pragma unroll
        for (int q = 0; q < NRHS; ++q) D[0][q][i] = r.d[q];
    }
    __syncthreads();
    int cur = 0;
    for (int st = 1; st < n; st <<= 1) {
        const int nxt = cur ^ 1;
        for (int i = threadIdx.x; i < n; i += kBlock) {
            const bool hl = i - st >= 0, hr = i + st < n;
            const double k1 = hl ? A[cur][i] / B[cur][i - st] : 0.0, k2 = hr ? Cc[cur][i] / B[cur][i + st] : 0.0;
            A[nxt][i] = hl ? -A[cur][i - st] * k1 : 0.0;
            Cc[nxt][i] = hr ? -Cc[cur][i + st] * k2 : 0.0;
            B[nxt][i] = B[cur][i] - (hl ? Cc[cur][i - st] * k1 : 0.0) - (hr ? A[cur][i + st] * k2 : 0.0);
#pragma unroll
            for (int q = 0; q < NRHS; ++q)
                D[nxt][q][i] = D[cur][q][i] - (hl ? D[cur][q][i - st] * k1 : 0.0) - (hr ? D[cur][q][i + st] * k2 : 0.0);
        }
        __syncthreads();
        cur = nxt;
    }
    for (int i = threadIdx.x; i < n; i += kBlock)
#pragma unroll
        for (int q = 0; q < NRHS; ++q) sol[(int64_t)q * n + i] = D[cur][q][i] / B[cur][i];
}

// Back-substitution of one level (1 right-hand side): z = solution of the level above (the chunks' last unknowns).
// OutT: double for the internal levels, real_t for level 0 (the caller's y).
template <typename T> struct TriGlobalOut {
    T *y;
    __device__ __forceinline__ void put(int64_t i, double v) const { y[i] = (T)v; }
};
struct TriLdsOut {
    double *sd;
    int64_t row0;
    __device__ __forceinline__ void put(int64_t i, double v) const { const int r = (int)(i - row0); sd[r + (r >> 3)] = v; }
};
template <typename Src, typename Out>
__device__ __forceinline__ void tri_backsub_chunk_v(const Src &src, int64_t n, double zl, double zr, int64_t k, const Out &y)
{
    const int64_t s = k * kChunk;
    const int m = (int)((n - s < kChunk) ? n - s : kChunk);
    double cp[kChunk], dp[kChunk];
    // Thomas on the interior rows 0 .. m-2 with the two boundary values moved to the right-hand side
#pragma unroll
    for (int i = 0; i < kChunk - 1; ++i)
        if (i < m - 1) {
            const TriRow r = src.template loadl<1>(i, s + i);
            double di = r.d[0];
            if (i == 0) di -= r.a * zl;
            if (i == m - 2) di -= r.c * zr;
            if (i == 0) {
                cp[0] = r.c / r.b;
                dp[0] = di / r.b;
            } else {
                const double den = r.b - r.a * cp[i - 1];
                cp[i] = r.c / den;
                dp[i] = (di - r.a * dp[i - 1]) / den;
            }
        }
    double yn = zr;
    y.put(s + m - 1, zr);
#pragma unroll
    for (int i = kChunk - 2; i >= 0; --i)
        if (i < m - 1) {
            yn = (i == m - 2) ? dp[i] : dp[i] - cp[i] * yn;
            y.put(s + i, yn);
        }
}
template <typename Src, typename Out>
__device__ __forceinline__ void tri_backsub_chunk(const Src &src, int64_t n, const double *__restrict__ z, int64_t k, const Out &y)
{
    tri_backsub_chunk_v(src, n, k > 0 ? z[k - 1] : 0.0, z[k], k, y);
}
// rows fetched through LDS (see k_tri_reduce); the solution leaves through LDS too, with dense stores
template <typename Src, typename OutT>
__global__ void __launch_bounds__(kBlock) k_tri_backsub(Src src, int64_t n, const double *__restrict__ z, int64_t nc,
                                                        OutT *__restrict__ y)
{
    __shared__ double lds[kTriPitch];
    const int64_t row0 = (int64_t)blockIdx.x * kTriTileRows;
    SrcRegs<1> R;
    tri_fetch_rows<Src, 1>(src, n, row0, lds, R);
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    __syncthreads();                                      // every thread has copied the last array out of LDS
    if (k < nc) tri_backsub_chunk(R, n, z, k, TriLdsOut{lds, row0});
    __syncthreads();
    const int64_t rows = (n - row0 < kTriTileRows) ? n - row0 : kTriTileRows;
    bool poison = false;
    if constexpr (Src::kUnitRhs) poison = src.refuse && src.nd_flag && *src.nd_flag != 0;      // (level 0: the reduction has checked every row)
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
        const int r = j * kBlock + (int)threadIdx.x;
        if (r < rows) y[row0 + r] = poison ? (OutT)__builtin_nan("") : (OutT)lds[r + (r >> 3)];
    }
}

// level 0 on CSC nzval (tri_fetch_rows_csc).  The tiles are walked from the LAST to the first: the reduction read J and b front to
// back just before, so their tail is what the Infinity Cache still holds.
template <typename OutT>
__global__ void __launch_bounds__(kBlock) k_tri_backsub_csc(SrcUser src, int64_t n, const double *__restrict__ z, int64_t nc,
                                                            OutT *__restrict__ y)
{
    __shared__ double lds[kTriRawSlots];
    const int64_t tile = (int64_t)gridDim.x - 1 - blockIdx.x;
    const int64_t row0 = tile * kTriTileRows;
    SrcRegs<1> R;
    tri_fetch_rows_csc<1>(src, n, row0, lds, R);
    const int64_t k = tile * kBlock + threadIdx.x;
    __syncthreads();
    if (k < nc) tri_backsub_chunk(R, n, z, k, TriLdsOut{lds, row0});
    __syncthreads();
    const int64_t rows = (n - row0 < kTriTileRows) ? n - row0 : kTriTileRows;
    const bool poison = src.refuse && src.nd_flag && *src.nd_flag != 0;
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
        const int r = j * kBlock + (int)threadIdx.x;
        if (r < rows) y[row0 + r] = poison ? (OutT)__builtin_nan("") : (OutT)lds[r + (r >> 3)];
    }
}

// ---- Two levels per launch (one right-hand side) ----------------------------------------------------------------------------
// The reduction of a level writes 8 doubles per chunk and the next level's reads them back: at N = 10^7 that is 80 MB out and in
// again (twice) between levels 0 and 1, and seven launches of ~7 us each further up (profiles/r05_h_solve_trace.md).  Here a
// workgroup reduces its tile of 2048 rows to 256 chunk summaries IN LDS, forms the 256 rows of the next level from them (the row
// of a tile's last chunk needs the first-row equation of the NEXT tile's first chunk: those 8 rows are fetched as a halo, one
// value per lane, and swept by one thread), and 32 threads reduce these to 32 summaries two levels up -- the only thing written
// (+ 4 doubles per tile: its chunk 0's first-row equation, which is the halo of the tile before in the back-substitution).
// The back-substitution re-derives the summaries from the rows it fetches anyway (arithmetic instead of 240 MB of traffic),
// solves the 256 middle-level rows in LDS with the 32 + 1 values from two levels up, then the tile's own rows.
// What made it pay (profiles/r05_p_solve.md): the one-level routines' arithmetic did NOT (91 + 105 us for the two level-0
// launches against 90 + 90: instruction issue); neither did persistent workgroups with the next tile's loads in flight in
// registers (two workgroups per compute unit: 82 + 87 us, 3.9 TB/s); one tile per workgroup with the lean arithmetic below at
// three workgroups per compute unit does: 73 + 70 us.
constexpr int kSup = kBlock / kChunk;            // rows two levels up per tile
constexpr int kSVals = 8;                        // of a summary's 12 values, the 8 one right-hand side uses
template <int P> struct SumLds {                 // summary table in LDS: S[slot][chunk - k0], pitch P
    double *S;
    int64_t k0;
    __device__ __forceinline__ void put(int v, int64_t k, double x) const { S[(v < 6 ? v : v - 2) * P + (int)(k - k0)] = x; }
};
template <int P> struct TriLdsZ {                // a level's unknowns of one tile: Z[1 + (i - k0)] (Z[0]: the one before the tile)
    double *Z;
    int64_t k0;
    __device__ __forceinline__ void put(int64_t i, double v) const { Z[1 + (int)(i - k0)] = v; }
};
// the halo: lane t < 32 holds coefficient t / 8 of row rowh + t % 8 (issued before the tile's loads, consumed after them)
template <typename Src> __device__ __forceinline__ double tri_halo_issue(const Src &src, int64_t n, int64_t rowh)
{
    const int64_t i = rowh + (threadIdx.x & 7);
    return src.raw((threadIdx.x >> 3) & 3, i < n ? i : n - 1);
}
template <typename Src> __device__ __forceinline__ void tri_halo_store(const Src &src, int64_t n, int64_t rowh, double hv, double *halo)
{
    const int64_t i = rowh + (threadIdx.x & 7);
    if (threadIdx.x < 32) halo[threadIdx.x] = i < n ? src.fix((threadIdx.x >> 3) & 3, i, hv) : 0.0;
}
// The arithmetic of the two-level kernels.  The one-level kernels above spend ~45 instructions per row (two IEEE divisions of ~14
// instructions each, every step under an `i < m` branch) and are, at ~1000 instructions per thread, already close to issue-bound;
// two levels per launch with the same routines measured NO faster than four launches (instruction issue, not memory).  Here:
//   * one reciprocal per row (v_rcp_f64 + two Newton steps: ~1 ulp), shared by the downward sweep, the upward sweep and the
//     back-substitution; explicit fused multiply-adds (the solver has no bit-for-bit reference: LinearAlgebra pivots);
//   * chunks are always 8 rows: rows past the level's end are identity rows (the system extended by y = 0 unknowns), so no step
//     is conditional.
// The sweep works IN PLACE on the rows in registers: a becomes the coupling f to the unknown before the chunk, b its RECIPROCAL;
// what is left is exactly what the back-substitution needs -- no second elimination.
__device__ __forceinline__ double tri_rcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(r, e, r);
}
struct SumNone {
    __device__ __forceinline__ void put(int, int64_t, double) const {}
};
__device__ __forceinline__ void tri_pad_rows(SrcRegs<1> &R, int64_t n, int64_t k)
{
    const int64_t s = k * kChunk;
    if (s + kChunk > n) {                                 // (the level's last chunk only)
#pragma unroll
        for (int i = 0; i < kChunk; ++i)
            if (s + i >= n) { R.a[i] = 0; R.b[i] = 1; R.c[i] = 0; R.d[0][i] = 0; }
    }
}
template <typename Sink> __device__ __forceinline__ void tri_sweep(SrcRegs<1> &R, const Sink &sink, int64_t k)
{
    const double b0 = R.b[0];
    double bl = b0;
    R.b[0] = tri_rcp(b0);
#pragma unroll
    for (int i = 1; i < kChunk; ++i) {
        const double mult = R.a[i] * R.b[i - 1];
        R.a[i] = -mult * R.a[i - 1];
        bl = __builtin_fma(-mult, R.c[i - 1], R.b[i]);
        R.d[0][i] = __builtin_fma(-mult, R.d[0][i - 1], R.d[0][i]);
        R.b[i] = tri_rcp(bl);
    }
    sink.put(0, k, R.a[kChunk - 1]);
    sink.put(1, k, bl);
    sink.put(2, k, R.c[kChunk - 1]);
    sink.put(3, k, R.d[0][kChunk - 1]);
    // upward: the first row expressed through the unknown before the chunk (f) and the chunk's last unknown (g)
    double cf = R.a[kChunk - 2], cg = R.c[kChunk - 2], cd = R.d[0][kChunk - 2], cr = R.b[kChunk - 2];
#pragma unroll
    for (int i = kChunk - 3; i >= 0; --i) {
        const double mult = R.c[i] * cr;
        cf = __builtin_fma(-mult, cf, R.a[i]);
        cg = -mult * cg;
        cd = __builtin_fma(-mult, cd, R.d[0][i]);
        cr = R.b[i];
    }
    sink.put(6, k, cf);
    sink.put(7, k, b0);
    sink.put(8, k, cg);
    sink.put(9, k, cd);
}
// the chunk's unknowns from its swept rows  f_i zl + y_i / r_i + c_i y_{i+1} = d_i  and its last unknown zr
template <typename Out>
__device__ __forceinline__ void tri_backsub_swept(const SrcRegs<1> &R, double zl, double zr, int64_t k, const Out &y)
{
    const int64_t s = k * kChunk;
    double yn = zr;
    y.put(s + kChunk - 1, zr);
#pragma unroll
    for (int i = kChunk - 2; i >= 0; --i) {
        yn = __builtin_fma(-R.c[i], yn, __builtin_fma(-R.a[i], zl, R.d[0][i])) * R.b[i];
        y.put(s + i, yn);
    }
}
// the 8 rows of chunk k2 of the middle level, from the summary table of the tile (SrcLevel::load's formulas)
template <int P> __device__ __forceinline__ void tri_level_rows(const double *S, int64_t k0, int64_t nc, int64_t k2, SrcRegs<1> &R)
{
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
        const int64_t k = k2 * kChunk + i;
        const int j = (int)(k - k0);
        if (k < nc) {
            const bool nxt = k + 1 < nc;
            const double t = nxt ? S[2 * P + j] * tri_rcp(S[5 * P + j + 1]) : 0.0;
            R.a[i] = S[0 * P + j];
            R.b[i] = __builtin_fma(-t, nxt ? S[4 * P + j + 1] : 0.0, S[1 * P + j]);
            R.c[i] = nxt ? -t * S[6 * P + j + 1] : 0.0;
            R.d[0][i] = __builtin_fma(-t, nxt ? S[7 * P + j + 1] : 0.0, S[3 * P + j]);
        } else {
            R.a[i] = 0; R.b[i] = 1; R.c[i] = 0; R.d[0][i] = 0;
        }
    }
}
template <typename Src, bool CSC> struct TriTileLoads;
template <typename Src> struct TriTileLoads<Src, false> {
    TriLoads<Src, 1> L;
    __device__ __forceinline__ void issue(const Src &src, int64_t n, int64_t row0) { tri_issue_rows<Src, 1>(src, n, row0, L); }
    __device__ __forceinline__ void land(const Src &src, int64_t n, int64_t row0, double *lds, SrcRegs<1> &R) { tri_land_rows<Src, 1>(src, n, row0, L, lds, R); }
};
template <> struct TriTileLoads<SrcUser, true> {
    TriLoadsCsc L;
    __device__ __forceinline__ void issue(const SrcUser &src, int64_t n, int64_t row0) { tri_issue_rows_csc(src, n, row0, L); }
    __device__ __forceinline__ void land(const SrcUser &src, int64_t n, int64_t row0, double *lds, SrcRegs<1> &R) { tri_land_rows_csc<1>(src, n, row0, L, lds, R); }
};
// WPC: workgroups per compute unit the registers are budgeted for (3 with the CSC fetch buffer's 51 KB of LDS)
template <typename Src, bool CSC, int WPC>
__global__ void __launch_bounds__(kBlock, WPC) k_tri_reduce2(Src src, int64_t n, int64_t nc1, double *__restrict__ sum2, int64_t nc2,
                                                             double *__restrict__ hsum)
{
    constexpr int P = kBlock + 1;
    __shared__ double lds[CSC ? kTriRawSlots : kTriPitch];
    __shared__ double halo[32];
    static_assert(kSVals * P <= kTriPitch, "the summary table reuses the fetch buffer");
    const int64_t tile = blockIdx.x, row0 = tile * kTriTileRows, k0 = tile * kBlock;
    const double hv = tri_halo_issue(src, n, row0 + kTriTileRows);
    TriTileLoads<Src, CSC> G;
    G.issue(src, n, row0);
    SrcRegs<1> R;
    G.land(src, n, row0, lds, R);
    if constexpr (Src::kUnitRhs) tri_guard_rows<1>(R, n, row0, src.nd_flag);
    tri_halo_store(src, n, row0 + kTriTileRows, hv, halo);
    __syncthreads();                                      // every thread has copied the last array out of the buffer
    const SumLds<P> S{lds, k0};
    const int64_t k = k0 + threadIdx.x;
    if (k < nc1) {
        tri_pad_rows(R, n, k);
        tri_sweep(R, S, k);
    }
    if (threadIdx.x == kBlock - 1 && k0 + kBlock < nc1) {                // the halo chunk (R is free again)
#pragma unroll
        for (int i = 0; i < kChunk; ++i) { R.a[i] = halo[i]; R.b[i] = halo[8 + i]; R.c[i] = halo[16 + i]; R.d[0][i] = halo[24 + i]; }
        tri_pad_rows(R, n, k0 + kBlock);
        tri_sweep(R, S, k0 + kBlock);
    }
    __syncthreads();
    const int64_t k2 = tile * kSup + threadIdx.x;
    if (threadIdx.x < kSup && k2 < nc2) {
        tri_level_rows<P>(lds, k0, nc1, k2, R);
        tri_sweep(R, SumGlobal{sum2, nc2}, k2);
    }
    if (threadIdx.x < 4) hsum[tile * 4 + threadIdx.x] = lds[(4 + threadIdx.x) * P];      // this tile's chunk 0: its first-row equation
}
template <typename Src, bool CSC, typename OutT, int WPC>
__global__ void __launch_bounds__(kBlock, WPC) k_tri_backsub2(Src src, int64_t n, int64_t nc1, const double *__restrict__ z2, int64_t nc2,
                                                              OutT *__restrict__ y, const double *__restrict__ hsum)
{
    constexpr int P = kBlock + 1;
    constexpr int kFetch = CSC ? kTriRawSlots : kTriPitch;
    constexpr int kZOff = kSVals * P, kOutOff = (kZOff + P + 7) / 8 * 8;
    constexpr int kLds = kFetch > kOutOff + kTriPitch ? kFetch : kOutOff + kTriPitch;
    __shared__ double lds[kLds];
    bool poison = false;
    if constexpr (Src::kUnitRhs) poison = src.refuse && src.nd_flag && *src.nd_flag != 0;      // (the reduction has checked every row)
    // level 0 walks the tiles from the last to the first (the reduction has just read J and b front to back: the tail is what the
    // Infinity Cache still holds)
    const int64_t ntiles = gridDim.x, tile = Src::kUnitRhs ? ntiles - 1 - blockIdx.x : (int64_t)blockIdx.x;
    const int64_t row0 = tile * kTriTileRows, k0 = tile * kBlock;
    const int64_t k2 = tile * kSup + (threadIdx.x & (kSup - 1)), k2c = k2 < nc2 ? k2 : nc2 - 1;
    // (needed two stages further down: in flight with the rows)
    const double zr2 = z2[k2c], zl2 = z2[k2c > 0 ? k2c - 1 : 0];
    const double hs = hsum[(tile + 1 < ntiles ? tile + 1 : tile) * 4 + (threadIdx.x & 3)];
    TriTileLoads<Src, CSC> G;
    G.issue(src, n, row0);
    SrcRegs<1> R;
    G.land(src, n, row0, lds, R);
    __syncthreads();
    const SumLds<P> S{lds, k0};
    const int64_t k = k0 + threadIdx.x;
    if (k < nc1) {
        tri_pad_rows(R, n, k);
        tri_sweep(R, S, k);
    }
    if (threadIdx.x < 4) lds[(4 + threadIdx.x) * P + kBlock] = hs;          // chunk 0 of the next tile (from the reduction)
    __syncthreads();
    double *Z = lds + kZOff;
    if (threadIdx.x < kSup && k2 < nc2) {
        SrcRegs<1> R2;
        tri_level_rows<P>(lds, k0, nc1, k2, R2);
        tri_sweep(R2, SumNone{}, k2);
        tri_backsub_swept(R2, k2 > 0 ? zl2 : 0.0, zr2, k2, TriLdsZ<P>{Z, k0});
    }
    if (threadIdx.x == 0) Z[0] = k2 > 0 ? zl2 : 0.0;      // the unknown before the tile = the last unknown of the chunk before
    __syncthreads();
    if (k < nc1) tri_backsub_swept(R, Z[threadIdx.x], Z[threadIdx.x + 1], k, TriLdsOut{lds + kOutOff, row0});
    __syncthreads();
    const int64_t rows = (n - row0 < kTriTileRows) ? n - row0 : kTriTileRows;
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
        const int r = j * kBlock + (int)threadIdx.x;
        if (r < rows) y[row0 + r] = poison ? (OutT)__builtin_nan("") : (OutT)lds[kOutOff + r + (r >> 3)];
    }
}
// The top of the two-level schedule: a level of at most kTop2 rows, one workgroup: a chunk per thread (swept in registers, the
// sweeps' arithmetic) -> <= 512 rows -> parallel cyclic reduction in LDS (one reciprocal per row and step) -> the chunks' rows from
// the swept registers.  Rows come straight from the summaries below (a few thousand rows: latency, not bandwidth).
constexpr int kTop2Block = 512;
constexpr int kTop2 = kTop2Block * kChunk;
struct TriBoundedOut {
    double *y;
    int64_t n;
    __device__ __forceinline__ void put(int64_t i, double v) const { if (i < n) y[i] = v; }
};
__global__ void __launch_bounds__(kTop2Block) k_tri_top2(SrcLevel src, int n, double *__restrict__ sol)
{
    constexpr int P = kTop2Block + 1;
    __shared__ double S[kSVals * P];
    __shared__ double A[2][kTop2Block], RB[2][kTop2Block], Cc[2][kTop2Block], D[2][kTop2Block];
    __shared__ double Z[P];
    const int k = threadIdx.x, nc = (n + kChunk - 1) / kChunk;
    SrcRegs<1> R;
    if (k < nc) {
        // (SrcLevel::load's formulas with the sweeps' reciprocal; every load of the chunk issued before the first use)
        const double *sum = src.sum;
        const int64_t ncl = src.nc;
        double s0[kChunk], s1[kChunk], s2[kChunk], s3[kChunk], n6[kChunk], n7[kChunk], n8[kChunk], n9[kChunk];
#pragma unroll
        for (int i = 0; i < kChunk; ++i) {
            const int64_t r = (int64_t)k * kChunk + i, rc = r < n ? r : n - 1, rn = rc + 1 < ncl ? rc + 1 : rc;
            s0[i] = sum[0 * ncl + rc]; s1[i] = sum[1 * ncl + rc]; s2[i] = sum[2 * ncl + rc]; s3[i] = sum[3 * ncl + rc];
            n6[i] = sum[6 * ncl + rn]; n7[i] = sum[7 * ncl + rn]; n8[i] = sum[8 * ncl + rn]; n9[i] = sum[9 * ncl + rn];
        }
#pragma unroll
        for (int i = 0; i < kChunk; ++i) {
            const int64_t r = (int64_t)k * kChunk + i;
            const bool nxt = r + 1 < ncl;
            const double t = nxt ? s2[i] * tri_rcp(n7[i]) : 0.0;
            R.a[i] = s0[i];
            R.b[i] = __builtin_fma(-t, nxt ? n6[i] : 0.0, s1[i]);
            R.c[i] = nxt ? -t * n8[i] : 0.0;
            R.d[0][i] = __builtin_fma(-t, nxt ? n9[i] : 0.0, s3[i]);
        }
        tri_pad_rows(R, n, k);
        tri_sweep(R, SumLds<P>{S, 0}, k);
    }
    __syncthreads();
    double Bk = 1.0;
    if (k < nc) {
        const bool nxt = k + 1 < nc;
        const double t = nxt ? S[2 * P + k] * tri_rcp(S[5 * P + k + 1]) : 0.0;
        Bk = __builtin_fma(-t, nxt ? S[4 * P + k + 1] : 0.0, S[1 * P + k]);
        A[0][k] = S[0 * P + k];
        Cc[0][k] = nxt ? -t * S[6 * P + k + 1] : 0.0;
        D[0][k] = __builtin_fma(-t, nxt ? S[7 * P + k + 1] : 0.0, S[3 * P + k]);
        RB[0][k] = tri_rcp(Bk);
    }
    __syncthreads();
    int cur = 0;
    for (int st = 1; st < nc; st <<= 1) {
        const int nxt = cur ^ 1;
        if (k < nc) {
            const bool hl = k - st >= 0, hr = k + st < nc;
            const int il = hl ? k - st : k, ir = hr ? k + st : k;
            const double k1 = hl ? A[cur][k] * RB[cur][il] : 0.0, kk2 = hr ? Cc[cur][k] * RB[cur][ir] : 0.0;
            const double a2 = -A[cur][il] * k1, c2 = -Cc[cur][ir] * kk2;
            Bk = __builtin_fma(-A[cur][ir], kk2, __builtin_fma(-Cc[cur][il], k1, Bk));
            const double d2 = __builtin_fma(-D[cur][ir], kk2, __builtin_fma(-D[cur][il], k1, D[cur][k]));
            A[nxt][k] = hl ? a2 : 0.0;
            Cc[nxt][k] = hr ? c2 : 0.0;
            D[nxt][k] = d2;
            RB[nxt][k] = tri_rcp(Bk);
        }
        __syncthreads();
        cur = nxt;
    }
    if (k < nc) Z[1 + k] = D[cur][k] * RB[cur][k];
    if (k == 0) Z[0] = 0.0;
    __syncthreads();
    if (k < nc) tri_backsub_swept(R, Z[k], Z[k + 1], k, TriBoundedOut{sol, n});
}

// Phase A epilogue: the six tip values (first / last local row of g, v, w).  The last row is the last unknown of every
// level (= the top level's last); the first row needs chunk 0 of every level back-substituted: one thread.
// lev_sum[l] = summary written by level l's reduction (defines level l+1), lev_n[l] = rows of level l; top_sol = solution
// of the top level (3 x n_top).  Level 0 rows come from `src0`.
struct TipArgs {
    const double *sum[kMaxLevels];
    int64_t n[kMaxLevels];
    int nlev;            // levels 0 .. nlev-1; level nlev-1 is the top (solved by PCR)
};
// couplings this rank holds for its neighbours: beta*A[g0-1, g0] (column g0) and beta*A[g0+n, g0+n-1] (column g0+n-1)
__global__ void k_tri_neighbour_couplings(SrcUser s, double *__restrict__ out2)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double cprev = 0.0, anext = 0.0;
    const int64_t j0 = s.g0, j1 = s.g0 + s.n;
    if (s.layout == FD_TRI_DIAGONALS) {
        const int64_t du0 = j0 > 0 ? j0 - 1 : 0;
        if (j0 > 0) cprev = s.beta * (double)s.p2[j0 - 1 - du0];      // du[j0-1] = A[j0-1, j0]
        if (j1 < s.N) anext = s.beta * (double)s.p0[j1 - 1 - j0];      // dl[j1-1] = A[j1, j1-1]
    } else {
        if (j0 > 0) cprev = s.beta * (double)s.p0[3 * j0 - 1 - s.e0];
        if (j1 < s.N) anext = s.beta * (double)s.p0[3 * j1 - 2 - s.e0];
    }
    out2[0] = cprev;
    out2[1] = anext;
}
__global__ void k_tri_tips(SrcUser src0, TipArgs ta, const double *__restrict__ top_sol,
                                          double *__restrict__ packet, const double *__restrict__ cpl)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int top = ta.nlev - 1;
    const int64_t ntop = ta.n[top];
    double first[3], last[3];
    for (int q = 0; q < 3; ++q) {
        last[q] = top_sol[(int64_t)q * ntop + ntop - 1];
        first[q] = top_sol[(int64_t)q * ntop + 0];
    }
    for (int l = top - 1; l >= 0; --l) {
        const int64_t n = ta.n[l];
        const int m = (int)(n < kChunk ? n : kChunk);
        if (m == 1) continue;
        TriRow r[kChunk];
        for (int i = 0; i < m - 1; ++i) {
            if (l == 0) r[i] = src0.load<3>(i);
            else { SrcLevel sl{ta.sum[l - 1], ta.n[l]}; r[i] = sl.load<3>(i); }
        }
        for (int q = 0; q < 3; ++q) {
            double cp[kChunk], dp[kChunk];
            for (int i = 0; i < m - 1; ++i) {
                double di = r[i].d[q];
                if (i == m - 2) di -= r[i].c * first[q];
                if (i == 0) { cp[0] = r[0].c / r[0].b; dp[0] = di / r[0].b; }
                else { const double den = r[i].b - r[i].a * cp[i - 1]; cp[i] = r[i].c / den; dp[i] = (di - r[i].a * dp[i - 1]) / den; }
            }
            double yn = dp[m - 2];
            for (int i = m - 3; i >= 0; --i) yn = dp[i] - cp[i] * yn;
            first[q] = yn;
        }
    }
    packet[0] = first[0]; packet[1] = last[0];
    packet[2] = first[1]; packet[3] = last[1];
    packet[4] = first[2]; packet[5] = last[2];
    packet[6] = cpl[0];
    packet[7] = cpl[1];
    // a rank whose rows failed the dominance guard REFUSES for everybody: its packet carries NaN, so every rank's interface system --
    // and with it every rank's y -- is poisoned, and every rank's status says so (k_tri_interface).  (Until round 6 only the rank that
    // owned the offending rows refused; its neighbours finished with interface values from its unreliable elimination.)
    if (src0.refuse && src0.nd_flag && *src0.nd_flag != 0)
        for (int k = 0; k < 6; ++k) packet[k] = __longlong_as_double(0x7FF8000000000000ll);
}

// Phase B: the interface system of all ranks (unknowns p_r = first, q_r = last local value of rank r), solved by one
// thread with banded Gaussian elimination; writes adj = {A_r * q_{r-1}, C_r * p_{r+1}} for THIS rank.
//   p_r + A_r v_f q_{r-1} + C_r w_f p_{r+1} = g_f ,   q_r + A_r v_l q_{r-1} + C_r w_l p_{r+1} = g_l
//   A_r = packet[r-1][7] (the previous rank knows it), C_r = packet[r+1][6].
constexpr int kMaxRanks = 1024;
__global__ void k_tri_interface(const double *__restrict__ packets, int W, int rank, double *__restrict__ adj,
                                double *__restrict__ work /* 2W x 6 doubles */, int *__restrict__ nd_flag)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n = 2 * W;
    // (a peer that refused -- NaN in its packet -- refuses for this rank too: status bit 2, and the NaN runs through the system below)
    if (nd_flag) {
        bool peer_nd = false;
        for (int r = 0; r < W; ++r)
            for (int k = 0; k < 6; ++k) peer_nd = peer_nd || packets[(size_t)r * kPacket + k] != packets[(size_t)r * kPacket + k];
        if (peer_nd) atomicOr(nd_flag, 2);
    }
    // band storage: row i, columns i-2 .. i+2 -> work[i*6 + (j - i + 2)], rhs at work[i*6 + 5]
    for (int i = 0; i < n * 6; ++i) work[i] = 0.0;
    for (int r = 0; r < W; ++r) {
        const double *pk = packets + (size_t)r * kPacket;
        const double Ar = r > 0 ? packets[(size_t)(r - 1) * kPacket + 7] : 0.0;
        const double Cr = r + 1 < W ? packets[(size_t)(r + 1) * kPacket + 6] : 0.0;
        const int ip = 2 * r, iq = 2 * r + 1;
        work[ip * 6 + 2] = 1.0;  work[ip * 6 + 5] = pk[0];
        work[iq * 6 + 2] = 1.0;  work[iq * 6 + 5] = pk[1];
        if (r > 0) { work[ip * 6 + 1] = Ar * pk[2]; work[iq * 6 + 0] = Ar * pk[3]; }          // * q_{r-1} (column 2r-1)
        if (r + 1 < W) { work[ip * 6 + 4] = Cr * pk[4]; work[iq * 6 + 3] = Cr * pk[5]; }      // * p_{r+1} (column 2r+2)
    }
    for (int i = 0; i < n; ++i) {            // elimination, no pivoting (unit diagonal, spikes of a dominant T are < 1)
        const double piv = work[i * 6 + 2];
        for (int k = 1; k <= 2 && i + k < n; ++k) {
            const double mult = work[(i + k) * 6 + 2 - k] / piv;
            if (mult == 0.0) continue;
            for (int j = 0; j <= 2 && i + j < n; ++j) {
                const int col = 2 - k + j;
                if (col <= 4) work[(i + k) * 6 + col] -= mult * work[i * 6 + 2 + j];
            }
            work[(i + k) * 6 + 5] -= mult * work[i * 6 + 5];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double v = work[i * 6 + 5];
        for (int j = 1; j <= 2 && i + j < n; ++j) v -= work[i * 6 + 2 + j] * work[(i + j) * 6 + 5];
        work[i * 6 + 5] = v / work[i * 6 + 2];
    }
    const double Ar = rank > 0 ? packets[(size_t)(rank - 1) * kPacket + 7] : 0.0;
    const double Cr = rank + 1 < W ? packets[(size_t)(rank + 1) * kPacket + 6] : 0.0;
    adj[0] = rank > 0 ? Ar * work[(2 * rank - 1) * 6 + 5] : 0.0;          // A_r * q_{r-1}
    adj[1] = rank + 1 < W ? Cr * work[(2 * rank + 2) * 6 + 5] : 0.0;      // C_r * p_{r+1}
}

}  // namespace fdjac

struct fd_tridiag_solver {
    fd_ctx *ctx = nullptr;
    int layout = 0;
    int64_t N = 0, g0 = 0, n = 0, e0 = 0;
    int nlev = 0;
    int64_t lev_n[fdjac::kMaxLevels] = {0};
    double *lev_sum[fdjac::kMaxLevels] = {nullptr};   // summary written by level l's reduction (level nlev-1 has none)
    double *lev_sol[fdjac::kMaxLevels] = {nullptr};   // solution of level l >= 1 (3 columns at the top level)
    double *lev_halo[fdjac::kMaxLevels] = {nullptr};  // two levels per launch: chunk 0's first-row equation of every tile of level l (4 doubles)
    double *packets = nullptr;                        // kMaxRanks x kPacket (the all-gather buffer)
    double *adj = nullptr, *cpl = nullptr, *work = nullptr;
    int *status = nullptr;                            // device word: bit 0 = the last solve met a row that is not diagonally dominant
    int refuse = 1;                                   // fd_tridiag_solver_set_policy: such a solve writes NaN (default) / its solution anyway
};

using namespace fdjac;

static SrcUser make_src(const fd_tridiag_solver *s, double alpha, double beta, const void *const *J, const void *rhs,
                        const double *adj)
{
    SrcUser u;
    u.layout = s->layout;
    u.p0 = (const real_t *)J[0];
    u.p1 = s->layout == FD_TRI_DIAGONALS ? (const real_t *)J[1] : nullptr;
    u.p2 = s->layout == FD_TRI_DIAGONALS ? (const real_t *)J[2] : nullptr;
    u.rhs = (const real_t *)rhs;
    u.adj = adj;
    u.g0 = s->g0; u.n = s->n; u.N = s->N; u.e0 = s->e0;
    u.alpha = alpha; u.beta = beta;
    u.nd_flag = s->status;
    u.refuse = s->refuse;
    return u;
}

template <int NRHS> static int tri_reduce_all(fd_tridiag_solver *s, const SrcUser &u)
{
    hipStream_t st = s->ctx->stream;
    for (int l = 0; l + 1 < s->nlev; ++l) {
        const int64_t nc = s->lev_n[l + 1];
        const unsigned g = (unsigned)((nc + kBlock - 1) / kBlock);
        if (l == 0 && u.layout == FD_TRI_CSC)
            hipLaunchKernelGGL((k_tri_reduce_csc<NRHS>), dim3(g), dim3(kBlock), 0, st, u, s->lev_n[0], s->lev_sum[0], nc);
        else if (l == 0)
            hipLaunchKernelGGL((k_tri_reduce<SrcUser, NRHS>), dim3(g), dim3(kBlock), 0, st, u, s->lev_n[0], s->lev_sum[0], nc);
        else
            hipLaunchKernelGGL((k_tri_reduce<SrcLevel, NRHS>), dim3(g), dim3(kBlock), 0, st, SrcLevel{s->lev_sum[l - 1], s->lev_n[l]},
                               s->lev_n[l], s->lev_sum[l], nc);
    }
    const int top = s->nlev - 1;
    if (top == 0)
        hipLaunchKernelGGL((k_tri_top<SrcUser, NRHS>), dim3(1), dim3(kBlock), 0, st, u, (int)s->lev_n[0], s->lev_sol[0]);
    else
        hipLaunchKernelGGL((k_tri_top<SrcLevel, NRHS>), dim3(1), dim3(kBlock), 0, st, SrcLevel{s->lev_sum[top - 1], s->lev_n[top]},
                           (int)s->lev_n[top], s->lev_sol[top]);
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

// One right-hand side: levels `from` and up two per launch (k_tri_reduce2 / k_tri_top2 / k_tri_backsub2), the levels below one per
// launch (from = 2: a test hook that keeps the one-level kernels on the two big levels).  N = 10^7: 5 launches instead of 13,
// 235 -> 185 us (profiles/r05_p_solve.md).
static int tri_two_level_from(const fd_tridiag_solver *s)
{
    const char *e = fdjac::test_switch("FDJAC_SOLVE_TWO_LEVEL");      // -1: never, 0 / 2: from that level
    const int from = (e && *e) ? atoi(e) : 0;
    return (from == 0 || from == 2) && from + 2 < s->nlev && s->lev_n[from] > kTop2 ? from : -1;
}
static int tri_solve_two_level(fd_tridiag_solver *s, const SrcUser &u, real_t *y, int from)
{
    hipStream_t st = s->ctx->stream;
    const bool csc = u.layout == FD_TRI_CSC;
    for (int l = 0; l < from; ++l) {                      // one level per launch
        const int64_t nc = s->lev_n[l + 1];
        const unsigned g = (unsigned)((nc + kBlock - 1) / kBlock);
        if (l == 0 && csc) hipLaunchKernelGGL((k_tri_reduce_csc<1>), dim3(g), dim3(kBlock), 0, st, u, s->lev_n[0], s->lev_sum[0], nc);
        else if (l == 0) hipLaunchKernelGGL((k_tri_reduce<SrcUser, 1>), dim3(g), dim3(kBlock), 0, st, u, s->lev_n[0], s->lev_sum[0], nc);
        else hipLaunchKernelGGL((k_tri_reduce<SrcLevel, 1>), dim3(g), dim3(kBlock), 0, st, SrcLevel{s->lev_sum[l - 1], s->lev_n[l]}, s->lev_n[l], s->lev_sum[l], nc);
    }
    int l = from;
    for (; s->lev_n[l] > kTop2; l += 2) {                 // (levels l + 1, l + 2 exist: the schedule runs down to <= kTop < kTop2 / 8)
        const int64_t n = s->lev_n[l], nc1 = s->lev_n[l + 1], nc2 = s->lev_n[l + 2];
        const unsigned g = (unsigned)((nc1 + kBlock - 1) / kBlock);
        double *hs = s->lev_halo[l];
        if (l == 0 && csc) hipLaunchKernelGGL((k_tri_reduce2<SrcUser, true, 3>), dim3(g), dim3(kBlock), 0, st, u, n, nc1, s->lev_sum[1], nc2, hs);
        else if (l == 0) hipLaunchKernelGGL((k_tri_reduce2<SrcUser, false, 4>), dim3(g), dim3(kBlock), 0, st, u, n, nc1, s->lev_sum[1], nc2, hs);
        else hipLaunchKernelGGL((k_tri_reduce2<SrcLevel, false, 3>), dim3(g), dim3(kBlock), 0, st, SrcLevel{s->lev_sum[l - 1], n}, n, nc1, s->lev_sum[l + 1], nc2, hs);
    }
    const SrcLevel top{s->lev_sum[l - 1], s->lev_n[l]};
    if (s->lev_n[l] <= kTop) hipLaunchKernelGGL((k_tri_top<SrcLevel, 1>), dim3(1), dim3(kBlock), 0, st, top, (int)s->lev_n[l], s->lev_sol[l]);
    else hipLaunchKernelGGL(k_tri_top2, dim3(1), dim3(kTop2Block), 0, st, top, (int)s->lev_n[l], s->lev_sol[l]);
    for (l -= 2; l >= from; l -= 2) {
        const int64_t n = s->lev_n[l], nc1 = s->lev_n[l + 1], nc2 = s->lev_n[l + 2];
        const unsigned g = (unsigned)((nc1 + kBlock - 1) / kBlock);
        const double *hs = s->lev_halo[l];
        if (l == 0 && csc) hipLaunchKernelGGL((k_tri_backsub2<SrcUser, true, real_t, 3>), dim3(g), dim3(kBlock), 0, st, u, n, nc1, s->lev_sol[2], nc2, y, hs);
        else if (l == 0) hipLaunchKernelGGL((k_tri_backsub2<SrcUser, false, real_t, 3>), dim3(g), dim3(kBlock), 0, st, u, n, nc1, s->lev_sol[2], nc2, y, hs);
        else hipLaunchKernelGGL((k_tri_backsub2<SrcLevel, false, double, 3>), dim3(g), dim3(kBlock), 0, st, SrcLevel{s->lev_sum[l - 1], n}, n, nc1, s->lev_sol[l + 2], nc2, s->lev_sol[l], hs);
    }
    for (l = from - 1; l >= 0; --l) {                     // one level per launch
        const int64_t nc = s->lev_n[l + 1];
        const unsigned g = (unsigned)((nc + kBlock - 1) / kBlock);
        if (l == 0 && csc) hipLaunchKernelGGL((k_tri_backsub_csc<real_t>), dim3(g), dim3(kBlock), 0, st, u, s->lev_n[0], s->lev_sol[1], nc, y);
        else if (l == 0) hipLaunchKernelGGL((k_tri_backsub<SrcUser, real_t>), dim3(g), dim3(kBlock), 0, st, u, s->lev_n[0], s->lev_sol[1], nc, y);
        else hipLaunchKernelGGL((k_tri_backsub<SrcLevel, double>), dim3(g), dim3(kBlock), 0, st, SrcLevel{s->lev_sum[l - 1], s->lev_n[l]}, s->lev_n[l], s->lev_sol[l + 1], nc, s->lev_sol[l]);
    }
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

static int tri_backsub_all(fd_tridiag_solver *s, const SrcUser &u, real_t *y)
{
    hipStream_t st = s->ctx->stream;
    for (int l = s->nlev - 2; l >= 0; --l) {
        const int64_t nc = s->lev_n[l + 1];
        const unsigned g = (unsigned)((nc + kBlock - 1) / kBlock);
        if (l == 0 && u.layout == FD_TRI_CSC)
            hipLaunchKernelGGL((k_tri_backsub_csc<real_t>), dim3(g), dim3(kBlock), 0, st, u, s->lev_n[0], s->lev_sol[1], nc, y);
        else if (l == 0)
            hipLaunchKernelGGL((k_tri_backsub<SrcUser, real_t>), dim3(g), dim3(kBlock), 0, st, u, s->lev_n[0], s->lev_sol[1], nc, y);
        else
            hipLaunchKernelGGL((k_tri_backsub<SrcLevel, double>), dim3(g), dim3(kBlock), 0, st, SrcLevel{s->lev_sum[l - 1], s->lev_n[l]},
                               s->lev_n[l], s->lev_sol[l + 1], nc, s->lev_sol[l]);
    }
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

__global__ void k_tri_copy_top(const double *__restrict__ sol, int n, FDJAC_REAL *__restrict__ y, const int *__restrict__ nd_flag, int refuse)
{
    const bool poison = refuse && nd_flag && *nd_flag != 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) y[i] = poison ? (FDJAC_REAL)__builtin_nan("") : (FDJAC_REAL)sol[i];
}

extern "C" {

int fd_tridiag_solver_create(fd_ctx *ctx, int64_t N, int64_t row_begin, int64_t row_end, int layout,
                             fd_tridiag_solver **out)
{
    FD_REQUIRE(ctx && out, FD_ERR_ARG, "NULL argument");
    *out = nullptr;
    FD_REQUIRE(layout == FD_TRI_DIAGONALS || layout == FD_TRI_CSC, FD_ERR_ARG, "unknown layout %d", layout);
    FD_REQUIRE(N >= 1, FD_ERR_SHAPE, "N < 1");
    if (row_begin == 0 && row_end == 0) row_end = N;
    FD_REQUIRE(row_begin >= 0 && row_begin < row_end && row_end <= N, FD_ERR_ARG, "rows [%lld,%lld) outside [0,%lld)",
               (long long)row_begin, (long long)row_end, (long long)N);
    FD_HIP_CHECK(hipSetDevice(ctx->device));
    fd_tridiag_solver *s = new (std::nothrow) fd_tridiag_solver();
    FD_REQUIRE(s != nullptr, FD_ERR_NOMEM, "out of host memory");
    s->ctx = ctx; s->layout = layout; s->N = N; s->g0 = row_begin; s->n = row_end - row_begin;
    s->e0 = row_begin > 0 ? 3 * row_begin - 1 : 0;   // first stored value of column row_begin in a tridiagonal CSC
    int64_t n = s->n;
    int l = 0;
    s->lev_n[0] = n;
    while (n > kTop) {
        n = (n + kChunk - 1) / kChunk;
        ++l;
        if (l >= kMaxLevels) { delete s; set_error("too many levels"); return FD_ERR_UNSUPPORTED; }
        s->lev_n[l] = n;
    }
    s->nlev = l + 1;
    bool ok = true;
    for (int k = 0; k + 1 < s->nlev && ok; ++k)
        ok = hipMalloc((void **)&s->lev_sum[k], sizeof(double) * (size_t)kSumVals * (size_t)s->lev_n[k + 1]) == hipSuccess;
    for (int k = (s->nlev > 1 ? 1 : 0); k < s->nlev && ok; ++k)
        ok = hipMalloc((void **)&s->lev_sol[k], sizeof(double) * 3 * (size_t)s->lev_n[k]) == hipSuccess;
    for (int k = 0; k + 2 < s->nlev && ok; ++k)
        ok = hipMalloc((void **)&s->lev_halo[k], sizeof(double) * 4 * (size_t)((s->lev_n[k + 1] + kBlock - 1) / kBlock)) == hipSuccess;
    ok = ok && hipMalloc((void **)&s->packets, sizeof(double) * kPacket * kMaxRanks) == hipSuccess;
    ok = ok && hipMalloc((void **)&s->adj, sizeof(double) * 2) == hipSuccess;
    ok = ok && hipMalloc((void **)&s->cpl, sizeof(double) * 2) == hipSuccess;
    ok = ok && hipMalloc((void **)&s->work, sizeof(double) * 12 * kMaxRanks) == hipSuccess;
    ok = ok && hipMalloc((void **)&s->status, sizeof(int)) == hipSuccess && hipMemset(s->status, 0, sizeof(int)) == hipSuccess;
    if (!ok) {
        fd_tridiag_solver_destroy(s);
        set_error("hipMalloc failed in fd_tridiag_solver_create");
        return FD_ERR_NOMEM;
    }
    *out = s;
    return FD_OK;
}

// bit 0: the last solve met a row with |alpha + beta J[i,i]| < |beta J[i,i-1]| + |beta J[i,i+1]| -- the elimination does not pivot, its
// result is then not guaranteed (LinearAlgebra's Tridiagonal \ pivots).  Synchronises the context's stream.
int fd_tridiag_solver_status(fd_tridiag_solver *s, int *flags_out)
{
    FD_REQUIRE(s && flags_out, FD_ERR_ARG, "NULL argument");
    FD_HIP_CHECK(hipSetDevice(s->ctx->device));
    FD_HIP_CHECK(hipMemcpyAsync(flags_out, s->status, sizeof(int), hipMemcpyDeviceToHost, s->ctx->stream));
    FD_HIP_CHECK(hipStreamSynchronize(s->ctx->stream));
    return FD_OK;
}

int fd_tridiag_solver_set_policy(fd_tridiag_solver *s, int trust_non_dominant)
{
    FD_REQUIRE(s != nullptr, FD_ERR_ARG, "solver is NULL");
    s->refuse = trust_non_dominant ? 0 : 1;
    return FD_OK;
}

int fd_tridiag_solver_destroy(fd_tridiag_solver *s)
{
    if (!s) return FD_OK;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    for (int k = 0; k < kMaxLevels; ++k) {
        if (s->lev_sum[k]) (void)hipFree(s->lev_sum[k]);
        if (s->lev_sol[k]) (void)hipFree(s->lev_sol[k]);
        if (s->lev_halo[k]) (void)hipFree(s->lev_halo[k]);
    }
    for (void *p : {(void *)s->packets, (void *)s->adj, (void *)s->cpl, (void *)s->work, (void *)s->status})
        if (p) (void)hipFree(p);
    delete s;
    return FD_OK;
}

// local solve of T_r y = rhs - adj (adj = nullptr: the rows are the whole system, or the couplings are zero)
static int tri_local_solve(fd_tridiag_solver *s, double alpha, double beta, const void *const *J, const void *rhs,
                           const double *adj, void *y)
{
    const SrcUser u = make_src(s, alpha, beta, J, rhs, adj);
    if (const int from = tri_two_level_from(s); from >= 0) return tri_solve_two_level(s, u, (real_t *)y, from);
    int rc = tri_reduce_all<1>(s, u);
    if (rc) return rc;
    if (s->nlev == 1) {
        hipLaunchKernelGGL(k_tri_copy_top, dim3(1), dim3(kBlock), 0, s->ctx->stream, s->lev_sol[0], (int)s->n, (real_t *)y, (const int *)s->status, s->refuse);
        FD_HIP_CHECK(hipGetLastError());
        return FD_OK;
    }
    return tri_backsub_all(s, u, (real_t *)y);
}

int fd_tridiag_solve_interface(fd_tridiag_solver *s, double alpha, double beta, const void *const *J, const void *rhs,
                               void *packet_dev)
{
    FD_REQUIRE(s && J && rhs && packet_dev, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(s->layout == FD_TRI_CSC ? J[0] != nullptr : J[1] != nullptr, FD_ERR_ARG, "J's values are NULL");   // (dl / du may be empty)
    FD_HIP_CHECK(hipSetDevice(s->ctx->device));
    (void)hipMemsetAsync(s->status, 0, sizeof(int), s->ctx->stream);      // (the dominance guard of this solve)
    const SrcUser u = make_src(s, alpha, beta, J, rhs, nullptr);
    int rc = tri_reduce_all<3>(s, u);
    if (rc) return rc;
    hipStream_t st = s->ctx->stream;
    hipLaunchKernelGGL(k_tri_neighbour_couplings, dim3(1), dim3(64), 0, st, u, s->cpl);
    TipArgs ta;
    ta.nlev = s->nlev;
    for (int l = 0; l < kMaxLevels; ++l) { ta.sum[l] = s->lev_sum[l]; ta.n[l] = s->lev_n[l]; }
    hipLaunchKernelGGL(k_tri_tips, dim3(1), dim3(64), 0, st, u, ta, s->lev_sol[s->nlev - 1], (double *)packet_dev,
                       s->cpl);
    FD_HIP_CHECK(hipGetLastError());
    return FD_OK;
}

int fd_tridiag_solve_finish(fd_tridiag_solver *s, double alpha, double beta, const void *const *J, const void *rhs,
                            const void *packets_dev, int rank, int nranks, void *y)
{
    FD_REQUIRE(s && J && rhs && packets_dev && y, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(s->layout == FD_TRI_CSC ? J[0] != nullptr : J[1] != nullptr, FD_ERR_ARG, "J's values are NULL");
    FD_REQUIRE(nranks >= 1 && nranks <= kMaxRanks && rank >= 0 && rank < nranks, FD_ERR_ARG, "rank %d of %d", rank, nranks);
    FD_HIP_CHECK(hipSetDevice(s->ctx->device));
    hipLaunchKernelGGL(k_tri_interface, dim3(1), dim3(64), 0, s->ctx->stream, (const double *)packets_dev, nranks, rank, s->adj,
                       s->work, s->refuse ? s->status : nullptr);
    FD_HIP_CHECK(hipGetLastError());
    return tri_local_solve(s, alpha, beta, J, rhs, s->adj, y);
}

int fd_tridiag_solve_async(fd_tridiag_solver *s, double alpha, double beta, const void *const *J, const void *rhs, void *y,
                           fd_comm *comm)
{
    FD_REQUIRE(s && J && rhs && y, FD_ERR_ARG, "NULL argument");
    FD_REQUIRE(s->layout == FD_TRI_CSC ? J[0] != nullptr : J[1] != nullptr, FD_ERR_ARG, "J's values are NULL");
    FD_HIP_CHECK(hipSetDevice(s->ctx->device));
    if (!comm) {
        FD_REQUIRE(s->n == s->N, FD_ERR_ARG, "a solver for rows [%lld,%lld) of %lld needs a communicator", (long long)s->g0,
                   (long long)(s->g0 + s->n), (long long)s->N);
        (void)hipMemsetAsync(s->status, 0, sizeof(int), s->ctx->stream);      // (the dominance guard of this solve)
        return tri_local_solve(s, alpha, beta, J, rhs, nullptr, y);
    }
    FD_REQUIRE(fdjac_comm_ctx(comm) == s->ctx, FD_ERR_ARG, "the communicator belongs to another context");
    const int W = fdjac_comm_nranks(comm), r = fdjac_comm_rank(comm);
    FD_REQUIRE(W <= kMaxRanks, FD_ERR_UNSUPPORTED, "more than %d ranks", kMaxRanks);
    int rc = fd_tridiag_solve_interface(s, alpha, beta, J, rhs, s->packets + (size_t)r * kPacket);
    if (rc) return rc;
    rc = fdjac_comm_allgather_f64(comm, s->packets, kPacket);
    if (rc) return rc;
    return fd_tridiag_solve_finish(s, alpha, beta, J, rhs, s->packets, r, W, y);
}

}  // extern "C"

