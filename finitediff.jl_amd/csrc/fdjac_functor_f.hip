// Built-in f! families whose lazy launchers store the Jacobian of a GENERAL sparsity pattern column by column through the plan's
// compact copy of the pattern (fd_csc_store, include/fdjac_device.h; FD_LAZY_CAP_STORE_CSC):
//   FD_F_LAP7    zero-Dirichlet 7-point Laplacian on an nx x ny x nz grid + x[k]^2 x[k+1]   (3-D stencil: offsets +-1, +-nx, +-nx*ny)
//   FD_F_SPARSE  f_r = sum over the entries (r, j) of a GIVEN pattern, ascending j, of w(r, j) phi(x_j)   (any pattern)
// Each family is ONE device functor  T f(r, X)  -- row r of the residual at the point whose coordinate j is X(j) -- used three
// times: by the plain launcher (X(j) = x[j] of a materialised point), by fd_csc_store_cols (X = the colour's point, formed as the
// reference forms it: x[j] + eps_c * (color[j] == c), src/jacobians.jl:562 / 603-604) and, through the same public template, by
// user code (examples/user_csc_store.hip).  The storing launch performs the reference's subtraction, division and assignment
// (src/jacobians.jl:565 / 607, ext/FiniteDiffSparseArraysExt.jl:38-47) on the operands the hand-over path (materialised points ->
// plain f! -> k_decompress_*) would have: same bits.
// Included by fdjac_builtin_f.hip (namespace fdjac, after BuiltinF).

constexpr real_t kSix = 6, kQuarter = 0.25, kEighth = 0.125;
template <bool FAST> __device__ __forceinline__ real_t div_shared(real_t a, real_t b, real_t y);   // (fdjac_builtin_f.hip, below the include)
template <bool FAST> __device__ __forceinline__ real_t div_shared(real_t a, real_t b, real_t y, bool bok);
__device__ __forceinline__ bool div_shared_ok(real_t b);

template <typename T> struct PlainPoint {      // a materialised point
    const T *x;
    __device__ T operator()(int64_t j) const { return x[j]; }
};

// ---- FD_F_LAP7 -----------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T lap7_row(T c, T d, T s, T w, T e, T n, T u)
{
    return ((((((d + s) + w) + e) + n) + u) - kSix * c) + (c * c) * e;
}
struct Lap7F {
    int nx, ny, nz;
    uint64_t m_pl, m_nx;       // fd_magic31(nx * ny), fd_magic31(nx): exact 32-bit divisions by one multiply (N < 2^31)
    template <typename T, class P> __device__ __forceinline__ T row(int64_t k, const P &X) const
    {
        const int pl = nx * ny;
        const int l = (int)fd_div31((uint32_t)k, m_pl), rem = (int)k - l * pl, j = (int)fd_div31((uint32_t)rem, m_nx), i = rem - j * nx;
        const T z = zero_of<T>();
        // (every coordinate is fetched unconditionally -- from a clamped index -- and a neighbour outside the grid selected away: loads
        //  inside per-lane conditionals are waited for one at a time, seven serial round trips per row)
        const bool hd = l > 0, hs = j > 0, hw = i > 0, he = i < nx - 1, hn = j < ny - 1, hu = l < nz - 1;
        const T c = X(k);
        const T vd = X(hd ? k - pl : k), vs = X(hs ? k - nx : k), vw = X(hw ? k - 1 : k), ve = X(he ? k + 1 : k), vn = X(hn ? k + nx : k),
                vu = X(hu ? k + pl : k);
        const T d = hd ? vd : z, s = hs ? vs : z, w = hw ? vw : z, e = he ? ve : z, n = hn ? vn : z, u = hu ? vu : z;
        return lap7_row<T>(c, d, s, w, e, n, u);
    }
    template <class P> __device__ __forceinline__ real_t operator()(long long k, const P &X) const { return row<real_t>(k, X); }
};

// The column-centric storing launch of the 7-point family with the column's neighbourhood IN REGISTERS: thread k loads the 25
// coordinates within L1 distance 2 of grid point k once (what the seven rows k, k +- 1, k +- nx, k +- nx ny read; 0 outside the
// grid, as the residual defines it), and evaluates every stored entry whose row is one of those seven from the window, at
// x + eps e_k (valid colouring: fd_csc_store.valid_coloring) -- 25 loads per column instead of 7 per evaluated row.  Entries of any
// other row (a pattern that is a superset of the stencil) go through the functor.  Same operands, same operations: same bits.
// what the 7-point launcher writes into a plan's note once it has seen that the plan's local columns hold exactly its stencil
__host__ __device__ __forceinline__ unsigned long long lap7_note_key(const Lap7F &f, const fd_csc_store &st)
{
    unsigned long long h = 0x9E3779B97F4A7C15ull;
    const long long v[6] = {f.nx, f.ny, f.nz, st.col_begin, st.col_end, st.N};
    for (int i = 0; i < 6; ++i) { h ^= (unsigned long long)v[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); }
    return h | 1ull;
}
// after a checking launch: no mismatch counted -> the pattern is verified for this (grid, column range)
__global__ void k_lap7_note(unsigned long long *note, unsigned long long key)
{
    if (note[0] != key && note[0] != (key ^ 2ull)) note[0] = note[1] == 0 ? key : (key ^ 2ull);
    note[1] = 0;
}

// (five wavefronts per SIMD: the kernel needs 97 / 100 VGPRs unconstrained -- four wavefronts by eight registers -- and fits 96 with
//  0 / 2 spilled: 193 -> 180 us; six -- 80 VGPRs -- spill the window: 305-317 / 223 us, profiles/r04_y_lap7_taken_apart.md)
template <typename CT, int MODE>
__global__ void __launch_bounds__(kBlock, 5)
k_f_lap7_store_cols(Lap7F f, const real_t *__restrict__ x, const real_t *__restrict__ eps, int c_lo, int c_hi, fd_csc_store st)
{
    // (a stencil column holds at most 7 entries: 448 per wavefront -- half of the general window, twice the resident workgroups)
    constexpr int kCap = 512;
    __shared__ __attribute__((aligned(16))) real_t s_win[kBlock / 64][kCap];
    const int64_t nblk = (st.col_end - st.col_begin + kBlock - 1) / kBlock, blk = fd_xcd_block(blockIdx.x, nblk);
    if (blk >= nblk) return;
    const int64_t kk = st.col_begin + blk * kBlock + threadIdx.x;
    const bool in = kk < st.col_end;
    const int64_t k = in ? kk : st.col_end - 1;      // (lanes beyond the range idle on the last column: they take part in the wave's stores only)
    // (everything that depends on the thread index alone first: the window's loads are in flight while colptr, the colour and the row
    //  indices arrive -- a lane whose column turns out to belong to another colour chunk has loaded in vain, nothing else)
    __shared__ real_t s_eps[32];
    if ((int)threadIdx.x < c_hi - c_lo && threadIdx.x < 32) s_eps[threadIdx.x] = eps[c_lo + threadIdx.x];
    const int nx = f.nx, ny = f.ny, nz = f.nz, pl = nx * ny;
    const int l = (int)fd_div31((uint32_t)k, f.m_pl), rem = (int)k - l * pl, j = (int)fd_div31((uint32_t)rem, f.m_nx), i = rem - j * nx;
    // W(dl, dj, di) = x at grid point (i + di, j + dj, l + dl) for |dl| + |dj| + |di| <= 2, 0 outside the grid
    // (a grid point at least two cells away from every face: no bounds tests -- the guards are ~200 vector instructions per column)
    const bool jl_deep = j >= 2 && j < ny - 2 && l >= 2 && l < nz - 2;
    const bool deep = i >= 2 && i < nx - 2 && jl_deep;
    // Every load below is issued unconditionally and the whole window is waited for ONCE.  (Loads inside per-lane conditionals
    // are waited for one by one at the joins: 25 + 7 serial memory round trips made a forward-difference wavefront live 14 us,
    // profiles/r04_y_lap7_taken_apart.md.)  A wavefront whose grid points are two cells from the j and l faces -- the common case --
    // adds wave-uniform offsets to the pointer (scalar unit) and indexes with ONE 32-bit lane offset: every address is inside the
    // array, also where the wavefront runs across the end of a grid row (one wavefront in three at nx = 200); there the twelve
    // coordinates with an i offset are selected to 0 where they lie outside the grid.  Other wavefronts load from a clamped address
    // and select 0 for the coordinates outside the grid.
    const bool wave_fast = __all(jl_deep) && st.N < ((long long)1 << 28);
    const uint32_t kb = (uint32_t)k * (uint32_t)sizeof(real_t);
    const real_t *base = (const real_t *)st.fx_base;
    real_t c0, xm1, xp1, xm2, xp2, ym1, yp1, ym2, yp2, zm1, zp1, zm2, zp2, xmym, xpym, xmyp, xpyp, xmzm, xpzm, xmzp, xpzp, ymzm, ypzm, ymzp, ypzp;
    // forward differences: f(x) of the seven rows, fetched with the window
    real_t bz0 = 0, by0 = 0, bx0 = 0, bc = 0, bx1 = 0, by1 = 0, bz1 = 0;
#define FD_LDX(p, off) (*(const real_t *)((const char *)((p) + (off)) + kb))
    if (wave_fast) {
        c0 = FD_LDX(x, 0);
        xm1 = FD_LDX(x, -1); xp1 = FD_LDX(x, 1); xm2 = FD_LDX(x, -2); xp2 = FD_LDX(x, 2);
        ym1 = FD_LDX(x, -nx); yp1 = FD_LDX(x, nx); ym2 = FD_LDX(x, -2 * nx); yp2 = FD_LDX(x, 2 * nx);
        zm1 = FD_LDX(x, -pl); zp1 = FD_LDX(x, pl); zm2 = FD_LDX(x, -2 * pl); zp2 = FD_LDX(x, 2 * pl);
        xmym = FD_LDX(x, -nx - 1); xpym = FD_LDX(x, -nx + 1); xmyp = FD_LDX(x, nx - 1); xpyp = FD_LDX(x, nx + 1);
        xmzm = FD_LDX(x, -pl - 1); xpzm = FD_LDX(x, -pl + 1); xmzp = FD_LDX(x, pl - 1); xpzp = FD_LDX(x, pl + 1);
        ymzm = FD_LDX(x, -pl - nx); ypzm = FD_LDX(x, -pl + nx); ymzp = FD_LDX(x, pl - nx); ypzp = FD_LDX(x, pl + nx);
        if (MODE == 0 && base) {       // (a row outside the grid does not exist: its value is never used)
            bc = FD_LDX(base, 0);
            bz0 = FD_LDX(base, -pl); by0 = FD_LDX(base, -nx); bx0 = FD_LDX(base, -1);
            bx1 = FD_LDX(base, 1); by1 = FD_LDX(base, nx); bz1 = FD_LDX(base, pl);
        }
        if (!__all(deep)) {            // the wavefront crosses the end of a grid row
            const real_t z = 0;
            const bool m1 = i >= 1, m2 = i >= 2, p1 = i < nx - 1, p2 = i < nx - 2;
            xm1 = m1 ? xm1 : z; xm2 = m2 ? xm2 : z; xp1 = p1 ? xp1 : z; xp2 = p2 ? xp2 : z;
            xmym = m1 ? xmym : z; xmyp = m1 ? xmyp : z; xmzm = m1 ? xmzm : z; xmzp = m1 ? xmzp : z;
            xpym = p1 ? xpym : z; xpyp = p1 ? xpyp : z; xpzm = p1 ? xpzm : z; xpzp = p1 ? xpzp : z;
        }
    } else {
        // W(dl, dj, di) = x at grid point (i + di, j + dj, l + dl) for |dl| + |dj| + |di| <= 2, 0 outside the grid
        auto ld = [&](const real_t *p, int dl, int dj, int di) -> real_t {
            const bool ok = deep || ((unsigned)(i + di) < (unsigned)nx && (unsigned)(j + dj) < (unsigned)ny && (unsigned)(l + dl) < (unsigned)nz);
            const int off = ok ? dl * pl + dj * nx + di : 0;
            const real_t v = p[k + off];
            return ok ? v : (real_t)0;
        };
        c0 = ld(x, 0, 0, 0);
        xm1 = ld(x, 0, 0, -1); xp1 = ld(x, 0, 0, 1); xm2 = ld(x, 0, 0, -2); xp2 = ld(x, 0, 0, 2);
        ym1 = ld(x, 0, -1, 0); yp1 = ld(x, 0, 1, 0); ym2 = ld(x, 0, -2, 0); yp2 = ld(x, 0, 2, 0);
        zm1 = ld(x, -1, 0, 0); zp1 = ld(x, 1, 0, 0); zm2 = ld(x, -2, 0, 0); zp2 = ld(x, 2, 0, 0);
        xmym = ld(x, 0, -1, -1); xpym = ld(x, 0, -1, 1); xmyp = ld(x, 0, 1, -1); xpyp = ld(x, 0, 1, 1);
        xmzm = ld(x, -1, 0, -1); xpzm = ld(x, -1, 0, 1); xmzp = ld(x, 1, 0, -1); xpzp = ld(x, 1, 0, 1);
        ymzm = ld(x, -1, -1, 0); ypzm = ld(x, -1, 1, 0); ymzp = ld(x, 1, -1, 0); ypzp = ld(x, 1, 1, 0);
        if (MODE == 0 && base) {       // (a row outside the grid does not exist: its value is never used)
            bc = ld(base, 0, 0, 0);
            bz0 = ld(base, -1, 0, 0); by0 = ld(base, 0, -1, 0); bx0 = ld(base, 0, 0, -1);
            bx1 = ld(base, 0, 0, 1); by1 = ld(base, 0, 1, 0); bz1 = ld(base, 1, 0, 0);
        }
    }
#undef FD_LDX
    const int a = in ? st.colptr[k - st.col_begin] : st.colptr[st.col_end - st.col_begin];
    const int b = in ? st.colptr[k - st.col_begin + 1] : a;
    const int c = (int)((const CT *)st.color)[k];
    const bool none = in && c == (int)(CT)(-1);
    const bool mine = in && !none && c >= c_lo && c < c_hi;
    fd_csc_wave_run<real_t> run;
    run.begin((real_t *)st.out, s_win[threadIdx.x >> 6], a, b, !in || mine || (none && c_lo == 0), kCap);
    __syncthreads();                // (s_eps)
    // Is this column's list of entries the stencil's?  Row t exists or not (ex[t]); the existing ones are the entries in this order
    // if it is: entry at[t] = number of existing rows before t.  Every lane fetches the row index at its own position for each t (no
    // dependent loads, no divergence) and compares.  The PLAN's pattern never changes, so the answer for the whole column range is
    // remembered in the plan's note (fd_csc_store.note): the first launch on a plan checks every column and counts the mismatches,
    // k_lap7_note records "verified" after it, and later launches read no row index at all (290 -> 210 us: the second dependent
    // memory round trip of every wavefront, profiles/r04_y_lap7_taken_apart.md).
    const bool regular = nx > 2 && ny > 2 && nz != 2;      // (otherwise stencil offsets coincide or wrap: everything through the functor)
    const bool ex[7] = {l > 0, j > 0, i > 0, true, i < nx - 1, j < ny - 1, l < nz - 1};
    const int off7[7] = {-pl, -nx, -1, 0, 1, nx, pl};
    const int cnt = b - a, lastq = cnt > 0 ? cnt - 1 : 0;
    int at[7], npos = 0;
#pragma unroll
    for (int t = 0; t < 7; ++t) { at[t] = npos < lastq ? npos : lastq; npos += ex[t] ? 1 : 0; }
    bool okp = regular && npos == cnt;
    const unsigned long long note0 = st.note ? st.note[0] : 0, key = lap7_note_key(f, st);
    const bool verified = note0 == key;                    // (key ^ 2: checked before, NOT the stencil -- nothing left to count)
    if (!verified) {
        int rv7[7];
#pragma unroll
        for (int t = 0; t < 7; ++t) rv7[t] = st.rowval[a + at[t]];          // (unconditional: at[t] is inside the column -- an empty one reads the pad)
#pragma unroll
        for (int t = 0; t < 7; ++t) okp = okp && (!ex[t] || rv7[t] == (int)k + off7[t]);
        if (st.note && note0 != (key ^ 2ull) && __any(in && !okp) && (threadIdx.x & 63) == 0) atomicAdd(&st.note[1], 1ull);
    }
    if (none && c_lo == 0)
        for (int q = a; q < b; ++q) run.put(q, (real_t)0);
    if (mine) {                     // (all lanes meet again at the flush below: the staged values leave in one wave-wide pass)
    const real_t h = c - c_lo < 32 ? s_eps[c - c_lo] : eps[c];
    const real_t z0 = 0;
    const real_t cp = c0 + h, cm = c0 - h;
    // The seven stencil rows in ascending order (= the order of a column's entries); `plus` / `minus`: the row at x + h e_k / x - h e_k.
    // The plus point holds x + 0.0 at its unperturbed coordinates, the minus point x - 0.0 == x: literally (the -0.0 case); a
    // coordinate outside the grid is the constant 0 of the residual (0 + 0.0 == 0).
#define Z +z0
#define FD_R0P lap7_row<real_t>(zm1 Z, zm2 Z, ymzm Z, xmzm Z, xpzm Z, ypzm Z, cp)
#define FD_R0M lap7_row<real_t>(zm1, zm2, ymzm, xmzm, xpzm, ypzm, cm)
#define FD_R1P lap7_row<real_t>(ym1 Z, ymzm Z, ym2 Z, xmym Z, xpym Z, cp, ymzp Z)
#define FD_R1M lap7_row<real_t>(ym1, ymzm, ym2, xmym, xpym, cm, ymzp)
#define FD_R2P lap7_row<real_t>(xm1 Z, xmzm Z, xmym Z, xm2 Z, cp, xmyp Z, xmzp Z)
#define FD_R2M lap7_row<real_t>(xm1, xmzm, xmym, xm2, cm, xmyp, xmzp)
#define FD_R3P lap7_row<real_t>(cp, zm1 Z, ym1 Z, xm1 Z, xp1 Z, yp1 Z, zp1 Z)
#define FD_R3M lap7_row<real_t>(cm, zm1, ym1, xm1, xp1, yp1, zp1)
#define FD_R4P lap7_row<real_t>(xp1 Z, xpzm Z, xpym Z, cp, xp2 Z, xpyp Z, xpzp Z)
#define FD_R4M lap7_row<real_t>(xp1, xpzm, xpym, cm, xp2, xpyp, xpzp)
#define FD_R5P lap7_row<real_t>(yp1 Z, ypzm Z, cp, xmyp Z, xpyp Z, yp2 Z, ypzp Z)
#define FD_R5M lap7_row<real_t>(yp1, ypzm, cm, xmyp, xpyp, yp2, ypzp)
#define FD_R6P lap7_row<real_t>(zp1 Z, cp, ymzp Z, xmzp Z, xpzp Z, ypzp Z, zp2 Z)
#define FD_R6M lap7_row<real_t>(zp1, cm, ymzp, xmzp, xpzp, ypzp, zp2)
    // ONE straight-line path for interior and boundary columns alike (a wavefront of 64 consecutive grid points meets a face of the
    // grid two times out of three: a separate boundary path would be the common case).  Stencil row t exists or not (ex[t]); the
    // existing ones are the column's entries in this order if the pattern is the stencil's: entry at[t] = number of existing rows
    // before t.  Every lane fetches the row index at its own position for each t (no dependent loads, no divergence), compares, and
    // stages the quotients of the rows that exist.  A column whose entries are anything else goes through the functor, entry by entry.
    // all quotients by ONE divisor: its reciprocal once, then the correctly rounded quotients of div_shared (the bits of IEEE a / b).
    // Evaluated before the row indices are back (the comparison only decides where the values go).
    const real_t div = MODE == 1 ? 2 * h : h;
    const real_t yd = sizeof(real_t) == 8 ? (real_t)1 / div : (real_t)0;
    // (forward differences without fx_base: f(x) of the row from the window -- the plain launcher's operands and operations)
    const bool own = MODE == 0 && !base;
    if (own) {
        const real_t cb = c0;
#define cm cb
        bz0 = FD_R0M; by0 = FD_R1M; bx0 = FD_R2M; bc = FD_R3M; bx1 = FD_R4M; by1 = FD_R5M; bz1 = FD_R6M;
#undef cm
    }
#define FD_FAST(plus, minus, bv) div_shared<true>(sub_exact((plus), MODE == 1 ? (minus) : (bv)), div, yd)
    const real_t v7[7] = {FD_FAST(FD_R0P, FD_R0M, bz0), FD_FAST(FD_R1P, FD_R1M, by0), FD_FAST(FD_R2P, FD_R2M, bx0), FD_FAST(FD_R3P, FD_R3M, bc),
                          FD_FAST(FD_R4P, FD_R4M, bx1), FD_FAST(FD_R5P, FD_R5M, by1), FD_FAST(FD_R6P, FD_R6M, bz1)};
#undef FD_FAST
    if (okp) {
#pragma unroll
        for (int t = 0; t < 7; ++t)
            if (ex[t]) run.put(a + at[t], v7[t]);
    } else {
        // rows that are no stencil neighbours, degenerate grids: every entry through the functor
        for (int q = a; q < b; ++q) {
            const int r = st.rowval[q];
            fd_column_point<real_t> X = {x, k, h, 0};
            const real_t vp = f(r, X);
            real_t vm, div = h;
            if (MODE == 1) { X.minus = 1; vm = f(r, X); div = 2 * h; }
            else if (base) vm = base[r];
            else { X.minus = 2; vm = f(r, X); }
            run.put(q, sub_exact(vp, vm) / div);
        }
    }
#undef Z
    }
    run.template flush<true>();
}

// ---- FD_F_SPARSE ----------------------------------------------------------------------------------------------------------------
#ifndef FD_SPARSE_U
#define FD_SPARSE_U 4
#endif
struct SparseF {
    const int32_t *srow, *scol;      // the pattern by rows, ascending columns (device)
    template <typename T, class P> __device__ __forceinline__ T row(int64_t r, const P &X) const
    {
        // FD_SPARSE_U entries at a time: their column indices first (one round trip, from positions clamped into the row), then their
        // coordinates (a second one), then the terms in the row's order -- entry by entry every term costs two dependent round trips
        // (the generic column kernel spent 81 of its 100 loads waiting for exactly one load each)
        const int a0 = srow[r], a1 = srow[r + 1];
        T s = zero_of<T>();
        for (int a = a0; a < a1; a += FD_SPARSE_U) {
            int64_t j[FD_SPARSE_U];
            T v[FD_SPARSE_U];
#pragma unroll
            for (int u = 0; u < FD_SPARSE_U; ++u) j[u] = scol[a + u < a1 ? a + u : a1 - 1];
#pragma unroll
            for (int u = 0; u < FD_SPARSE_U; ++u) v[u] = X(j[u]);
#pragma unroll
            for (int u = 0; u < FD_SPARSE_U; ++u) {
                const T t = ((real_t)1 + kEighth * (real_t)(int)((r + 3 * j[u]) & 7)) * (v[u] + (kQuarter * v[u]) * v[u]);
                if (a + u < a1) s = a + u == a0 ? t : s + t;
            }
        }
        return s;
    }
    template <class P> __device__ __forceinline__ real_t operator()(long long r, const P &X) const { return row<real_t>(r, X); }
    // the residual is SEPARABLE on its pattern (include/fdjac_device.h): entry (r, j)'s term -- the one `row` adds, the same bits
    static constexpr bool fd_separable = true;
    template <typename T> __device__ __forceinline__ T term(long long r, long long j, T v) const
    {
        return ((real_t)1 + kEighth * (real_t)(int)((r + 3 * j) & 7)) * (v + (kQuarter * v) * v);
    }

    // fd_csc_store_cols_win (include/fdjac_device.h): the row pattern of the rows [r_lo, r_hi) a workgroup's columns can touch, kept in
    // LDS -- the rows' offsets (int32, r_hi - r_lo + 1 of them) and up to `cap` of their column indices as 16-bit distances from the
    // start of the workgroup's x window (0xFFFF: the column lies outside the window -> index and coordinate are read from memory).  A row whose entries are ALL staged with valid codes is flagged "fast":
    // its evaluation is a straight line of LDS reads and 32-bit arithmetic.  Anything else is served from memory: same indices, same
    // order, same bits.
    struct Staged {
        const int32_t *srow, *scol;
        const FD_LDS_PTR(int32_t) l_row;        // LDS: srow[r_lo .. r_hi]
        const FD_LDS_PTR(uint16_t) l_col;       // LDS: codes of the entries [e_lo, e_lo + e_n)
        const FD_LDS_PTR(uint8_t) l_fast;       // LDS: 1 = every entry of the row has a valid code
        int64_t r_lo, r_hi, w0;
        int e_lo, e_n;
        __device__ __forceinline__ int64_t col_of(int a) const
        {
            const int k = a - e_lo;
            if (k >= 0 && k < e_n) {
                const unsigned cde = l_col[k];
                if (cde != 0xFFFFu) return w0 + (int64_t)cde;
            }
            return scol[a];
        }
        template <typename T, class P> __device__ __forceinline__ T row(int64_t r, const P &X) const
        {
            const bool st = r >= r_lo && r < r_hi;
            const int ri = st ? (int)(r - r_lo) : 0;
            const int a0 = st ? l_row[ri] : srow[r], a1 = st ? l_row[ri + 1] : srow[r + 1];
            T s = zero_of<T>();
            if (st && l_fast[ri]) {
                // (the terms and their order are those of SparseF::row -- same bits; (r + 3 j) mod 8 from the low bits alone)
                const unsigned rw = (unsigned)((r + 3 * w0) & 7);
                for (int a = a0; a < a1; a += 2) {
                    const bool two = a + 1 < a1;
                    const unsigned c0 = l_col[a - e_lo], c1 = l_col[(two ? a + 1 : a) - e_lo];
                    const T v0 = X.at(c0), v1 = X.at(c1);
                    const T t0 = ((real_t)1 + kEighth * (real_t)(int)((rw + 3 * c0) & 7)) * (v0 + (kQuarter * v0) * v0);
                    const T t1 = ((real_t)1 + kEighth * (real_t)(int)((rw + 3 * c1) & 7)) * (v1 + (kQuarter * v1) * v1);
                    s = a == a0 ? t0 : s + t0;
                    if (two) s = s + t1;
                }
                return s;
            }
            for (int a = a0; a < a1; a += FD_SPARSE_U) {
                int64_t j[FD_SPARSE_U];
                T v[FD_SPARSE_U];
#pragma unroll
                for (int u = 0; u < FD_SPARSE_U; ++u) j[u] = col_of(a + u < a1 ? a + u : a1 - 1);
#pragma unroll
                for (int u = 0; u < FD_SPARSE_U; ++u) v[u] = X(j[u]);
#pragma unroll
                for (int u = 0; u < FD_SPARSE_U; ++u) {
                    const T t = ((real_t)1 + kEighth * (real_t)(int)((r + 3 * j[u]) & 7)) * (v[u] + (kQuarter * v[u]) * v[u]);
                    if (a + u < a1) s = a + u == a0 ? t : s + t;
                }
            }
            return s;
        }
        template <class P> __device__ __forceinline__ real_t operator()(long long r, const P &X) const { return row<real_t>(r, X); }
    };
    static size_t stage_bytes(int64_t rows, int cap) { return ((size_t)(rows + 2) * 4 + 15) / 16 * 16 + ((size_t)cap * 2 + 15) / 16 * 16 + (size_t)rows + 32; }
    __device__ __forceinline__ Staged stage(FD_LDS_PTR(unsigned char) lds, long long r_lo, long long r_hi, long long w0, long long w1, int cap) const
    {
        FD_LDS_PTR(int32_t) l_row = (FD_LDS_PTR(int32_t))lds;
        const int nr = (int)(r_hi - r_lo);
        const unsigned col_off = ((unsigned)(nr + 2) * 4u + 15u) / 16u * 16u;
        FD_LDS_PTR(uint16_t) l_col = (FD_LDS_PTR(uint16_t))(lds + col_off);
        FD_LDS_PTR(uint8_t) l_fast = (FD_LDS_PTR(uint8_t))(lds + col_off + ((unsigned)cap * 2u + 15u) / 16u * 16u);
        // (batches of loads issued together: one memory round trip per batch, not per element)
        for (int i0 = 0; i0 <= nr; i0 += 4 * kBlock) {
            int v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * kBlock + (int)threadIdx.x; v[u] = i <= nr ? srow[r_lo + i] : 0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * kBlock + (int)threadIdx.x; if (i <= nr) l_row[i] = v[u]; }
        }
        const int e_lo = srow[r_lo], e_hi = srow[r_hi];
        const int e_n = e_hi - e_lo < cap ? e_hi - e_lo : cap;
        const int64_t wlen = w1 - w0 < 0xFFFF ? w1 - w0 : 0xFFFF;
        // the column indices as aligned 16-byte quads (scol comes from hipMalloc: index alignment = address alignment); the quad that
        // would read past the staged range's end falls back to single loads
        const int q_lo = e_lo >> 2, q_hi = (e_lo + e_n + 3) >> 2;           // quads [q_lo, q_hi) cover the entries [e_lo, e_lo + e_n)
        for (int q0 = q_lo; q0 < q_hi; q0 += 4 * kBlock) {
            int4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u * kBlock + (int)threadIdx.x;
                v[u] = int4{0, 0, 0, 0};
                if (q < q_hi) {
                    if (4 * q + 3 < e_lo + e_n) v[u] = *reinterpret_cast<const int4 *>(scol + 4 * (int64_t)q);
                    else {
                        if (4 * q + 0 < e_lo + e_n) v[u].x = scol[4 * (int64_t)q + 0];
                        if (4 * q + 1 < e_lo + e_n) v[u].y = scol[4 * (int64_t)q + 1];
                        if (4 * q + 2 < e_lo + e_n) v[u].z = scol[4 * (int64_t)q + 2];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u * kBlock + (int)threadIdx.x;
                if (q >= q_hi) continue;
                const int vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int k = 4 * q + t - e_lo;
                    if (k < 0 || k >= e_n) continue;
                    const int64_t d = (int64_t)vv[t] - w0;
                    l_col[k] = (d >= 0 && d < wlen) ? (uint16_t)d : (uint16_t)0xFFFFu;      // (only coordinates INSIDE the staged window get a code)
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nr; i += kBlock) {
            const int a0 = l_row[i], a1 = l_row[i + 1];
            bool ok = a1 <= e_lo + e_n;
            for (int a = a0; ok && a < a1; ++a) ok = l_col[a - e_lo] != 0xFFFFu;
            l_fast[i] = ok ? 1 : 0;
        }
        return Staged{srow, scol, l_row, l_col, l_fast, r_lo, r_hi, w0, e_lo, e_n};
    }
};

template <typename T, class F>
__global__ void __launch_bounds__(kBlock) k_f_rows(T *__restrict__ fx, const T *__restrict__ x, F f, int64_t xs, int64_t fs, int64_t r0, int64_t r1)
{
    const int64_t r = r0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= r1) return;
    const PlainPoint<T> P = {x + (int64_t)blockIdx.y * xs};
    fx[(int64_t)blockIdx.y * fs + r] = f.template row<T>(r, P);
}

// ---- the sparse family's column store, ENTRY-BALANCED (round 5) ---------------------------------------------------------------------
// fd_csc_store_cols_win gives every lane a column: the lane evaluates its column's ~6 rows one after the other, and a wavefront runs
// as long as its longest (row lengths of a scattered pattern are Poisson-like: the max over 64 lanes is about twice the mean -- 2447
// VALU instructions per wavefront, profiles/r05_band_store.md).  Here the workgroup's stored entries -- one contiguous run of nzval --
// are the work items: they are counting-sorted by the length of their row in LDS (which order the entries are evaluated in changes
// nothing: every entry is computed on its own), and lane i of round k takes sorted entry 256 k + i, so the lanes of a wavefront
// evaluate rows of (nearly) equal length.  Everything an entry needs comes from LDS: the window of x and the rows' pattern as in
// fd_csc_store_cols_win (SparseF::stage), the entry's column (one byte), its column's step and reciprocal.  Same operations per entry
// as the column kernels: same bits.
constexpr int kSpChunk = 2304;       // entries sorted at a time (9 per column on average; longer runs go in several chunks)
static size_t sparse_sorted_lds_bytes(int64_t reach, int cap)
{
    return sizeof(real_t) * (size_t)(fd_csc_win_xlen(reach) + 2 * kBlock) + (size_t)kSpChunk * 3 + 72 * 4 + 64 + SparseF::stage_bytes(fd_csc_win_rlen(reach), cap) + 32;
}
template <typename CT, int MODE>
__global__ void __launch_bounds__(kBlock) k_f_sparse_store_sorted(SparseF f, const real_t *__restrict__ x, const real_t *__restrict__ eps, int c_lo, int c_hi,
                                                                  fd_csc_store st, int reach, int stage_cap, unsigned long long rows_key)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];
    if (rows_key && st.note && st.note[0] == rows_key) return;        // the row-wise launch that follows stores this plan's Jacobian (k_f_sparse_store_rows)
    const long long nblk = (st.col_end - st.col_begin + kBlock - 1) / kBlock, blk = fd_xcd_block(blockIdx.x, nblk);
    if (blk >= nblk) return;
    const long long j0 = st.col_begin + blk * kBlock, jn = j0 + kBlock < st.col_end ? j0 + kBlock : st.col_end;
    const long long r_lo = j0 - reach > 0 ? j0 - reach : 0, r_hi = jn + reach < st.M ? jn + reach : st.M;
    long long w0 = r_lo - reach > 0 ? r_lo - reach : 0, w1 = r_hi + reach < st.N ? r_hi + reach : st.N;
    w0 &= ~1ll;
    // LDS: [x window][step of every column][its reciprocal][sorted entry ids u16][entry -> column u8][histogram: 32 counts, 32 bucket
    // cursors, the total][the functor's rows]
    FD_LDS_PTR(real_t) s_x = (FD_LDS_PTR(real_t))sp_lds;
    FD_LDS_PTR(real_t) s_h = s_x + fd_csc_win_xlen(reach);
    FD_LDS_PTR(real_t) s_y = s_h + kBlock;
    FD_LDS_PTR(uint16_t) s_sorted = (FD_LDS_PTR(uint16_t))(s_y + kBlock);
    FD_LDS_PTR(uint8_t) s_ecol = (FD_LDS_PTR(uint8_t))(s_sorted + kSpChunk);
    FD_LDS_PTR(int) s_hist = (FD_LDS_PTR(int))(((FD_LDS_PTR(unsigned char))(s_ecol + kSpChunk)) + ((4 - ((kSpChunk * 3) & 3)) & 3));
    const unsigned f_off = ((unsigned)(sizeof(real_t) * (size_t)(fd_csc_win_xlen(reach) + 2 * kBlock)) + (unsigned)kSpChunk * 3u + 4u + 72u * 4u + 15u) & ~15u;
    FD_LDS_PTR(unsigned char) s_f = (FD_LDS_PTR(unsigned char))sp_lds + f_off;
    {   // the window of x: 16-byte pairs, every load of a batch in flight together
        const long long nx = w1 - w0, npair = nx / 2;
        for (long long i0 = 0; i0 < npair; i0 += 4 * kBlock) {
            r2_t vx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long long i = i0 + u * kBlock + threadIdx.x; if (i < npair) vx[u] = *reinterpret_cast<const r2_t *>(x + w0 + 2 * i); }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long long i = i0 + u * kBlock + threadIdx.x; if (i < npair) { s_x[2 * i] = vx[u].x; s_x[2 * i + 1] = vx[u].y; } }
        }
        if ((nx & 1) && threadIdx.x == 0) s_x[nx - 1] = x[w0 + nx - 1];
    }
    // this thread's column: its entries, colour, step
    const long long j = j0 + threadIdx.x;
    const bool in = j < st.col_end;
    const int a = in ? st.colptr[j - st.col_begin] : st.colptr[st.col_end - st.col_begin];
    const int b = in ? st.colptr[j - st.col_begin + 1] : a;
    const CT *color = (const CT *)st.color;
    const int c = in ? (int)color[j] : 0;
    const bool none = in && c == (int)(CT)(-1);
    const bool mine = in && !none && c >= c_lo && c < c_hi;
    const real_t h = mine ? eps[c] : (real_t)0;
    const real_t dv = MODE == 1 ? 2 * h : h;
    s_h[threadIdx.x] = h;                                              // (0: the column's entries are not this launch's)
    s_y[threadIdx.x] = mine ? (real_t)1 / dv : (real_t)0;
    real_t *out = (real_t *)st.out;
    if (none && c_lo == 0)
        for (int q = a; q < b; ++q) out[q] = (real_t)0;
    const auto fs = f.stage(s_f, r_lo, r_hi, w0, w1, stage_cap);     // (ends with the rows' flags; the barrier below publishes everything)
    const int qa = st.colptr[j0 - st.col_begin], qb = st.colptr[jn - st.col_begin];
    const real_t *base = (const real_t *)st.fx_base;
    for (int q0 = qa; q0 < qb; q0 += kSpChunk) {
        const int Ec = qb - q0 < kSpChunk ? qb - q0 : kSpChunk;
        __syncthreads();                                               // (LDS of the chunk before is no longer read; the staging is complete)
        if (threadIdx.x < 64) s_hist[threadIdx.x] = 0;
        // entry -> column: every column marks its own entries of this chunk
        for (int q = (a > q0 ? a : q0); q < b && q < q0 + Ec; ++q) s_ecol[q - q0] = (uint8_t)threadIdx.x;
        __syncthreads();
        // row lengths (clamped to 31) -> histogram; entries of columns that are not this launch's get no slot
        int len[(kSpChunk + kBlock - 1) / kBlock], rr[(kSpChunk + kBlock - 1) / kBlock];
#pragma unroll
        for (int k = 0; k < (kSpChunk + kBlock - 1) / kBlock; ++k) {
            const int e = k * kBlock + (int)threadIdx.x;
            rr[k] = e < Ec ? st.rowval[q0 + e] : 0;
        }
#pragma unroll
        for (int k = 0; k < (kSpChunk + kBlock - 1) / kBlock; ++k) {
            const int e = k * kBlock + (int)threadIdx.x;
            len[k] = -1;
            if (e < Ec && s_h[s_ecol[e]] != (real_t)0) {
                const int r = rr[k];
                const bool stg = r >= r_lo && r < r_hi;
                const int n = stg ? fs.l_row[r - r_lo + 1] - fs.l_row[r - r_lo] : f.srow[r + 1] - f.srow[r];
                len[k] = n > 31 ? 31 : n;
                atomicAdd((int *)(s_hist + len[k]), 1);
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {      // exclusive prefix, LONGEST rows first (one wavefront)
            const int idx = 31 - (int)threadIdx.x;
            int v = (threadIdx.x < 32) ? s_hist[idx] : 0, incl = v;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { const int t = __shfl_up(incl, o, 64); if ((int)threadIdx.x >= o) incl += t; }
            if (threadIdx.x < 32) s_hist[32 + idx] = incl - v;          // start of the bucket
            if (threadIdx.x == 31) s_hist[64] = incl;                   // the number of sorted entries
        }
        __syncthreads();
        const int nsorted = s_hist[64];
#pragma unroll
        for (int k = 0; k < (kSpChunk + kBlock - 1) / kBlock; ++k) {
            const int e = k * kBlock + (int)threadIdx.x;
            if (len[k] >= 0) s_sorted[atomicAdd((int *)(s_hist + 32 + len[k]), 1)] = (uint16_t)e;
        }
        __syncthreads();
        // the entries, in sorted order: lane i of round k takes sorted entry 256 k + i.  Row index and f(x) of every round are requested
        // up front (two memory round trips in all; taken entry by entry they were two DEPENDENT round trips per entry, 12 per thread --
        // the first form of this kernel spent most of its 21 us per workgroup there)
        constexpr int K = (kSpChunk + kBlock - 1) / kBlock;
        int ee[K];
        long long rw[K];
        real_t bv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { const int i = k * kBlock + (int)threadIdx.x; ee[k] = i < nsorted ? (int)s_sorted[i] : -1; }
#pragma unroll
        for (int k = 0; k < K; ++k) rw[k] = st.rowval[q0 + (ee[k] >= 0 ? ee[k] : 0)];
#pragma unroll
        for (int k = 0; k < K; ++k) bv[k] = (MODE == 0) ? base[rw[k]] : (real_t)0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (ee[k] < 0) continue;
            const int e = ee[k], tc = s_ecol[e];
            const long long r = rw[k];
            const real_t he = s_h[tc], ye = s_y[tc];
            const long long jc = j0 + tc;
            fd_window_column_point<real_t> X = {x, s_x, w0, w1, jc, he, 0, (jc >= w0 && jc < w1) ? (unsigned)(jc - w0) : 0xFFFFFFFFu};
            const real_t vp = fs.template row<real_t>(r, X);
            real_t vm = bv[k];
            if (MODE == 1) { X.minus = 1; vm = fs.template row<real_t>(r, X); }
            out[q0 + e] = fd_div_shared<real_t>(vp - vm, MODE == 1 ? 2 * he : he, ye);
        }
    }
}

// ---- the sparse family's store, ROW BY ROW (round 5) ---------------------------------------------------------------------------------
// The column kernels give every stored entry (r, j) its own evaluation of row r: nnz x (row length) terms -- 36 per column of the
// random band, each with its index decode, window lookup and perturbation select: instruction issue, 196 us.  But the L entries of
// a row share everything except ONE term: a thread that owns row r forms the row's L plain terms once (keeps them in LDS, their
// left-to-right sum is f(x)_r), then for entry k the sum  (t_0 + .. + t_{k-1}) + t'_k + t_{k+1} + .. + t_{L-1}  -- the prefix
// carried from entry to entry, the perturbed term evaluated once, the suffix added from LDS: the additions of the full evaluation
// in the same order, so the same bits, with 2 L term evaluations and L (L - 1) / 2 additions per row instead of L^2 evaluations.
// The quotient goes to the entry's slot in J's CSC storage: `sdest`, the functor's own map from its row-major list to the CSC
// order it was BUILT from.  That must be the plan's pattern: the first launch on a plan only checks (every entry's slot holds its row
// and lies in its column; the entry counts agree) while the column kernel stores; the last workgroup to report records the verdict in
// the plan's note, and from then on this kernel stores and the column kernel returns at once (or the other way round: a different
// pattern).  The launcher reads the verdict back asynchronously (never a synchronisation) and then launches only the one that works.
// x and the colours of the rows' window [r0 - reach, r1 + reach) and the tile's run of the row-major lists (columns, slots: one
// contiguous range) are staged in LDS with lane-consecutive loads; what does not fit or lies outside is read from memory.
constexpr int kRowRegs = 14;         // entries of a row k_f_sparse_store_rows keeps in registers
__device__ __forceinline__ real_t sparse_weight(unsigned k)          // 1 + k / 8, k < 8: the three leading mantissa bits
{
    if constexpr (sizeof(real_t) == 8) return __hiloint2double((int)(0x3FF00000u | (k << 17)), 0);
    else return __uint_as_float(0x3F800000u | (k << 20));
}
__device__ __forceinline__ real_t sparse_term(unsigned k, real_t v) { return sparse_weight(k) * (v + (kQuarter * v) * v); }
// LDS: [x window][step, reciprocal per colour of the batch][column, slot of the tile's entries][colours of the window]
static size_t sparse_rows_lds_bytes(int64_t reach, int ncol, int cap)
{
    const size_t xlen = (size_t)(kBlock + 2 * reach + 2);
    return sizeof(real_t) * (xlen + 2 * (size_t)ncol) + 4 * (2 * (size_t)cap + xlen) + 64;
}
template <typename CT, int MODE>
__global__ void __launch_bounds__(kBlock) k_f_sparse_store_rows(SparseF f, const int32_t *__restrict__ sdest, long long e_base, const real_t *__restrict__ x,
                                                                const real_t *__restrict__ eps, int c_lo, int c_hi, fd_csc_store st, int reach, int cap,
                                                                long long row_lo, long long row_hi, unsigned long long key, long long expect, int known)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sr_lds[];
    const long long ntile = (row_hi - row_lo + kBlock - 1) / kBlock, tile = fd_xcd_block(blockIdx.x, ntile);
    if (tile >= ntile) return;
    const unsigned long long note0 = known ? key : st.note[0];          // (known: the launcher has read the verdict -- one memory round trip less)
    if (note0 == (key ^ 2ull)) return;                                  // another pattern: the column kernel has stored
    const long long R0 = row_lo + tile * kBlock, R1 = R0 + kBlock < row_hi ? R0 + kBlock : row_hi;
    const long long r = R0 + threadIdx.x;
    const bool in = r < R1;
    const int a0 = in ? f.srow[r] : 0, a1 = in ? f.srow[r + 1] : 0, L = a1 - a0;
    const long long cb = st.col_begin, ce = st.col_end;
    if (note0 != key) {                                                 // first launch on this plan: check, store nothing
        bool bad = false;
        for (int a = a0; a < a1; ++a) {
            const long long j = f.scol[a];
            if (j < cb || j >= ce) continue;
            const long long q = (long long)sdest[a] - e_base;
            const int qa = st.colptr[j - cb], qb = st.colptr[j - cb + 1];
            bad = bad || !(q >= qa && q < qb) || st.rowval[q >= qa && q < qb ? q : qa] != (int)r;
        }
        // note[2]: workgroups that have reported (low word) and that met a mismatch (high word); the last one records the verdict
        const int any_bad = __syncthreads_or(bad ? 1 : 0);
        if (threadIdx.x == 0) {
            const unsigned long long add = 1ull + (any_bad ? (1ull << 32) : 0ull);
            const unsigned long long old = atomicAdd(&st.note[2], add);
            if ((old & 0xFFFFFFFFull) + 1 == (unsigned long long)ntile) {
                const bool same = ((old + add) >> 32) == 0 && (long long)st.colptr[ce - cb] == expect;
                st.note[2] = 0;
                st.note[0] = same ? key : (key ^ 2ull);
            }
        }
        return;
    }
    long long w0 = R0 - reach > 0 ? R0 - reach : 0, w1 = R1 + reach < st.N ? R1 + reach : st.N;
    w0 &= ~1ll;
    const int xlen = kBlock + 2 * reach + 2, nchunk = c_hi - c_lo;
    FD_LDS_PTR(real_t) s_x = (FD_LDS_PTR(real_t))sr_lds;
    FD_LDS_PTR(real_t) s_h = s_x + xlen;                                // step of colour c_lo + i
    FD_LDS_PTR(real_t) s_y = s_h + nchunk;                              // 1 / (step or 2 step)
    FD_LDS_PTR(int) s_j = (FD_LDS_PTR(int))(s_y + nchunk);              // column of the tile's i-th entry
    FD_LDS_PTR(int) s_q = s_j + cap;                                    // its slot in the plan's nzval
    FD_LDS_PTR(int) s_c = s_q + cap;                                    // colour of column w0 + i (-1: none)
    const CT *color = (const CT *)st.color;
    // the tile's entries are one contiguous run of the functor's row-major lists: staged with lane-consecutive loads -- after the
    // window of x and of the colours, whose loads do not depend on the row offsets and travel with them
    const int A0 = f.srow[R0], A1 = f.srow[R1];
    {
        const long long nx = w1 - w0, npair = nx / 2;
        for (long long i0 = 0; i0 < npair; i0 += 4 * kBlock) {
            r2_t vx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long long i = i0 + u * kBlock + threadIdx.x; if (i < npair) vx[u] = *reinterpret_cast<const r2_t *>(x + w0 + 2 * i); }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long long i = i0 + u * kBlock + threadIdx.x; if (i < npair) { s_x[2 * i] = vx[u].x; s_x[2 * i + 1] = vx[u].y; } }
        }
        if ((nx & 1) && threadIdx.x == 0) s_x[nx - 1] = x[w0 + nx - 1];
        for (long long i = threadIdx.x; i < nx; i += kBlock) { const CT c = color[w0 + i]; s_c[i] = c == (CT)(-1) ? -1 : (int)c; }
        for (int i = threadIdx.x; i < nchunk; i += kBlock) { const real_t h = eps[c_lo + i]; s_h[i] = h; s_y[i] = (real_t)1 / (MODE == 1 ? 2 * h : h); }
    }
    const int nst = A1 - A0 < cap ? A1 - A0 : cap;
    for (int i0 = 0; i0 < nst; i0 += 4 * kBlock) {
        int vj[4], vq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * kBlock + (int)threadIdx.x; const int ic = i < nst ? i : nst - 1; vj[u] = f.scol[A0 + ic]; vq[u] = sdest[A0 + ic]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * kBlock + (int)threadIdx.x; if (i < nst) { s_j[i] = vj[u]; s_q[i] = vq[u]; } }
    }
    __syncthreads();
    if (!in || L == 0) return;
    const unsigned wlen = (unsigned)(w1 - w0);
    const unsigned rw = (unsigned)r & 7u;
    const int b0 = a0 - A0;                                             // the row's entries are the tile's [b0, b0 + L)
    real_t *out = (real_t *)st.out;
    // forward differences: the subtrahend is what the plan hands over (the caller's f_in, or f(x) from a plain evaluation) -- or,
    // without one (FD_LAZY_CAP_STORE_CSC_BASE), the row's own plain sum
    const bool given = MODE == 0 && st.fx_base != nullptr;
    const real_t fx_given = given ? ((const real_t *)st.fx_base)[r] : (real_t)0;
    // PURE: the whole run is staged and every column is local (so inside the window): the loops below then contain no load from
    // memory at all.  That matters more than it looks: this target counts loads and stores in ONE counter, and a loop body that MAY
    // load makes the compiler wait for everything outstanding -- the store of the iteration before -- on every trip: 12 trips x a
    // store's round trip = the 19 us a wavefront of the first form of this kernel lived (56 % of its cycles in s_waitcnt, 240 us).
    auto work = [&](auto pure_tag) {
        constexpr bool PURE = decltype(pure_tag)::value;
        auto coord = [&](long long j) -> real_t {
            const unsigned off = (unsigned)(j - w0);
            if constexpr (PURE) return s_x[off];
            else return off < wlen ? (real_t)s_x[off] : x[j];
        };
        auto col_of = [&](int i) -> int { if constexpr (PURE) return s_j[i]; else return i < nst ? (int)s_j[i] : f.scol[A0 + i]; };
        auto plain = [&](int i) -> real_t {                             // the tile's i-th entry's term at x
            const long long j = col_of(i);
            return sparse_term((rw + 3u * (unsigned)j) & 7u, coord(j));
        };
        if constexpr (PURE) {
            // rows of at most kRowRegs entries (all but one in a thousand of a Poisson(6) pattern): columns, slots, coordinates, colours
            // and plain terms in REGISTERS, every loop unrolled and predicated -- the loops over LDS below pay one LDS round trip per
            // term added (the suffix sums alone: 66 dependent trips for a wavefront whose longest row has 12 entries)
            constexpr int RL = kRowRegs;
            if (L <= RL) {
                int jj[RL], qq[RL];
                real_t tt[RL];
#pragma unroll
                for (int u = 0; u < RL; ++u) { const int i = b0 + (u < L ? u : L - 1); jj[u] = s_j[i]; qq[u] = s_q[i]; }
                real_t fx = 0;
#pragma unroll
                for (int u = 0; u < RL; ++u) {
                    tt[u] = sparse_term((rw + 3u * (unsigned)jj[u]) & 7u, s_x[(unsigned)(jj[u] - (int)w0)]);
                    fx = u == 0 ? tt[0] : (u < L ? fx + tt[u] : fx);
                }
                real_t pre = 0;
#pragma unroll
                for (int k = 0; k < RL; ++k) {
                    if (k < L) {
                        const unsigned off = (unsigned)(jj[k] - (int)w0);
                        const int c = s_c[off];
                        const real_t v = s_x[off];
                        const long long q = (long long)qq[k] - e_base;
                        if (c < 0) {
                            if (c_lo == 0) out[q] = (real_t)0;
                        } else if (c >= c_lo && c < c_hi) {
                            const real_t h = s_h[c - c_lo], y = s_y[c - c_lo];
                            const unsigned wk = (rw + 3u * (unsigned)jj[k]) & 7u;
                            real_t sp = sparse_term(wk, v + h), sm = MODE == 1 ? sparse_term(wk, v - h) : (real_t)0;
                            if (k > 0) { sp = pre + sp; if (MODE == 1) sm = pre + sm; }
#pragma unroll
                            for (int u = k + 1; u < RL; ++u)
                                if (u < L) { sp = sp + tt[u]; if (MODE == 1) sm = sm + tt[u]; }
                            out[q] = fd_div_shared<real_t>(sp - (MODE == 1 ? sm : given ? fx_given : fx), MODE == 1 ? 2 * h : h, y);
                        }
                        pre = k == 0 ? tt[0] : pre + tt[k];
                    }
                }
                return;
            }
        }
        real_t fx = 0;
        for (int u = 0; u < L; ++u) {
            const int j = col_of(b0 + u);
            const real_t t = sparse_term((rw + 3u * (unsigned)j) & 7u, coord(j));
            fx = u == 0 ? t : fx + t;
        }
        real_t pre = 0;
        for (int k = 0; k < L; ++k) {
            const long long j = col_of(b0 + k);
            long long q;
            if constexpr (PURE) q = (long long)s_q[b0 + k] - e_base;
            else q = (long long)(b0 + k < nst ? (int)s_q[b0 + k] : sdest[a0 + k]) - e_base;
            const real_t tk = plain(b0 + k);
            if (PURE || (j >= cb && j < ce)) {
                const unsigned off = (unsigned)(j - w0);
                int c;
                if constexpr (PURE) c = s_c[off];
                else c = off < wlen ? (int)s_c[off] : (color[j] == (CT)(-1) ? -1 : (int)color[j]);
                if (c < 0) {
                    if (c_lo == 0) out[q] = (real_t)0;
                } else if (c >= c_lo && c < c_hi) {
                    const real_t h = s_h[c - c_lo], y = s_y[c - c_lo], v = coord(j);
                    const unsigned wk = (rw + 3u * (unsigned)j) & 7u;
                    real_t sp = sparse_term(wk, v + h), sm = MODE == 1 ? sparse_term(wk, v - h) : (real_t)0;
                    if (k > 0) { sp = pre + sp; if (MODE == 1) sm = pre + sm; }
                    for (int u = k + 1; u < L; ++u) { const real_t t = plain(b0 + u); sp = sp + t; if (MODE == 1) sm = sm + t; }
                    out[q] = fd_div_shared<real_t>(sp - (MODE == 1 ? sm : given ? fx_given : fx), MODE == 1 ? 2 * h : h, y);
                }
            }
            pre = k == 0 ? tk : pre + tk;
        }
    };
    if (A1 - A0 <= cap && cb == 0 && ce == st.N) work(std::true_type{});
    else work(std::false_type{});
}

// ---- the complex step through the column store (FD_LAZY_CAP_STORE_CSC_COMPLEX) ----------------------------------------------------------
// Every stored entry (r, j): row r at the complex point x + i eps_c m_c, imag / eps_c stored (src/jacobians.jl:633-635 +
// ext/FiniteDiffSparseArraysExt.jl:38-47).  With a verified colouring the point differs from x in coordinate j only as far as row r can
// see; otherwise the whole colour's point is formed.  The operations of the functor's row<cd> on materialised points: same bits as the
// hand-over path (whose decompression divides imag by eps the same way).
template <typename CT> struct CplxColourPoint {
    const real_t *x;
    const CT *color;
    int c;
    real_t e;
    __device__ __forceinline__ cd operator()(int64_t j) const { return cd{x[j], ((int)color[j] == c) ? e : (real_t)0}; }
};
struct CplxColumnPoint {
    const real_t *x;
    int64_t j;
    real_t e;
    __device__ __forceinline__ cd operator()(int64_t i) const { return cd{x[i], i == j ? e : (real_t)0}; }
};
template <typename CT, class F>
__global__ void __launch_bounds__(kBlock) k_csc_store_cols_cplx(F f, const real_t *__restrict__ x, const real_t *__restrict__ eps, int c_lo, int c_hi, fd_csc_store st)
{
    __shared__ __attribute__((aligned(16))) real_t s_win[kBlock / 64][FD_CSC_WAVE_CAP];
    const long long nblk = (st.col_end - st.col_begin + kBlock - 1) / kBlock, blk = fd_xcd_block(blockIdx.x, nblk);
    if (blk >= nblk) return;
    const long long j = st.col_begin + blk * kBlock + threadIdx.x;
    const bool in = j < st.col_end;
    const int a = in ? st.colptr[j - st.col_begin] : st.colptr[st.col_end - st.col_begin];
    const int b = in ? st.colptr[j - st.col_begin + 1] : a;
    const CT *color = (const CT *)st.color;
    const int c = in ? (int)color[j] : 0;
    const bool none = in && c == (int)(CT)(-1);
    const bool mine = in && !none && c >= c_lo && c < c_hi;
    fd_csc_wave_run<real_t> run;
    run.begin((real_t *)st.out, s_win[threadIdx.x >> 6], a, b, !in || mine || (none && c_lo == 0));
    if (none && c_lo == 0)
        for (int q = a; q < b; ++q) run.put(q, (real_t)0);
    if (mine) {
        const real_t h = eps[c];
        constexpr int U = 4;
        for (int q0 = a; q0 < b; q0 += U) {
            long long r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = st.rowval[q0 + u < b ? q0 + u : b - 1];
            real_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (q0 + u >= b) { v[u] = 0; continue; }
                cd w;
                if (st.valid_coloring) { const CplxColumnPoint X = {x, j, h}; w = f.template row<cd>(r[u], X); }
                else { const CplxColourPoint<CT> X = {x, color, c, h}; w = f.template row<cd>(r[u], X); }
                v[u] = w.im / h;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (q0 + u < b) run.put(q0 + u, v[u]);
        }
    }
    run.template flush<true>();
}

// ---- launchers --------------------------------------------------------------------------------------------------------------
template <typename T>
static int functor_family_launch(BuiltinF *b, void *fx, const void *x, int64_t nbatch, int64_t xs, int64_t fs, int64_t r0, int64_t r1, hipStream_t s)
{
    const dim3 g((unsigned)((r1 - r0 + kBlock - 1) / kBlock), (unsigned)nbatch, 1);
    if (b->family == FD_F_LAP7) {
        const Lap7F f = {(int)b->prm[0], (int)b->prm[1], (int)b->prm[2], fd_magic31((uint32_t)(b->prm[0] * b->prm[1])), fd_magic31((uint32_t)b->prm[0])};
        hipLaunchKernelGGL((k_f_rows<T, Lap7F>), g, dim3(kBlock), 0, s, (T *)fx, (const T *)x, f, xs, fs, r0, r1);
    } else {
        const SparseF f = {b->d_srow, b->d_scol};
        hipLaunchKernelGGL((k_f_rows<T, SparseF>), g, dim3(kBlock), 0, s, (T *)fx, (const T *)x, f, xs, fs, r0, r1);
    }
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// the lazy launcher of these families serves exactly one request: store column by column (forward / central); everything else is
// declined (the library materialises the points and calls the plain launcher)
template <typename CT>
static int functor_family_lazy(BuiltinF *b, const fd_lazy_points *lp, hipStream_t s)
{
    if (!lp->store || lp->store_kind != FD_STORE_CSC) return FD_LAZY_DECLINED;
    const fd_csc_store st = *(const fd_csc_store *)lp->store;
    if (st.elem_bytes != (int)sizeof(real_t) || st.color_bytes != (int)sizeof(CT) || st.M != b->M || st.N != b->N || st.col_end <= st.col_begin)
        return FD_LAZY_DECLINED;
    const unsigned g = fd_xcd_grid((st.col_end - st.col_begin + kBlock - 1) / kBlock);
    const int c_lo = lp->c_lo, c_hi = lp->c_lo + lp->ncolors;
    const real_t *x = (const real_t *)lp->x, *eps = (const real_t *)lp->eps;
    if (lp->is_complex) {       // the complex step: imag(row at x + i eps e_j) / eps, stored by this launch
        if (b->family == FD_F_LAP7) {
            const Lap7F f = {(int)b->prm[0], (int)b->prm[1], (int)b->prm[2], fd_magic31((uint32_t)(b->prm[0] * b->prm[1])), fd_magic31((uint32_t)b->prm[0])};
            hipLaunchKernelGGL((k_csc_store_cols_cplx<CT, Lap7F>), dim3(g), dim3(kBlock), 0, s, f, x, eps, c_lo, c_hi, st);
        } else {
            const SparseF f = {b->d_srow, b->d_scol};
            hipLaunchKernelGGL((k_csc_store_cols_cplx<CT, SparseF>), dim3(g), dim3(kBlock), 0, s, f, x, eps, c_lo, c_hi, st);
        }
        return hipGetLastError() == hipSuccess ? 0 : 4;
    }
#define FD_COLS(FT, fobj)                                                                                                                  \
    do {                                                                                                                                   \
        if (lp->pts == 2) hipLaunchKernelGGL((fd_csc_store_cols<real_t, CT, 1, FT>), dim3(g), dim3(kBlock), 0, s, fobj, x, eps, c_lo, c_hi, st); \
        else hipLaunchKernelGGL((fd_csc_store_cols<real_t, CT, 0, FT>), dim3(g), dim3(kBlock), 0, s, fobj, x, eps, c_lo, c_hi, st);              \
    } while (0)
    if (b->family == FD_F_LAP7) {
        const Lap7F f = {(int)b->prm[0], (int)b->prm[1], (int)b->prm[2], fd_magic31((uint32_t)(b->prm[0] * b->prm[1])), fd_magic31((uint32_t)b->prm[0])};
        if (st.valid_coloring) {      // the neighbourhood in registers (one perturbed coordinate per column)
            if (lp->pts == 2) hipLaunchKernelGGL((k_f_lap7_store_cols<CT, 1>), dim3(g), dim3(kBlock), 0, s, f, x, eps, c_lo, c_hi, st);
            else hipLaunchKernelGGL((k_f_lap7_store_cols<CT, 0>), dim3(g), dim3(kBlock), 0, s, f, x, eps, c_lo, c_hi, st);
            // (every launch visits all local columns: one that counted no mismatch has verified the pattern)
            if (st.note) hipLaunchKernelGGL(k_lap7_note, dim3(1), dim3(1), 0, s, st.note, lap7_note_key(f, st));
        } else {
            FD_COLS(Lap7F, f);
        }
    } else {
        const SparseF f = {b->d_srow, b->d_scol};
        // a locally banded pattern (the plan measured its reach) with a verified colouring: the workgroup's window of x, f(x) of its
        // rows and the rows' pattern are staged in LDS (column kernels), or the Jacobian is stored row by row (k_f_sparse_store_rows);
        // otherwise the plain column kernel.  Forward differences without fx_base (FD_LAZY_CAP_STORE_CSC_BASE): the row-wise kernel has
        // f(x) of its rows anyway; the generic column kernels evaluate the unperturbed row themselves.
        const int64_t reach = st.reach;
        const bool win = st.valid_coloring && reach > 0 && reach <= 700 && st.M == st.N;
        const double per_row = b->M > 0 ? (double)b->prm[2] / (double)b->M : 1.0;
        // the row-wise store: whichever of the two kernels the plan's note names does the work; until the verdict is known on the host
        // both are enqueued (they store the same bits), afterwards only the one that works
        const int cap_r = (int)std::min<int64_t>((int64_t)(kBlock * per_row * 1.25) + 64, 3072);
        const size_t lds_r = win ? sparse_rows_lds_bytes(reach, lp->ncolors, cap_r) : 0;
        const bool rows = win && st.note != nullptr && st.plan_serial != 0 && lds_r <= 64 * 1024 && b->d_sdest;
        const long long e_base = b->h_colptr[(size_t)st.col_begin], expect = (long long)b->h_colptr[(size_t)st.col_end] - e_base;
        const unsigned long long key = rows ? ((0x5BA25E0000000000ull ^ ((unsigned long long)(uintptr_t)b->d_sdest << 3) ^ ((unsigned long long)st.col_begin * 0x9E3779B97F4A7C15ull) ^
                                                (unsigned long long)st.col_end) & ~2ull) | 4ull : 0ull;
        int verdict = 0;                       // 0 unknown, 1 the plan's pattern is the functor's, 2 it is not
        BuiltinF::RowsMemo *memo = nullptr;
        if (rows) {
            std::lock_guard<std::mutex> lock(b->rows_mutex);
            if (b->rows_memo.size() > 1024 && !b->rows_memo.count(st.plan_serial)) {
                // (a process that keeps creating plans for one residual: forget the plans nothing is in flight for -- one that is
                //  still alive is simply checked again)
                for (auto it = b->rows_memo.begin(); it != b->rows_memo.end();) {
                    if (it->second.pending) { ++it; continue; }
                    if (it->second.ev) (void)hipEventDestroy(it->second.ev);
                    if (it->second.h_note) (void)hipHostFree(it->second.h_note);
                    it = b->rows_memo.erase(it);
                }
            }
            memo = &b->rows_memo[st.plan_serial];
            if (memo->verdict == 0 && memo->pending && hipEventQuery(memo->ev) == hipSuccess) {
                memo->pending = false;
                memo->verdict = *memo->h_note == key ? 1 : *memo->h_note == (key ^ 2ull) ? 2 : 0;
            }
            verdict = memo->verdict;
        }
        // the plan's pattern IS the residual's (verdict 1) and the plan keeps it by rows, sorted tile by tile (FD_PLAN_STORE_CSC_ROWS): the
        // row-wise store every separable functor gets (fd_csc_store_rows, include/fdjac_device.h) on the plan's lists
        if (verdict == 1 && st.row_ptr && st.row_pack && st.row_tile && st.col_begin == 0 && st.col_end == st.N && st.N >= 2 &&
            fd_csc_rows_lds_bytes<real_t>(reach, lp->ncolors, cap_r) <= 64 * 1024) {
            const unsigned gr = fd_xcd_grid((st.M + kBlock - 1) / kBlock);
            const size_t lds_e = fd_csc_ents_lds_bytes<real_t>(reach, lp->ncolors, st.ent_tile_max);
            const char *sw = test_switch("FDJAC_ROWS_ENTS");
            if (st.ent_col && st.ent_slot && st.ent_info && lds_e <= 64 * 1024 && (sw && *sw == '1')) {      // a thread per entry (fd_csc_store_ents)
                if (lp->pts == 2) hipLaunchKernelGGL((fd_csc_store_ents<real_t, CT, 1, SparseF>), dim3(gr), dim3(kBlock), lds_e, s, f, x, eps, c_lo, c_hi, st, (int)reach);
                else hipLaunchKernelGGL((fd_csc_store_ents<real_t, CT, 0, SparseF>), dim3(gr), dim3(kBlock), lds_e, s, f, x, eps, c_lo, c_hi, st, (int)reach);
                b->row_stores.fetch_add(1);
                return hipGetLastError() == hipSuccess ? 0 : 4;
            }
            const size_t lds_g = fd_csc_rows_lds_bytes<real_t>(reach, lp->ncolors, cap_r);
            if (lp->pts == 2) hipLaunchKernelGGL((fd_csc_store_rows<real_t, CT, 1, SparseF>), dim3(gr), dim3(kBlock), lds_g, s, f, x, eps, c_lo, c_hi, st, (int)reach, cap_r);
            else hipLaunchKernelGGL((fd_csc_store_rows<real_t, CT, 0, SparseF>), dim3(gr), dim3(kBlock), lds_g, s, f, x, eps, c_lo, c_hi, st, (int)reach, cap_r);
            b->row_stores.fetch_add(1);
            return hipGetLastError() == hipSuccess ? 0 : 4;
        }
        if (verdict != 1) {                    // a column kernel
            bool done = false;
            if (win) {
                const int64_t wrows = fd_csc_win_rlen(reach);
                const int cap = (int)std::min<int64_t>((int64_t)(wrows * per_row * 1.10) + 128, 16384);      // (a window's rows hold rows x per_row entries +- a few per cent; what does not fit is read from memory)
                const size_t sb = SparseF::stage_bytes(wrows, cap);
                const size_t lds = fd_csc_win_lds_bytes<real_t>(reach, lp->pts == 1 && st.fx_base != nullptr) + sb;
                const size_t lds_s = sparse_sorted_lds_bytes(reach, cap);
                if (lds_s <= 64 * 1024 && (lp->pts == 2 || st.fx_base != nullptr)) {      // the entry-balanced form (see k_f_sparse_store_sorted)
                    const unsigned long long gate = (rows && verdict == 0) ? key : 0ull;
                    if (lp->pts == 2) hipLaunchKernelGGL((k_f_sparse_store_sorted<CT, 1>), dim3(g), dim3(kBlock), lds_s, s, f, x, eps, c_lo, c_hi, st, (int)reach, cap, gate);
                    else hipLaunchKernelGGL((k_f_sparse_store_sorted<CT, 0>), dim3(g), dim3(kBlock), lds_s, s, f, x, eps, c_lo, c_hi, st, (int)reach, cap, gate);
                    done = true;
                } else if (lds <= 64 * 1024) {
                    if (lp->pts == 2) hipLaunchKernelGGL((fd_csc_store_cols_win<real_t, CT, 1, SparseF>), dim3(g), dim3(kBlock), lds, s, f, x, eps, c_lo, c_hi, st, (int)reach, (int)sb, cap);
                    else hipLaunchKernelGGL((fd_csc_store_cols_win<real_t, CT, 0, SparseF>), dim3(g), dim3(kBlock), lds, s, f, x, eps, c_lo, c_hi, st, (int)reach, (int)sb, cap);
                    done = true;
                }
            }
            if (!done) FD_COLS(SparseF, f);
        }
        if (verdict == 1) b->row_stores.fetch_add(1);
        if (rows && verdict != 2) {
            const long long row_lo = std::max<long long>(0, st.col_begin - reach), row_hi = std::min<long long>(st.M, st.col_end + reach);
            const unsigned gr = fd_xcd_grid((row_hi - row_lo + kBlock - 1) / kBlock);
            if (lp->pts == 2) hipLaunchKernelGGL((k_f_sparse_store_rows<CT, 1>), dim3(gr), dim3(kBlock), lds_r, s, f, b->d_sdest, e_base, x, eps, c_lo, c_hi, st, (int)reach, cap_r, row_lo, row_hi, key, expect, verdict == 1 ? 1 : 0);
            else hipLaunchKernelGGL((k_f_sparse_store_rows<CT, 0>), dim3(gr), dim3(kBlock), lds_r, s, f, b->d_sdest, e_base, x, eps, c_lo, c_hi, st, (int)reach, cap_r, row_lo, row_hi, key, expect, verdict == 1 ? 1 : 0);
            if (verdict == 0) {
                std::lock_guard<std::mutex> lock(b->rows_mutex);
                if (!memo->pending) {                        // the verdict on its way to the host: read at a later call, never waited for
                    if (!memo->h_note && hipHostMalloc((void **)&memo->h_note, sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) memo->h_note = nullptr;
                    if (memo->h_note && !memo->ev && hipEventCreateWithFlags(&memo->ev, hipEventDisableTiming) != hipSuccess) memo->ev = nullptr;
                    if (memo->h_note && memo->ev) {
                        *memo->h_note = 0;
                        if (hipMemcpyAsync(memo->h_note, st.note, sizeof(unsigned long long), hipMemcpyDeviceToHost, s) == hipSuccess &&
                            hipEventRecord(memo->ev, s) == hipSuccess)
                            memo->pending = true;
                    }
                }
            }
        }
    }
#undef FD_COLS
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
