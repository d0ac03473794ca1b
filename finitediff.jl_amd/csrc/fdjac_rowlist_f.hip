// Built-in f! families whose lazy launchers store the Jacobian of a GENERAL sparsity pattern through the plan's per-(row, colour)
// destination table (fd_rowlist_store, include/fdjac_device.h; FD_LAZY_CAP_STORE_ROWLIST):
//   FD_F_LAP7    zero-Dirichlet 7-point Laplacian on an nx x ny x nz grid + x[k]^2 x[k+1]   (3-D stencil: offsets +-1, +-nx, +-nx*ny)
//   FD_F_SPARSE  f_r = sum over the entries (r, j) of a GIVEN pattern, ascending j, of w(r, j) phi(x_j)   (any pattern)
// Row-centric: a thread owns a row, keeps (LAP7) or re-reads through the caches (SPARSE) the coordinates the row depends on, and
// for every stored entry the row feeds -- the plan's table lists them with the colour of their column -- evaluates the row at
// that colour's point, formed exactly as the reference forms it (x[j] + eps_c * (color[j] == c): x[j] + 0.0 elsewhere,
// src/jacobians.jl:562 / 603-604), subtracts (src/jacobians.jl:565 / 607), divides (IEEE) and stores (ext/FiniteDiffSparseArraysExt.jl:38-47).
// The operations of the hand-over path (materialised points -> plain f! -> k_decompress_*) on the same operands: same bits.
// Included by fdjac_builtin_f.hip (namespace fdjac, after BuiltinF).

constexpr real_t kSix = 6, kQuarter = 0.25, kEighth = 0.125;

// ---- FD_F_LAP7 -----------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T lap7_row(T c, T d, T s, T w, T e, T n, T u)
{
    return ((((((d + s) + w) + e) + n) + u) - kSix * c) + (c * c) * e;
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_f_lap7(T *__restrict__ fx, const T *__restrict__ x, int nx, int ny, int nz, int64_t xs, int64_t fs, int64_t r0, int64_t r1)
{
    const T *xb = x + (int64_t)blockIdx.y * xs;
    T *fb = fx + (int64_t)blockIdx.y * fs;
    const int64_t k = r0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= r1) return;
    const int64_t pl = (int64_t)nx * ny;
    const int l = (int)(k / pl), rem = (int)(k - (int64_t)l * pl), j = rem / nx, i = rem - j * nx;
    const T z = zero_of<T>();
    fb[k] = lap7_row<T>(xb[k], l > 0 ? xb[k - pl] : z, j > 0 ? xb[k - nx] : z, i > 0 ? xb[k - 1] : z, i < nx - 1 ? xb[k + 1] : z,
                        j < ny - 1 ? xb[k + nx] : z, l < nz - 1 ? xb[k + pl] : z);
}

// the storing launch: MODE 0 forward, 1 central
template <typename CT, int MODE>
__global__ void __launch_bounds__(kBlock)
k_f_lap7_store(const real_t *__restrict__ x, const real_t *__restrict__ eps, int c_lo, int c_hi, int nx, int ny, int nz, fd_rowlist_store st)
{
    const int64_t k = st.row_begin + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= st.row_end) return;
    const int e0 = st.rowptr[k - st.row_begin], e1 = st.rowptr[k - st.row_begin + 1];
    if (e0 == e1) return;
    const CT *color = (const CT *)st.color;
    const CT *ecolor = (const CT *)st.ecolor;
    const int none = (int)(CT)(-1);      // "none" is all-ones in CT
    const int64_t pl = (int64_t)nx * ny;
    const int l = (int)(k / pl), rem = (int)(k - (int64_t)l * pl), j = rem / nx, i = rem - j * nx;
    // the row's seven coordinates and the colours of their columns (a neighbour outside the grid: the value 0, no colour)
    const bool hd = l > 0, hs = j > 0, hw = i > 0, he = i < nx - 1, hn = j < ny - 1, hu = l < nz - 1;
    const real_t xc = x[k], xd = hd ? x[k - pl] : 0, xsn = hs ? x[k - nx] : 0, xw = hw ? x[k - 1] : 0, xe = he ? x[k + 1] : 0,
                 xn = hn ? x[k + nx] : 0, xu = hu ? x[k + pl] : 0;
    const int cc = (int)color[k], cd_ = hd ? (int)color[k - pl] : -2, cs = hs ? (int)color[k - nx] : -2, cw = hw ? (int)color[k - 1] : -2,
              ce = he ? (int)color[k + 1] : -2, cn = hn ? (int)color[k + nx] : -2, cu = hu ? (int)color[k + pl] : -2;
    real_t base = 0;
    if (MODE == 0) base = lap7_row<real_t>(xc, xd, xsn, xw, xe, xn, xu);
    real_t *out = (real_t *)st.out;
    const real_t z = 0;
    for (int e = e0; e < e1; ++e) {
        const int c = (int)ecolor[e];
        if (c == none) { if (c_lo == 0) out[st.dest[e]] = 0; continue; }
        if (c < c_lo || c >= c_hi) continue;
        const real_t h = eps[c];
        // the colour's point: x + h where the column has colour c, x + 0.0 elsewhere (coordinates outside the grid are the constant 0)
#define FD_PT(v, col, has, sg) ((has) ? (v) + ((col) == c ? (sg) : z) : (v))
        const real_t vp = lap7_row<real_t>(FD_PT(xc, cc, true, h), FD_PT(xd, cd_, hd, h), FD_PT(xsn, cs, hs, h), FD_PT(xw, cw, hw, h), FD_PT(xe, ce, he, h),
                                           FD_PT(xn, cn, hn, h), FD_PT(xu, cu, hu, h));
        real_t vm = base, div = h;
        if (MODE == 1) {
            // (the minus point holds x - 0.0 == x at the unperturbed coordinates)
#define FD_MT(v, col, has) (((has) && (col) == c) ? (v) - h : (v))
            vm = lap7_row<real_t>(FD_MT(xc, cc, true), FD_MT(xd, cd_, hd), FD_MT(xsn, cs, hs), FD_MT(xw, cw, hw), FD_MT(xe, ce, he), FD_MT(xn, cn, hn),
                                  FD_MT(xu, cu, hu));
#undef FD_MT
            div = 2 * h;
        }
#undef FD_PT
        out[st.dest[e]] = sub_exact(vp, vm) / div;
    }
}

// ---- FD_F_SPARSE ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ real_t sparse_weight(int64_t r, int64_t j) { return (real_t)1 + kEighth * (real_t)(int)((r + 3 * j) & 7); }
template <typename T> __device__ __forceinline__ T sparse_phi(T t) { return t + (kQuarter * t) * t; }

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_f_sparse(T *__restrict__ fx, const T *__restrict__ x, const int32_t *__restrict__ srow, const int32_t *__restrict__ scol, int64_t xs, int64_t fs,
           int64_t r0, int64_t r1)
{
    const T *xb = x + (int64_t)blockIdx.y * xs;
    T *fb = fx + (int64_t)blockIdx.y * fs;
    const int64_t r = r0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= r1) return;
    const int a0 = srow[r], a1 = srow[r + 1];
    T s = zero_of<T>();
    for (int a = a0; a < a1; ++a) {
        const int64_t j = scol[a];
        const T t = sparse_weight(r, j) * sparse_phi<T>(xb[j]);
        s = a == a0 ? t : s + t;
    }
    fb[r] = s;
}

template <typename CT, int MODE>
__global__ void __launch_bounds__(kBlock)
k_f_sparse_store(const real_t *__restrict__ x, const real_t *__restrict__ eps, int c_lo, int c_hi, const int32_t *__restrict__ srow,
                 const int32_t *__restrict__ scol, fd_rowlist_store st)
{
    const int64_t r = st.row_begin + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= st.row_end) return;
    const int e0 = st.rowptr[r - st.row_begin], e1 = st.rowptr[r - st.row_begin + 1];
    if (e0 == e1) return;
    const CT *color = (const CT *)st.color;
    const CT *ecolor = (const CT *)st.ecolor;
    const int none = (int)(CT)(-1);      // "none" is all-ones in CT
    const int a0 = srow[r], a1 = srow[r + 1];
    // the row at the point of colour c (c < 0: at x itself), sign sg: the reference's x[j] + eps_c * (color[j] == c) / x[j] - ...
    auto row_at = [&](int c, real_t h, bool minus) {
        real_t s = 0;
        for (int a = a0; a < a1; ++a) {
            const int64_t j = scol[a];
            real_t xv = x[j];
            if (c >= 0) {
                const bool hit = (int)color[j] == c;
                xv = minus ? (hit ? xv - h : xv) : xv + (hit ? h : (real_t)0);
            }
            const real_t t = sparse_weight(r, j) * sparse_phi<real_t>(xv);
            s = a == a0 ? t : s + t;
        }
        return s;
    };
    real_t base = 0;
    if (MODE == 0) base = row_at(-1, 0, false);
    real_t *out = (real_t *)st.out;
    for (int e = e0; e < e1; ++e) {
        const int c = (int)ecolor[e];
        if (c == none) { if (c_lo == 0) out[st.dest[e]] = 0; continue; }
        if (c < c_lo || c >= c_hi) continue;
        const real_t h = eps[c];
        const real_t vp = row_at(c, h, false);
        real_t vm = base, div = h;
        if (MODE == 1) { vm = row_at(c, h, true); div = 2 * h; }
        out[st.dest[e]] = sub_exact(vp, vm) / div;
    }
}

// ---- launchers --------------------------------------------------------------------------------------------------------------
template <typename T>
static int rowlist_family_launch(BuiltinF *b, void *fx, const void *x, int64_t nbatch, int64_t xs, int64_t fs, int64_t r0, int64_t r1, hipStream_t s)
{
    const dim3 g((unsigned)((r1 - r0 + kBlock - 1) / kBlock), (unsigned)nbatch, 1);
    if (b->family == FD_F_LAP7)
        hipLaunchKernelGGL((k_f_lap7<T>), g, dim3(kBlock), 0, s, (T *)fx, (const T *)x, (int)b->prm[0], (int)b->prm[1], (int)b->prm[2], xs, fs, r0, r1);
    else
        hipLaunchKernelGGL((k_f_sparse<T>), g, dim3(kBlock), 0, s, (T *)fx, (const T *)x, b->d_srow, b->d_scol, xs, fs, r0, r1);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// the lazy launcher of these families serves exactly one request: store through the destination table (forward / central);
// everything else is declined (the library materialises the points and calls the plain launcher)
template <typename CT>
static int rowlist_family_lazy(BuiltinF *b, const fd_lazy_points *lp, hipStream_t s)
{
    if (!lp->store || lp->store_kind != FD_STORE_ROWLIST || lp->is_complex) return FD_LAZY_DECLINED;
    const fd_rowlist_store st = *(const fd_rowlist_store *)lp->store;
    if (st.elem_bytes != (int)sizeof(real_t) || st.color_bytes != (int)sizeof(CT) || st.M != b->M || st.N != b->N || st.row_end <= st.row_begin)
        return FD_LAZY_DECLINED;
    const unsigned g = (unsigned)((st.row_end - st.row_begin + kBlock - 1) / kBlock);
    const int c_lo = lp->c_lo, c_hi = lp->c_lo + lp->ncolors;
    const real_t *x = (const real_t *)lp->x, *eps = (const real_t *)lp->eps;
    if (b->family == FD_F_LAP7) {
        if (lp->pts == 2)
            hipLaunchKernelGGL((k_f_lap7_store<CT, 1>), dim3(g), dim3(kBlock), 0, s, x, eps, c_lo, c_hi, (int)b->prm[0], (int)b->prm[1], (int)b->prm[2], st);
        else
            hipLaunchKernelGGL((k_f_lap7_store<CT, 0>), dim3(g), dim3(kBlock), 0, s, x, eps, c_lo, c_hi, (int)b->prm[0], (int)b->prm[1], (int)b->prm[2], st);
    } else {
        if (lp->pts == 2)
            hipLaunchKernelGGL((k_f_sparse_store<CT, 1>), dim3(g), dim3(kBlock), 0, s, x, eps, c_lo, c_hi, b->d_srow, b->d_scol, st);
        else
            hipLaunchKernelGGL((k_f_sparse_store<CT, 0>), dim3(g), dim3(kBlock), 0, s, x, eps, c_lo, c_hi, b->d_srow, b->d_scol, st);
    }
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
