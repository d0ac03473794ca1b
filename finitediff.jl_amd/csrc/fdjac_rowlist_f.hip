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

template <typename T> struct PlainPoint {      // a materialised point
    const T *x;
    __device__ T operator()(int64_t j) const { return x[j]; }
};

// ---- FD_F_LAP7 -----------------------------------------------------------------------------------------------------------------
struct Lap7F {
    int nx, ny, nz;
    template <typename T, class P> __device__ __forceinline__ T row(int64_t k, const P &X) const
    {
        const int64_t pl = (int64_t)nx * ny;
        const int l = (int)(k / pl), rem = (int)(k - (int64_t)l * pl), j = rem / nx, i = rem - j * nx;
        const T z = zero_of<T>();
        const T c = X(k);
        const T d = l > 0 ? X(k - pl) : z, s = j > 0 ? X(k - nx) : z, w = i > 0 ? X(k - 1) : z, e = i < nx - 1 ? X(k + 1) : z,
                n = j < ny - 1 ? X(k + nx) : z, u = l < nz - 1 ? X(k + pl) : z;
        return ((((((d + s) + w) + e) + n) + u) - kSix * c) + (c * c) * e;
    }
    template <class P> __device__ __forceinline__ real_t operator()(long long k, const P &X) const { return row<real_t>(k, X); }
};

// ---- FD_F_SPARSE ----------------------------------------------------------------------------------------------------------------
struct SparseF {
    const int32_t *srow, *scol;      // the pattern by rows, ascending columns (device)
    template <typename T, class P> __device__ __forceinline__ T row(int64_t r, const P &X) const
    {
        const int a0 = srow[r], a1 = srow[r + 1];
        T s = zero_of<T>();
        for (int a = a0; a < a1; ++a) {
            const int64_t j = scol[a];
            const T v = X(j);
            const T t = ((real_t)1 + kEighth * (real_t)(int)((r + 3 * j) & 7)) * (v + (kQuarter * v) * v);
            s = a == a0 ? t : s + t;
        }
        return s;
    }
    template <class P> __device__ __forceinline__ real_t operator()(long long r, const P &X) const { return row<real_t>(r, X); }
};

template <typename T, class F>
__global__ void __launch_bounds__(kBlock) k_f_rows(T *__restrict__ fx, const T *__restrict__ x, F f, int64_t xs, int64_t fs, int64_t r0, int64_t r1)
{
    const int64_t r = r0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= r1) return;
    const PlainPoint<T> P = {x + (int64_t)blockIdx.y * xs};
    fx[(int64_t)blockIdx.y * fs + r] = f.template row<T>(r, P);
}

// ---- launchers --------------------------------------------------------------------------------------------------------------
template <typename T>
static int rowlist_family_launch(BuiltinF *b, void *fx, const void *x, int64_t nbatch, int64_t xs, int64_t fs, int64_t r0, int64_t r1, hipStream_t s)
{
    const dim3 g((unsigned)((r1 - r0 + kBlock - 1) / kBlock), (unsigned)nbatch, 1);
    if (b->family == FD_F_LAP7) {
        const Lap7F f = {(int)b->prm[0], (int)b->prm[1], (int)b->prm[2]};
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
static int rowlist_family_lazy(BuiltinF *b, const fd_lazy_points *lp, hipStream_t s)
{
    if (!lp->store || lp->store_kind != FD_STORE_CSC || lp->is_complex) return FD_LAZY_DECLINED;
    const fd_csc_store st = *(const fd_csc_store *)lp->store;
    if (st.elem_bytes != (int)sizeof(real_t) || st.color_bytes != (int)sizeof(CT) || st.M != b->M || st.N != b->N || st.col_end <= st.col_begin ||
        (lp->pts == 1 && !st.fx_base))
        return FD_LAZY_DECLINED;
    const unsigned g = (unsigned)((st.col_end - st.col_begin + kBlock - 1) / kBlock);
    const int c_lo = lp->c_lo, c_hi = lp->c_lo + lp->ncolors;
    const real_t *x = (const real_t *)lp->x, *eps = (const real_t *)lp->eps;
#define FD_COLS(FT, fobj)                                                                                                                  \
    do {                                                                                                                                   \
        if (lp->pts == 2) hipLaunchKernelGGL((fd_csc_store_cols<real_t, CT, 1, FT>), dim3(g), dim3(kBlock), 0, s, fobj, x, eps, c_lo, c_hi, st); \
        else hipLaunchKernelGGL((fd_csc_store_cols<real_t, CT, 0, FT>), dim3(g), dim3(kBlock), 0, s, fobj, x, eps, c_lo, c_hi, st);              \
    } while (0)
    if (b->family == FD_F_LAP7) {
        const Lap7F f = {(int)b->prm[0], (int)b->prm[1], (int)b->prm[2]};
        FD_COLS(Lap7F, f);
    } else {
        const SparseF f = {b->d_srow, b->d_scol};
        FD_COLS(SparseF, f);
    }
#undef FD_COLS
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
