/*
 * Plain-C clients of libfdjac, ONE PER METHOD of the Julia shim (finitediff.jl_amd/julia/FiniteDiffMI355X.jl): each
 * function below performs exactly the foreign calls of the shim method named in its comment, with Julia-layout
 * arrays (1-based Int64 colptr / rowval / colorvec, column-major matrices) and -- as the shim's AMDGPU.jl methods do
 * -- DEVICE pointers for x and J's storage, on the caller's own stream, through fd_jacobian_async.  Julia is not
 * installed in the build image; these clients are what executes the shim's call sequences on the GPU box
 * (tests/test_gpu_edge.py::test_c_clients_every_plan_kind).
 *
 * No HIP headers: like a Julia process, the client reaches the HIP runtime through its C entry points only.
 *
 *   gcc -O2 -Iinclude examples/c_abi_clients.c -o c_abi_clients -Lfinitediff.jl_amd/lib -lfdjac \
 *       -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/finitediff.jl_amd/lib -Wl,-rpath,/opt/rocm/lib
 *   ./c_abi_clients all        # or: csc csc_device csc_dense coo_dense entries dense tridiagonal banded blockbanded csc_f32 jvp solve host complex_x resize dropin [N reps]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fdjac.h"

/* the five HIP runtime entry points a host language binds (AMDGPU.jl: Mem.alloc, copyto!, HIPStream, synchronize) */
extern int hipMalloc(void **ptr, size_t size);
extern int hipFree(void *ptr);
extern int hipMemcpy(void *dst, const void *src, size_t size, int kind); /* 1 = host->device, 2 = device->host */
extern int hipStreamCreate(void **stream);
extern int hipStreamSynchronize(void *stream);
extern int hipStreamDestroy(void *stream);
extern int hipMemset(void *dst, int value, size_t size);

#define CHECK(call)                                                                                \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        if (rc_ != 0) {                                                                            \
            fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, fd_last_error()); \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

static void *g_stream = NULL;   /* the caller's stream: every plan is created on a context bound to it */
static fd_ctx *g_ctx = NULL;

static void *to_dev(const void *h, size_t bytes)
{
    void *d = NULL;
    if (hipMalloc(&d, bytes ? bytes : 16) != 0) return NULL;
    if (h && bytes) hipMemcpy(d, h, bytes, 1);
    return d;
}
static double *dev_nan(size_t n)
{
    double *h = malloc(sizeof(double) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) h[i] = NAN;
    double *d = to_dev(h, sizeof(double) * n);
    free(h);
    return d;
}
static void from_dev(void *h, const void *d, size_t bytes) { hipMemcpy(h, d, bytes, 2); }

static double *make_x(int64_t N)
{
    double *x = malloc(sizeof(double) * (size_t)N);
    for (int64_t j = 0; j < N; ++j) x[j] = 0.5 + 0.25 * sin((double)(j + 1));
    return x;
}
static int64_t *cyclic_colors(int64_t N, int64_t C)
{
    int64_t *c = malloc(sizeof(int64_t) * (size_t)N);
    for (int64_t j = 0; j < N; ++j) c[j] = j % C + 1;   /* mod1(j, C) */
    return c;
}
/* sparse(Tridiagonal(...)) pattern as Julia stores it */
static void tridiag_csc(int64_t N, int64_t **colptr, int64_t **rowval)
{
    *colptr = malloc(sizeof(int64_t) * (size_t)(N + 1));
    *rowval = malloc(sizeof(int64_t) * (size_t)(3 * N));
    int64_t p = 0;
    for (int64_t j = 1; j <= N; ++j) {
        (*colptr)[j - 1] = p + 1;
        for (int64_t r = j - 1; r <= j + 1; ++r)
            if (r >= 1 && r <= N) (*rowval)[p++] = r;
    }
    (*colptr)[N] = p + 1;
}
/* analytic Jacobian of the tridiag_nl fixture, f_i = x[i-1] - 2x[i] + x[i+1] + x[i]^2 x[i+1] (0-based r, c) */
static double tridiag_nl_J(const double *x, int64_t N, int64_t r, int64_t c)
{
    const double xp = r + 1 < N ? x[r + 1] : 0.0;
    if (r == c) return -2.0 + 2.0 * x[r] * xp;
    if (c == r + 1) return 1.0 + x[r] * x[r];
    if (c + 1 == r) return 1.0;
    return 0.0;
}
static int report(const char *name, double worst, double tol, int64_t fcalls, int64_t want_calls)
{
    const int ok = worst <= tol && fcalls == want_calls;
    printf("%-12s max|J - analytic| = %.3e (tol %.1e)  f!_evaluations = %lld (expected %lld)  %s\n", name, worst, tol,
           (long long)fcalls, (long long)want_calls, ok ? "ok" : "FAILED");
    return ok ? 0 : 3;
}
static int new_f(int family, const int64_t *prm, int nprm, fd_f_launch *f, void **fctx)
{
    return fd_builtin_f_create(g_ctx, family, prm, nprm, f, fctx);
}
static int64_t f_points(void *fctx)
{
    int64_t l = 0, p = 0;
    fd_builtin_f_counts(fctx, &l, &p);
    return p;
}
/* install_lazy!(plan, f) of the shim */
static int install_lazy(fd_plan *plan, void *fctx)
{
    fd_f_launch_lazy lz = NULL;
    int caps = 0;
    if (fd_builtin_f_lazy(fctx, &lz) != FD_OK) return 0;
    if (fd_plan_set_lazy_f(plan, lz) != FD_OK) return 1;
    fd_builtin_f_lazy_caps(fctx, &caps);
    return fd_plan_set_lazy_caps(plan, caps);
}

/* shim: make_plan(::SparseMatrixCSC J, ::SparseMatrixCSC sparsity) -> fd_plan_create_csc;
         finite_difference_jacobian!(J::ROCSparseMatrixCSC, f::DeviceF, x::ROCVector, cache) -> fd_jacobian_async */
static int client_csc(int fdtype)
{
    const int64_t N = 100003;
    int64_t *colptr, *rowval, *colors = cyclic_colors(N, 3);
    tridiag_csc(N, &colptr, &rowval);
    const int64_t nnz = colptr[N] - 1;
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *nzd = dev_nan((size_t)nnz);
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = fdtype;
    CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &plan));
    CHECK(install_lazy(plan, fctx));
    void *outs[3] = {nzd, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *nz = malloc(sizeof(double) * (size_t)nnz);
    from_dev(nz, nzd, sizeof(double) * (size_t)nnz);
    double worst = 0;
    for (int64_t j = 0; j < N; ++j)
        for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
            const double d = fabs(nz[p] - tridiag_nl_J(x, N, rowval[p] - 1, j));
            if (!(d <= worst)) worst = d;
        }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(nzd); free(nz); free(x); free(colptr); free(rowval); free(colors);
    const char *nm = fdtype == FD_FORWARD ? "csc/forward" : fdtype == FD_CENTRAL ? "csc/central" : "csc/complex";
    return report(nm, worst, fdtype == FD_FORWARD ? 2e-6 : fdtype == FD_CENTRAL ? 2e-8 : 1e-13, calls,
                  fdtype == FD_FORWARD ? 4 : fdtype == FD_CENTRAL ? 6 : 3);
}

/* shim: make_plan(J::SparseMatrixCSC, ...; store_csc = true) with a residual of ANY pattern (fd_builtin_f_create_sparse): the column
   store -- every stored entry evaluates its own row at the point perturbed in its column, differences, divides and stores, column
   by column, in ONE launch after the step sizes -- for forward, central and the complex step (FD_LAZY_CAP_STORE_CSC_COMPLEX).
   Pattern: column j holds rows j-5, j-2, j, j+3, j+7 (13 cyclic colours); f_r = sum_j w(r, j) (x_j + x_j^2 / 4). */
static int client_csc_store_cols(int fdtype)
{
    const int64_t N = 20011, C = 13;
    static const int64_t off[5] = {-5, -2, 0, 3, 7};
    int64_t *colptr = malloc(sizeof(int64_t) * (size_t)(N + 1)), *rowval = malloc(sizeof(int64_t) * (size_t)(5 * N)), *colors = cyclic_colors(N, C);
    int64_t p = 0;
    for (int64_t j = 0; j < N; ++j) {
        colptr[j] = p + 1;
        for (int q = 0; q < 5; ++q)
            if (j + off[q] >= 0 && j + off[q] < N) rowval[p++] = j + off[q] + 1;
    }
    colptr[N] = p + 1;
    const int64_t nnz = p;
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *nzd = dev_nan((size_t)nnz);
    fd_f_launch f; void *fctx; fd_plan *plan;
    CHECK(fd_builtin_f_create_sparse(g_ctx, N, N, colptr, rowval, 8, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = fdtype; o.flags = FD_PLAN_STORE_CSC | FD_PLAN_STORE_CSC_ALWAYS;
    CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &plan));
    CHECK(install_lazy(plan, fctx));
    void *outs[3] = {nzd, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    int64_t table = 0, stored = 0;
    CHECK(fd_plan_info(plan, FD_INFO_STORE_CSC, &table));
    CHECK(fd_plan_info(plan, FD_INFO_LAZY_STORE, &stored));
    double *nz = malloc(sizeof(double) * (size_t)nnz);
    from_dev(nz, nzd, sizeof(double) * (size_t)nnz);
    double worst = 0;
    for (int64_t j = 0; j < N; ++j)
        for (int64_t q = colptr[j] - 1; q < colptr[j + 1] - 1; ++q) {
            const int64_t r = rowval[q] - 1;
            const double want = (1.0 + 0.125 * (double)((r + 3 * j) % 8)) * (1.0 + 0.5 * x[j]);
            const double d = fabs(nz[q] - want);
            if (!(d <= worst)) worst = d;
        }
    if (!(table > 0 && stored == 1)) { printf("csc_store_cols: the column store did not run (table %lld, stored %lld)\n", (long long)table, (long long)stored); worst = NAN; }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(nzd); free(nz); free(x); free(colptr); free(rowval); free(colors);
    const char *nm = fdtype == FD_FORWARD ? "cols/forward" : fdtype == FD_CENTRAL ? "cols/central" : "cols/complex";
    return report(nm, worst, fdtype == FD_FORWARD ? 2e-6 : fdtype == FD_CENTRAL ? 2e-8 : 1e-13, calls,
                  fdtype == FD_FORWARD ? 1 + C : fdtype == FD_CENTRAL ? 2 * C : C);
}

/* shim: DeviceF(src::String, functor, params) -> fd_f_compile_rows: the residual handed over as SOURCE, compiled by the library with
   hiprtc; make_plan(...; store_csc_always = true) + install of the compiled lazy launcher: the Jacobian is the step-size launch + ONE
   launch of the column store instantiated for the functor.  Checked against the analytic Jacobian and, bit for bit, against the
   built-in family's result; the timing lines compare the opaque call (plain launcher, materialised points) with the one-launch call. */
static double now_ms(void);
static const char *kJitTridiagNL =
    "struct TridiagNL {\n"
    "    long long n;\n"
    "    template <class P> __device__ real_t operator()(long long i, const P &X) const\n"
    "    {\n"
    "        const real_t xi = X(i), xm = X(i > 0 ? i - 1 : i), xp = X(i + 1 < n ? i + 1 : i);\n"
    "        const real_t a = i > 0 ? xm : (real_t)0, b = i + 1 < n ? xp : (real_t)0;\n"
    "        real_t v = (a - (real_t)2 * xi) + b;\n"
    "        v = v + (xi * xi) * b;\n"
    "        return v;\n"
    "    }\n"
    "};\n";
static int client_jit(int64_t N, int reps)
{
    int64_t *colptr, *rowval, *colors = cyclic_colors(N, 3);
    tridiag_csc(N, &colptr, &rowval);
    const int64_t nnz = colptr[N] - 1;
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *nzd = dev_nan((size_t)nnz), *refd = dev_nan((size_t)nnz), *opd = dev_nan((size_t)nnz);
    fd_f_launch fb, fj; void *fbctx, *fjctx; fd_f_launch_lazy lz = NULL; int caps = 0;
    fd_plan *pb, *pj, *po;
    const int64_t prm[1] = {N};
    const long long params[1] = {(long long)N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &fb, &fbctx));
    int rc = fd_f_compile_rows(g_ctx, kJitTridiagNL, "TridiagNL", params, sizeof params, N, N, 8, &fj, &lz, &caps, &fjctx);
    if (rc != FD_OK) { fprintf(stderr, "fd_f_compile_rows -> %d: %s\n%s\n", rc, fd_last_error(), fd_f_compile_log()); return 3; }
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_FORWARD;
    CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &pb));
    CHECK(install_lazy(pb, fbctx));
    CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &po));           /* opaque: no lazy launcher */
    o.flags = FD_PLAN_STORE_CSC | FD_PLAN_STORE_CSC_ALWAYS;
    CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &pj));
    CHECK(fd_plan_set_lazy_f(pj, lz));
    CHECK(fd_plan_set_lazy_caps(pj, caps));
    void *outs[3] = {nzd, NULL, NULL}, *outr[3] = {refd, NULL, NULL}, *outo[3] = {opd, NULL, NULL};
    CHECK(fd_jacobian_async(pb, fb, fbctx, xd, NULL, -1.0, -1.0, 1.0, outr));
    CHECK(fd_jacobian_async(pj, fj, fjctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_jacobian_async(po, fj, fjctx, xd, NULL, -1.0, -1.0, 1.0, outo));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *nz = malloc(sizeof(double) * (size_t)nnz), *ref = malloc(sizeof(double) * (size_t)nnz), *op = malloc(sizeof(double) * (size_t)nnz);
    from_dev(nz, nzd, sizeof(double) * (size_t)nnz);
    from_dev(ref, refd, sizeof(double) * (size_t)nnz);
    from_dev(op, opd, sizeof(double) * (size_t)nnz);
    double worst = 0;
    for (int64_t j = 0; j < N; ++j)
        for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
            const double d = fabs(nz[p] - tridiag_nl_J(x, N, rowval[p] - 1, j));
            if (!(d <= worst)) worst = d;
        }
    const int same = memcmp(nz, ref, sizeof(double) * (size_t)nnz) == 0 && memcmp(op, ref, sizeof(double) * (size_t)nnz) == 0;
    int64_t info_store = 0, fc = 0;
    fd_plan_info(pj, FD_INFO_LAZY_STORE, &info_store);
    fd_plan_info(pj, FD_INFO_FCALLS_LAST, &fc);
    /* what the two routes cost: wall clock of `reps` back-to-back calls each, synchronised at the end */
    double t_jit = 0, t_opq = 0, t_blt = 0;
    for (int k = 0; k < 3; ++k) {
        fd_plan *pl = k == 0 ? pj : k == 1 ? po : pb;
        void **oo = k == 0 ? outs : k == 1 ? outo : outr;
        for (int r = 0; r < 3; ++r) CHECK(fd_jacobian_async(pl, k == 2 ? fb : fj, k == 2 ? fbctx : fjctx, xd, NULL, -1.0, -1.0, 1.0, oo));
        CHECK(fd_ctx_synchronize(g_ctx));
        const double t0 = now_ms();
        for (int r = 0; r < reps; ++r) CHECK(fd_jacobian_async(pl, k == 2 ? fb : fj, k == 2 ? fbctx : fjctx, xd, NULL, -1.0, -1.0, 1.0, oo));
        CHECK(fd_ctx_synchronize(g_ctx));
        const double t = (now_ms() - t0) / reps;
        if (k == 0) t_jit = t; else if (k == 1) t_opq = t; else t_blt = t;
    }
    printf("jit N=%lld: runtime-compiled functor, one storing launch (fd_band_store_cols) %.4f ms | same functor as an opaque f! %.4f ms | built-in family (band store) %.4f ms\n",
           (long long)N, t_jit, t_opq, t_blt);
    CHECK(fd_plan_destroy(pb)); CHECK(fd_plan_destroy(pj)); CHECK(fd_plan_destroy(po));
    CHECK(fd_builtin_f_destroy(fbctx)); CHECK(fd_f_compiled_destroy(fjctx));
    hipFree(xd); hipFree(nzd); hipFree(refd); hipFree(opd); free(nz); free(ref); free(op); free(x); free(colptr); free(rowval); free(colors);
    if (!same || !info_store) { printf("jit          bits differ from the built-in family / column store not taken (%d, %lld)  FAILED\n", same, (long long)info_store); return 3; }
    return report("jit", worst, 2e-6, fc, 4);
}

/* shim: DeviceF{T}(src, terms, plan, M, N) on a plan made with PlanOpts(fd; store_rows = true) -> fd_plan_row_lists + fd_f_compile_terms:
   a residual that is SEPARABLE on the Jacobian's pattern, given by its term alone; the Jacobian is the step-size launch + ONE
   row-wise launch (fd_csc_store_rows).  Pattern: column j holds the rows j - 3, j, j + 2 that exist (not an exact band: a general
   CSC plan), colours mod1(j, 6) (columns sharing a row differ by 2, 3 or 5).  f_r = sum over the entries (r, j), ascending j, of
   w(r, j) (x_j + x_j^2 / 4), w = 1 + ((r + 3 j) mod 8) / 8 -- the built-in sparse family's residual, so the hand-over path of that
   family is the bit reference; analytic: J[r, j] = w(r, j) (1 + x_j / 2). */
static const char kTermsSource[] =
    "struct SparseTerms {\n"
    "    template <class T> __device__ T term(long long r, long long j, T v) const\n"
    "    { return ((real_t)1 + (real_t)0.125 * (real_t)(int)((r + 3 * j) & 7)) * (v + ((real_t)0.25 * v) * v); }\n"
    "};\n";
static int client_terms(int64_t N, int fdtype)
{
    int64_t *colptr = malloc(sizeof(int64_t) * (size_t)(N + 1)), *rowval = malloc(sizeof(int64_t) * (size_t)(3 * N)), *colors = cyclic_colors(N, 6);
    int64_t nnz = 0;
    const int64_t offs[3] = {-3, 0, 2};
    for (int64_t j = 0; j < N; ++j) {
        colptr[j] = nnz + 1;
        for (int t = 0; t < 3; ++t)
            if (j + offs[t] >= 0 && j + offs[t] < N) rowval[nnz++] = j + offs[t] + 1;
    }
    colptr[N] = nnz + 1;
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *nzd = dev_nan((size_t)nnz), *refd = dev_nan((size_t)nnz);
    fd_f_launch fb, ft; void *fbctx, *ftctx; fd_f_launch_lazy lz = NULL; int caps = 0;
    fd_plan *pr, *ph;
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = fdtype;
    CHECK(fd_builtin_f_create_sparse(g_ctx, N, N, colptr, rowval, 8, 1, &fb, &fbctx));
    CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &ph));           /* the reference bits: hand-over path */
    o.flags = FD_PLAN_STORE_CSC | FD_PLAN_STORE_CSC_ROWS;
    CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &pr));
    const void *row_ptr = NULL, *row_col = NULL, *row_slot = NULL; int64_t entries = 0; uint64_t serial = 0;
    CHECK(fd_plan_row_lists(pr, &row_ptr, &row_col, &row_slot, &entries, &serial));
    int rc = fd_f_compile_terms(g_ctx, kTermsSource, "SparseTerms", NULL, 0, N, N, 8, row_ptr, row_col, serial, &ft, &lz, &caps, &ftctx);
    if (rc != FD_OK) { fprintf(stderr, "fd_f_compile_terms -> %d: %s\n%s\n", rc, fd_last_error(), fd_f_compile_log()); return 3; }
    CHECK(fd_plan_set_lazy_f(pr, lz));
    CHECK(fd_plan_set_lazy_caps(pr, caps));
    void *outs[3] = {nzd, NULL, NULL}, *outr[3] = {refd, NULL, NULL};
    CHECK(fd_jacobian_async(ph, fb, fbctx, xd, NULL, -1.0, -1.0, 1.0, outr));
    CHECK(fd_jacobian_async(pr, ft, ftctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *nz = malloc(sizeof(double) * (size_t)nnz), *ref = malloc(sizeof(double) * (size_t)nnz);
    from_dev(nz, nzd, sizeof(double) * (size_t)nnz);
    from_dev(ref, refd, sizeof(double) * (size_t)nnz);
    double worst = 0;
    for (int64_t j = 0; j < N; ++j)
        for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
            const int64_t r = rowval[p] - 1;
            const double want = (1.0 + 0.125 * (double)((r + 3 * j) & 7)) * (1.0 + 0.5 * x[j]);
            const double d = fabs(nz[p] - want);
            if (!(d <= worst)) worst = d;
        }
    const int same = memcmp(nz, ref, sizeof(double) * (size_t)nnz) == 0;
    int64_t info_store = 0, fc = 0, rows = 0;
    fd_plan_info(pr, FD_INFO_LAZY_STORE, &info_store);
    fd_plan_info(pr, FD_INFO_FCALLS_LAST, &fc);
    CHECK(fd_f_compiled_row_stores(ftctx, &rows));
    CHECK(fd_plan_destroy(pr)); CHECK(fd_plan_destroy(ph));
    CHECK(fd_builtin_f_destroy(fbctx)); CHECK(fd_f_compiled_destroy(ftctx));
    hipFree(xd); hipFree(nzd); hipFree(refd); free(nz); free(ref); free(x); free(colptr); free(rowval); free(colors);
    if (!same || !info_store || rows != 1 || entries != nnz) {
        printf("terms        bits differ from the hand-over path / row-wise store not taken (same %d, store %lld, row-wise launches %lld, entries %lld)  FAILED\n",
               same, (long long)info_store, (long long)rows, (long long)entries);
        return 3;
    }
    return report("terms", worst, fdtype == FD_CENTRAL ? 1e-8 : 2e-6, fc, fdtype == FD_CENTRAL ? 12 : 7);
}

/* shim (AMDGPU extension): make_plan(::ROCSparseMatrixCSC J, sparsity === J, colorvec::ROCVector) -> fd_plan_create_csc_device:
   colPtr / rowVal / colorvec already live on the device (rocSPARSE CSC: Int32, 1-based); nothing crosses PCIe, the plan is
   compiled by kernels.  Checked against the host-pattern plan through fd_plan_checksum and against the analytic Jacobian. */
static int client_csc_device(void)
{
    const int64_t N = 200003;
    int64_t *colptr, *rowval, *colors = cyclic_colors(N, 3);
    tridiag_csc(N, &colptr, &rowval);
    const int64_t nnz = colptr[N] - 1;
    int32_t *cp32 = malloc(sizeof(int32_t) * (size_t)(N + 1)), *rv32 = malloc(sizeof(int32_t) * (size_t)nnz), *cv32 = malloc(sizeof(int32_t) * (size_t)N);
    for (int64_t j = 0; j <= N; ++j) cp32[j] = (int32_t)colptr[j];
    for (int64_t p = 0; p < nnz; ++p) rv32[p] = (int32_t)rowval[p];
    for (int64_t j = 0; j < N; ++j) cv32[j] = (int32_t)colors[j];
    void *cpd = to_dev(cp32, sizeof(int32_t) * (size_t)(N + 1)), *rvd = to_dev(rv32, sizeof(int32_t) * (size_t)nnz), *cvd = to_dev(cv32, sizeof(int32_t) * (size_t)N);
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *nzd = dev_nan((size_t)nnz);
    fd_f_launch f; void *fctx; fd_plan *plan, *ref;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_FORWARD;
    CHECK(fd_plan_create_csc_device(g_ctx, N, N, cpd, rvd, 4, 1, cvd, 4, &o, &plan));
    CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &ref));
    uint64_t ca = 0, cb = 1;
    CHECK(fd_plan_checksum(plan, &ca)); CHECK(fd_plan_checksum(ref, &cb));
    int64_t on_dev = 0;
    CHECK(fd_plan_info(plan, FD_INFO_BUILT_ON_DEVICE, &on_dev));
    CHECK(install_lazy(plan, fctx));
    void *outs[3] = {nzd, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *nz = malloc(sizeof(double) * (size_t)nnz);
    from_dev(nz, nzd, sizeof(double) * (size_t)nnz);
    double worst = 0;
    for (int64_t j = 0; j < N; ++j)
        for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
            const double d = fabs(nz[p] - tridiag_nl_J(x, N, rowval[p] - 1, j));
            if (!(d <= worst)) worst = d;
        }
    if (ca != cb || !on_dev) { fprintf(stderr, "csc_device: plan differs from the host-pattern plan (%llx vs %llx, built on device %lld)\n",
                                       (unsigned long long)ca, (unsigned long long)cb, (long long)on_dev); worst = INFINITY; }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_plan_destroy(ref)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(nzd); hipFree(cpd); hipFree(rvd); hipFree(cvd);
    free(nz); free(x); free(colptr); free(rowval); free(colors); free(cp32); free(rv32); free(cv32);
    return report("csc_device", worst, 2e-6, calls, 4);
}

/* shim: make_plan(::Matrix J, ::SparseMatrixCSC sparsity) -> fd_plan_create_csc_dense (ext/FiniteDiffSparseArraysExt.jl:20-28) */
static int client_csc_dense(void)
{
    const int64_t N = 300;
    int64_t *colptr, *rowval, *colors = cyclic_colors(N, 3);
    tridiag_csc(N, &colptr, &rowval);
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *Jd = dev_nan((size_t)(N * N));
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_CENTRAL;
    CHECK(fd_plan_create_csc_dense(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &plan));
    void *outs[3] = {Jd, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *J = malloc(sizeof(double) * (size_t)(N * N));
    from_dev(J, Jd, sizeof(double) * (size_t)(N * N));
    double worst = 0;
    for (int64_t c = 0; c < N; ++c)
        for (int64_t r = 0; r < N; ++r) {   /* entries outside the pattern are zero: fill_matrix!(J, false) */
            const double d = fabs(J[r + N * c] - tridiag_nl_J(x, N, r, c));
            if (!(d <= worst)) worst = d;
        }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(Jd); free(J); free(x); free(colptr); free(rowval); free(colors);
    return report("csc_dense", worst, 2e-8, calls, 6);
}

/* shim: make_plan(::Matrix J, ::DenseMatrix sparsity) -> FiniteDiff._findstructralnz + fd_plan_create_coo_dense
   (src/jacobians.jl:473-488, src/iteration_utils.jl:25-32) */
static int client_coo_dense(void)
{
    const int64_t N = 200;
    int64_t *rows = malloc(sizeof(int64_t) * (size_t)(3 * N)), *cols = malloc(sizeof(int64_t) * (size_t)(3 * N));
    int64_t nnz = 0, *colors = cyclic_colors(N, 3);
    for (int64_t j = 1; j <= N; ++j)            /* column-major scan of the dense 0/1 pattern matrix */
        for (int64_t i = 1; i <= N; ++i)
            if (i >= j - 1 && i <= j + 1) { rows[nnz] = i; cols[nnz] = j; ++nnz; }
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *Jd = dev_nan((size_t)(N * N));
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_COMPLEX;
    CHECK(fd_plan_create_coo_dense(g_ctx, N, N, rows, cols, nnz, 8, 1, colors, 8, &o, &plan));
    void *outs[3] = {Jd, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *J = malloc(sizeof(double) * (size_t)(N * N));
    from_dev(J, Jd, sizeof(double) * (size_t)(N * N));
    double worst = 0;
    for (int64_t c = 0; c < N; ++c)
        for (int64_t r = 0; r < N; ++r) {
            const double d = fabs(J[r + N * c] - tridiag_nl_J(x, N, r, c));
            if (!(d <= worst)) worst = d;
        }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(Jd); free(J); free(x); free(rows); free(cols); free(colors);
    return report("coo_dense", worst, 1e-13, calls, 3);
}

/* shim: make_plan_entries(J::SparseMatrixCSC, sparsity::SparseMatrixCSC) -- J stores MORE entries than the pattern
   (a pentadiagonal J, a tridiagonal sparsity): (row, col, position in J.nzval) enumerated once, fd_plan_create_entries
   (ext/FiniteDiffSparseArraysExt.jl:20-28 through J's setindex!) */
static int client_entries(void)
{
    const int64_t N = 5000;
    int64_t *scolptr, *srowval, *colors = cyclic_colors(N, 3);
    tridiag_csc(N, &scolptr, &srowval);
    /* J: pentadiagonal CSC */
    int64_t *jcolptr = malloc(sizeof(int64_t) * (size_t)(N + 1)), *jrowval = malloc(sizeof(int64_t) * (size_t)(5 * N));
    int64_t q = 0;
    for (int64_t j = 1; j <= N; ++j) {
        jcolptr[j - 1] = q + 1;
        for (int64_t r = j - 2; r <= j + 2; ++r) if (r >= 1 && r <= N) jrowval[q++] = r;
    }
    jcolptr[N] = q + 1;
    const int64_t jnnz = q, snnz = scolptr[N] - 1;
    int64_t *rows = malloc(sizeof(int64_t) * (size_t)snnz), *cols = malloc(sizeof(int64_t) * (size_t)snnz), *dest = malloc(sizeof(int64_t) * (size_t)snnz);
    for (int64_t j = 1, k = 0; j <= N; ++j)
        for (int64_t p = scolptr[j - 1]; p < scolptr[j]; ++p, ++k) {
            rows[k] = srowval[p - 1]; cols[k] = j;
            int64_t pos = -1;                           /* searchsortedfirst in J's column */
            for (int64_t t = jcolptr[j - 1]; t < jcolptr[j]; ++t) if (jrowval[t - 1] == rows[k]) pos = t - 1;
            dest[k] = pos;
        }
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *nzd = dev_nan((size_t)jnnz);
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_FORWARD;
    CHECK(fd_plan_create_entries(g_ctx, N, N, rows, cols, dest, snnz, jnnz, 8, 1, colors, 8, &o, &plan));
    void *outs[3] = {nzd, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *nz = malloc(sizeof(double) * (size_t)jnnz);
    from_dev(nz, nzd, sizeof(double) * (size_t)jnnz);
    double worst = 0;
    for (int64_t j = 0; j < N; ++j)
        for (int64_t p = jcolptr[j] - 1; p < jcolptr[j + 1] - 1; ++p) {   /* stored entries outside the pattern: 0 */
            const double d = fabs(nz[p] - tridiag_nl_J(x, N, jrowval[p] - 1, j));
            if (!(d <= worst)) worst = d;
        }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(nzd); free(nz); free(x); free(scolptr); free(srowval); free(jcolptr); free(jrowval); free(rows); free(cols); free(dest); free(colors);
    return report("entries", worst, 2e-6, calls, 4);
}

/* shim: make_plan(::Matrix J, ::Nothing) -> fd_plan_create_dense: the `sparsity === nothing` arm with its per-element
   step (src/jacobians.jl:548-557); colorvec = 1:N */
static int client_dense(void)
{
    const int64_t N = 257;
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *Jd = dev_nan((size_t)(N * N));
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_FORWARD;
    CHECK(fd_plan_create_dense(g_ctx, N, N, N /* maximum(colorvec) */, &o, &plan));
    void *outs[3] = {Jd, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *J = malloc(sizeof(double) * (size_t)(N * N)), *eps = malloc(sizeof(double) * (size_t)N);
    from_dev(J, Jd, sizeof(double) * (size_t)(N * N));
    CHECK(fd_plan_get_epsilons(plan, eps));
    double worst = 0, eworst = 0;
    const double rel = sqrt(2.220446049250313e-16);
    for (int64_t c = 0; c < N; ++c) {
        const double want = fmax(rel * fabs(x[c]), rel);       /* compute_epsilon(Val(:forward), x[c], relstep, absstep, dir) */
        eworst = fmax(eworst, fabs(eps[c] - want) / want);
        for (int64_t r = 0; r < N; ++r) {
            const double d = fabs(J[r + N * c] - tridiag_nl_J(x, N, r, c));
            if (!(d <= worst)) worst = d;
        }
    }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(Jd); free(J); free(eps); free(x);
    if (eworst > 1e-15) { printf("dense: per-element step sizes off by %.3e\n", eworst); return 3; }
    return report("dense", worst, 2e-6, calls, N + 1);
}

/* shim: make_plan(::Tridiagonal J) -> fd_plan_create_tridiagonal; outs = (dl, d, du) */
static int client_tridiagonal(void)
{
    const int64_t N = 100001;
    int64_t *colors = cyclic_colors(N, 3);
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N);
    double *dl = dev_nan((size_t)(N - 1)), *d = dev_nan((size_t)N), *du = dev_nan((size_t)(N - 1));
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_CENTRAL;
    CHECK(fd_plan_create_tridiagonal(g_ctx, N, colors, 8, &o, &plan));
    CHECK(install_lazy(plan, fctx));
    void *outs[3] = {dl, d, du};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *h = malloc(sizeof(double) * (size_t)(3 * N));
    from_dev(h, dl, sizeof(double) * (size_t)(N - 1)); from_dev(h + N, d, sizeof(double) * (size_t)N); from_dev(h + 2 * N, du, sizeof(double) * (size_t)(N - 1));
    double worst = 0;
    for (int64_t i = 0; i < N; ++i) {
        double e = fabs(h[N + i] - tridiag_nl_J(x, N, i, i));
        if (i + 1 < N) { e = fmax(e, fabs(h[i] - tridiag_nl_J(x, N, i + 1, i))); e = fmax(e, fabs(h[2 * N + i] - tridiag_nl_J(x, N, i, i + 1))); }
        if (!(e <= worst)) worst = e;
    }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(dl); hipFree(d); hipFree(du); free(h); free(x); free(colors);
    return report("tridiagonal", worst, 2e-8, calls, 6);
}

/* shim (BandedMatrices extension): make_plan(::BandedMatrix J) -> fd_plan_create_banded; outs = (bandeddata(J),)
   (ext/FiniteDiffBandedMatricesExt.jl:13-27) */
static int client_banded(void)
{
    const int64_t N = 60001, l = 1, u = 1, w = l + u + 1;
    int64_t *colors = malloc(sizeof(int64_t) * (size_t)N), nc = 0;
    CHECK(fd_color_banded(N, l, u, colors, &nc));     /* matrix_colors(::BandedMatrix) */
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *dd = dev_nan((size_t)(w * N));
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_FORWARD;
    CHECK(fd_plan_create_banded(g_ctx, N, N, l, u, colors, 8, &o, &plan));
    CHECK(install_lazy(plan, fctx));
    void *outs[3] = {dd, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *data = malloc(sizeof(double) * (size_t)(w * N));
    from_dev(data, dd, sizeof(double) * (size_t)(w * N));
    double worst = 0;
    for (int64_t j = 0; j < N; ++j)
        for (int64_t k = 0; k < w; ++k) {                /* data[u + i - j + 1, j] = J[i, j]; slots outside the matrix: 0 */
            const int64_t i = j - u + k;
            const double want = (i >= 0 && i < N) ? tridiag_nl_J(x, N, i, j) : 0.0;
            const double e = fabs(data[k + w * j] - want);
            if (!(e <= worst)) worst = e;
        }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(dd); free(data); free(x); free(colors);
    return (nc == 3 ? 0 : 3) | report("banded", worst, 2e-6, calls, 4);
}

/* shim (BlockBandedMatrices extension): make_plan(::BlockBandedMatrix J) -> fd_plan_create_blockbanded with
   blocklengths / block_starts / block_strides as BlockSkylineSizes holds them; outs = (J.data,)
   (ext/FiniteDiffBlockBandedMatricesExt.jl:44-68).  f = block-coupled family, complex step. */
static int client_blockbanded(void)
{
    const int64_t nb = 300, bs = 8, N = nb * bs, bl = 1, bu = 1, w = bl + bu + 1;
    int64_t *sizes = malloc(sizeof(int64_t) * (size_t)nb), *starts = calloc((size_t)(w * nb), sizeof(int64_t)), *strides = malloc(sizeof(int64_t) * (size_t)nb);
    int64_t *colors = malloc(sizeof(int64_t) * (size_t)N);
    int64_t off = 1;
    for (int64_t J = 0; J < nb; ++J) {
        sizes[J] = bs;
        const int64_t K0 = J - bu > 0 ? J - bu : 0, K1 = J + bl < nb - 1 ? J + bl : nb - 1;
        strides[J] = (K1 - K0 + 1) * bs;
        int64_t o = off;
        for (int64_t K = K0; K <= K1; ++K) { starts[(bu + K - J) + w * J] = o; o += bs; }
        off += strides[J] * bs;
        for (int64_t j = 0; j < bs; ++j) colors[J * bs + j] = bs * (J % w) + j + 1;
    }
    const int64_t len = off - 1;
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *dd = dev_nan((size_t)len);
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[2] = {nb, bs};
    CHECK(new_f(FD_F_BLOCKCOUPLED, prm, 2, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_COMPLEX;
    CHECK(fd_plan_create_blockbanded(g_ctx, nb, sizes, bl, bu, starts, strides, 8, 1, colors, 8, &o, &plan));
    CHECK(install_lazy(plan, fctx));
    void *outs[3] = {dd, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *data = malloc(sizeof(double) * (size_t)len);
    from_dev(data, dd, sizeof(double) * (size_t)len);
    /* f_b[k] = x_b[k]*(sig_{b-1} + sig_b + sig_{b+1}) + sin(x_b[k]), sig_b = sum_j w_j x_b[j], w_j = (j+1)/bs:
       dF[b,k]/dx[c,j] = x_b[k]*w_j (|b-c| <= 1)  +  (b==c && k==j) * (S_b + cos(x_b[k])) */
    double *sig = calloc((size_t)nb, sizeof(double));
    for (int64_t b = 0; b < nb; ++b) for (int64_t j = 0; j < bs; ++j) sig[b] += (double)(j + 1) / bs * x[b * bs + j];
    double worst = 0;
    for (int64_t J = 0; J < nb; ++J) {
        const int64_t K0 = J - bu > 0 ? J - bu : 0, K1 = J + bl < nb - 1 ? J + bl : nb - 1;
        for (int64_t j = 0; j < bs; ++j)
            for (int64_t K = K0; K <= K1; ++K)
                for (int64_t k = 0; k < bs; ++k) {
                    const double S = (K > 0 ? sig[K - 1] : 0) + sig[K] + (K + 1 < nb ? sig[K + 1] : 0);
                    double want = x[K * bs + k] * (double)(j + 1) / bs;
                    if (K == J && k == j) want += S + cos(x[K * bs + k]);
                    const double got = data[starts[(bu + K - J) + w * J] - 1 + j * strides[J] + k];
                    const double e = fabs(got - want) / fmax(1.0, fabs(want));
                    if (!(e <= worst)) worst = e;
                }
    }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(dd); free(data); free(sig); free(x); free(sizes); free(starts); free(strides); free(colors);
    return report("blockbanded", worst, 1e-12, calls, 3 * bs);
}

/* shim: make_plan(ctx, J::BandedBlockBandedMatrix, ...) (BlockBandedMatrices ext, round 4): the structural plan of
   ext/FiniteDiffBlockBandedMatricesExt.jl:16-42 -- blocklengths, blockbandwidths, subblockbandwidths, and per in-band block the start
   of bandeddata(view(J, K, J)) in J.data with its column stride; no entry list.  The 2-D 5-point Laplacian on an nx x ny grid IS
   such a matrix: ny blocks of nx, block bandwidths (1, 1), sub-block bandwidths (1, 1). */
static int client_bandedblockbanded(int fdtype)
{
    const int64_t nx = 64, ny = 50, nb = ny, N = nx * ny, bl = 1, bu = 1, lam = 1, mu = 1, w = bl + bu + 1, sw = lam + mu + 1, R = w * sw;
    int64_t *sizes = malloc(sizeof(int64_t) * (size_t)nb), *starts = calloc((size_t)(w * nb), sizeof(int64_t)), *strides = malloc(sizeof(int64_t) * (size_t)nb);
    int64_t *colors = malloc(sizeof(int64_t) * (size_t)N);
    for (int64_t J = 0; J < nb; ++J) {
        sizes[J] = nx;
        strides[J] = R;
        for (int64_t K = (J - bu > 0 ? J - bu : 0); K <= (J + bl < nb - 1 ? J + bl : nb - 1); ++K)
            starts[(bu + K - J) + w * J] = 1 + J * nx * R + (bu + K - J) * sw;          /* 1-based, BlockBandedMatrices' layout */
        for (int64_t j = 0; j < nx; ++j) colors[J * nx + j] = sw * (J % w) + (j % sw) + 1;
    }
    const int64_t len = R * N;
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *dd = dev_nan((size_t)len);
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[2] = {nx, ny};
    CHECK(new_f(FD_F_LAP5, prm, 2, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = fdtype;
    CHECK(fd_plan_create_bandedblockbanded(g_ctx, nb, sizes, bl, bu, lam, mu, starts, strides, len, 8, 1, colors, 8, &o, &plan));
    CHECK(install_lazy(plan, fctx));
    void *outs[3] = {dd, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *data = malloc(sizeof(double) * (size_t)len);
    from_dev(data, dd, sizeof(double) * (size_t)len);
    /* the Laplacian's Jacobian: -4 on the diagonal, 1 for the four neighbours; every other slot of every slab 0 (also the slabs
       reserved for blocks outside the matrix) */
    double worst = 0;
    for (int64_t J = 0; J < nb; ++J)
        for (int64_t j = 0; j < nx; ++j)
            for (int64_t d = 0; d < w; ++d)
                for (int64_t t = 0; t < sw; ++t) {
                    const int64_t K = J + d - bu, k = j + t - mu;
                    double want = 0.0;
                    if (K >= 0 && K < nb && k >= 0 && k < nx) {
                        if (K == J) want = k == j ? -4.0 : 1.0;
                        else want = k == j ? 1.0 : 0.0;
                    }
                    const double got = data[J * nx * R + j * R + d * sw + t];
                    const double e = fabs(got - want);
                    if (!(e <= worst)) worst = e;
                }
    /* uniform blocks + a valid colouring: f!, difference, division and the store into the slabs are ONE launch (fd_bbb_store) */
    int64_t stored = 0;
    CHECK(fd_plan_info(plan, FD_INFO_LAZY_STORE, &stored));
    if (stored != 1) { printf("bandedblockbanded: the storing launch did not run\n"); worst = NAN; }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(dd); free(data); free(x); free(sizes); free(starts); free(strides); free(colors);
    return report(fdtype == FD_CENTRAL ? "bbb/central" : "bbb/forward", worst, 1e-6, calls, fdtype == FD_CENTRAL ? 2 * 9 : 1 + 9);
}

/* shim: the Float32 methods generated by the eltype loop: x::ROCVector{Float32} -> the fd32_* symbols */
static int client_csc_f32(void)
{
    const int64_t N = 50001;
    int64_t *colptr, *rowval, *colors = cyclic_colors(N, 3);
    tridiag_csc(N, &colptr, &rowval);
    const int64_t nnz = colptr[N] - 1;
    float *x = malloc(sizeof(float) * (size_t)N);
    for (int64_t j = 0; j < N; ++j) x[j] = 0.5f + 0.25f * sinf((float)(j + 1));
    float *xd = to_dev(x, sizeof(float) * (size_t)N);
    float *nzd = NULL; hipMalloc((void **)&nzd, sizeof(float) * (size_t)nnz); hipMemset(nzd, 0xFF, sizeof(float) * (size_t)nnz);
    fd_f_launch f; void *fctx; fd32_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(fd32_builtin_f_create(g_ctx, FD_F_TRIDIAG, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_CENTRAL;
    CHECK(fd32_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &plan));
    void *outs[3] = {nzd, NULL, NULL};
    CHECK(fd32_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    float *nz = malloc(sizeof(float) * (size_t)nnz);
    from_dev(nz, nzd, sizeof(float) * (size_t)nnz);
    double worst = 0;
    for (int64_t j = 0; j < N; ++j)
        for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
            const double e = fabs((double)nz[p] - (rowval[p] - 1 == j ? -2.0 : 1.0));   /* linear fixture: the exact stencil */
            if (!(e <= worst)) worst = e;
        }
    int64_t l = 0, pts = 0;
    fd32_builtin_f_counts(fctx, &l, &pts);
    CHECK(fd32_plan_destroy(plan)); CHECK(fd32_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(nzd); free(nz); free(x); free(colptr); free(rowval); free(colors);
    return report("csc_f32", worst, 2e-3, pts, 6);
}

/* shim: finite_difference_jvp!(jvp::ROCVector, f::DeviceF, x, v, cache::JVPCache) -> fd_jvp_async (src/jvp.jl:238-274) */
static int client_jvp(void)
{
    const int64_t N = 200000;
    double *x = make_x(N), *v = malloc(sizeof(double) * (size_t)N);
    for (int64_t j = 0; j < N; ++j) v[j] = cos(0.37 * (double)j);
    double *xd = to_dev(x, sizeof(double) * (size_t)N), *vd = to_dev(v, sizeof(double) * (size_t)N), *od = dev_nan((size_t)N);
    fd_f_launch f; void *fctx; fd_jvp_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    CHECK(fd_jvp_plan_create(g_ctx, N, N, FD_CENTRAL, &plan));
    fd_f_launch_lazy_jvp lz = NULL;
    if (fd_builtin_f_lazy_jvp(fctx, &lz) == FD_OK) {
        int caps = 0;
        CHECK(fd_jvp_plan_set_lazy_f(plan, lz));
        fd_builtin_f_lazy_jvp_caps(fctx, &caps);
        CHECK(fd_jvp_plan_set_lazy_caps(plan, caps));
    }
    CHECK(fd_jvp_async(plan, f, fctx, xd, vd, NULL, -1.0, -1.0, 1.0, od));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *out = malloc(sizeof(double) * (size_t)N);
    from_dev(out, od, sizeof(double) * (size_t)N);
    double worst = 0;
    for (int64_t r = 0; r < N; ++r) {
        double want = tridiag_nl_J(x, N, r, r) * v[r];
        if (r > 0) want += tridiag_nl_J(x, N, r, r - 1) * v[r - 1];
        if (r + 1 < N) want += tridiag_nl_J(x, N, r, r + 1) * v[r + 1];
        const double e = fabs(out[r] - want);
        if (!(e <= worst)) worst = e;
    }
    const int64_t calls = f_points(fctx);
    CHECK(fd_jvp_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(vd); hipFree(od); free(out); free(x); free(v);
    return report("jvp", worst, 1e-6, calls, 2);
}

/* shim: TridiagSolver{Float64}(N, :diagonals) + solve!(y, solver, J::Tridiagonal, b): the Jacobian a Tridiagonal plan
   just wrote is consumed where it lies -- one Rosenbrock / implicit-Euler stage (I - gamma*J) y = b */
static int client_solve(void)
{
    const int64_t N = 100001;
    const double gamma = 0.05;
    int64_t *colors = cyclic_colors(N, 3);
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N);
    double *dl = dev_nan((size_t)(N - 1)), *d = dev_nan((size_t)N), *du = dev_nan((size_t)(N - 1));
    double *b = malloc(sizeof(double) * (size_t)N);
    for (int64_t i = 0; i < N; ++i) b[i] = cos(0.01 * (double)i);
    double *bd = to_dev(b, sizeof(double) * (size_t)N), *yd = dev_nan((size_t)N);
    fd_f_launch f; void *fctx; fd_plan *plan; fd_tridiag_solver *solver;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_COMPLEX;
    CHECK(fd_plan_create_tridiagonal(g_ctx, N, colors, 8, &o, &plan));
    CHECK(fd_tridiag_solver_create(g_ctx, N, 0, 0, FD_TRI_DIAGONALS, &solver));
    void *outs[3] = {dl, d, du};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_tridiag_solve_async(solver, 1.0, -gamma, (const void *const *)outs, bd, yd, NULL));   /* same stream: no sync in between */
    CHECK(fd_ctx_synchronize(g_ctx));
    double *y = malloc(sizeof(double) * (size_t)N);
    from_dev(y, yd, sizeof(double) * (size_t)N);
    double worst = 0;                         /* residual with the ANALYTIC Jacobian (complex step: J exact to ~1e-16) */
    for (int64_t i = 0; i < N; ++i) {
        double r = (1.0 - gamma * tridiag_nl_J(x, N, i, i)) * y[i] - b[i];
        if (i > 0) r -= gamma * tridiag_nl_J(x, N, i, i - 1) * y[i - 1];
        if (i + 1 < N) r -= gamma * tridiag_nl_J(x, N, i, i + 1) * y[i + 1];
        if (!(fabs(r) <= worst)) worst = fabs(r);
    }
    const int64_t calls = f_points(fctx);
    CHECK(fd_tridiag_solver_destroy(solver)); CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(dl); hipFree(d); hipFree(du); hipFree(bd); hipFree(yd); free(y); free(b); free(x); free(colors);
    return report("solve", worst, 1e-12, calls, 3);
}

/* shim: BandedSolver -- the consumer for a BandedMatrix jac_prototype: the Jacobian lands in BandedMatrix data, W y = b is solved there */
static int client_bandsolve(void)
{
    const int64_t N = 100001;
    const double gamma = 0.05;
    int64_t *colors = cyclic_colors(N, 3);
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N);
    double *data = dev_nan((size_t)(3 * N));                  /* BandedMatrix data, (l + u + 1) x N column-major, l = u = 1 */
    double *b = malloc(sizeof(double) * (size_t)N);
    for (int64_t i = 0; i < N; ++i) b[i] = cos(0.01 * (double)i);
    double *bd = to_dev(b, sizeof(double) * (size_t)N), *yd = dev_nan((size_t)N);
    fd_f_launch f; void *fctx; fd_plan *plan; fd_banded_solver *solver;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_COMPLEX;
    CHECK(fd_plan_create_banded(g_ctx, N, N, 1, 1, colors, 8, &o, &plan));
    CHECK(fd_banded_solver_create(g_ctx, N, 1, 1, FD_BAND_SOLVE_BANDED, &solver));
    void *outs[3] = {data, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_banded_solve_async(solver, 1.0, -gamma, data, bd, yd));      /* same stream: no sync in between */
    int flags = -1;
    CHECK(fd_banded_solver_status(solver, &flags));
    double *y = malloc(sizeof(double) * (size_t)N);
    from_dev(y, yd, sizeof(double) * (size_t)N);
    double worst = flags == 0 ? 0 : 1;        /* residual with the ANALYTIC Jacobian (complex step: J exact to ~1e-16) */
    for (int64_t i = 0; i < N; ++i) {
        double r = (1.0 - gamma * tridiag_nl_J(x, N, i, i)) * y[i] - b[i];
        if (i > 0) r -= gamma * tridiag_nl_J(x, N, i, i - 1) * y[i - 1];
        if (i + 1 < N) r -= gamma * tridiag_nl_J(x, N, i, i + 1) * y[i + 1];
        if (!(fabs(r) <= worst)) worst = fabs(r);
    }
    const int64_t calls = f_points(fctx);
    CHECK(fd_banded_solver_destroy(solver)); CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(data); hipFree(bd); hipFree(yd); free(y); free(b); free(x); free(colors);
    return report("bandsolve", worst, 1e-12, calls, 3);
}

/* shim: the host-array method (x::Vector{Float64}, J::SparseMatrixCSC on the host) -> fd_jacobian with FD_HOST */
static int client_host(void)
{
    const int64_t N = 4001;
    int64_t *colptr, *rowval, *colors = cyclic_colors(N, 3);
    tridiag_csc(N, &colptr, &rowval);
    const int64_t nnz = colptr[N] - 1;
    double *x = make_x(N), *nz = malloc(sizeof(double) * (size_t)nnz), *fin = malloc(sizeof(double) * (size_t)N);
    for (int64_t i = 0; i < N; ++i) {               /* f_in = f(x), computed by the caller (src/jacobians.jl:540-545) */
        const double xm = i > 0 ? x[i - 1] : 0.0, xp = i + 1 < N ? x[i + 1] : 0.0;
        fin[i] = ((xm - 2 * x[i]) + xp) + (x[i] * x[i]) * xp;
    }
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_FORWARD;
    CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &plan));
    void *outs[3] = {nz, NULL, NULL};
    CHECK(fd_jacobian(plan, f, fctx, x, FD_HOST, fin, FD_HOST, -1.0, -1.0, 1.0, outs, FD_HOST));
    double worst = 0;
    for (int64_t j = 0; j < N; ++j)
        for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
            const double d = fabs(nz[p] - tridiag_nl_J(x, N, rowval[p] - 1, j));
            if (!(d <= worst)) worst = d;
        }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    free(nz); free(fin); free(x); free(colptr); free(rowval); free(colors);
    return report("host+f_in", worst, 2e-6, calls, 3);   /* f_in given: 3 evaluations, not 4 */
}

/* shim: finite_difference_jacobian!(J::SparseMatrixCSC{ComplexF64}, f, x::Vector{ComplexF64}, cache{returntype = ComplexF64}) ->
   PlanOpts(flags = FD_PLAN_COMPLEX_X), fd_plan_create_csc, fd_jacobian_async: complex-valued x with forward / central differences
   (src/jacobians.jl:94-128, 537-622; test/finitedifftests.jl:480-513).  x, nzval are Complex{Float64} arrays as they lie in
   memory ((re, im) pairs); the built-in fixture is evaluated on complex points (is_complex = 1). */
static int client_complex_x(int fdtype)
{
    const int64_t N = 60001;
    int64_t *colptr, *rowval, *colors = cyclic_colors(N, 3);
    tridiag_csc(N, &colptr, &rowval);
    const int64_t nnz = colptr[N] - 1;
    double *x = malloc(sizeof(double) * 2 * (size_t)N);                       /* (re, im) */
    for (int64_t j = 0; j < N; ++j) { x[2 * j] = 0.5 + 0.25 * sin((double)(j + 1)); x[2 * j + 1] = 0.3 * cos(0.7 * (double)(j + 1)); }
    double *xd = to_dev(x, sizeof(double) * 2 * (size_t)N), *nzd = dev_nan(2 * (size_t)nnz);
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = fdtype; o.flags = FD_PLAN_COMPLEX_X;
    CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &plan));
    int64_t len = 0;
    CHECK(fd_plan_info(plan, FD_INFO_OUT0_LEN, &len));
    if (len != 2 * nnz) { printf("complex_x: out length %lld, expected %lld  FAILED\n", (long long)len, (long long)(2 * nnz)); return 3; }
    void *outs[3] = {nzd, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *nz = malloc(sizeof(double) * 2 * (size_t)nnz);
    from_dev(nz, nzd, sizeof(double) * 2 * (size_t)nnz);
    /* analytic Jacobian of f_i = x[i-1] - 2x[i] + x[i+1] + x[i]^2 x[i+1] at complex x (holomorphic: d/dx along the real axis) */
    double worst = 0;
    for (int64_t j = 0; j < N; ++j)
        for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
            const int64_t r = rowval[p] - 1;
            const double xr = x[2 * r], xi = x[2 * r + 1];
            const double pr = r + 1 < N ? x[2 * (r + 1)] : 0.0, pi = r + 1 < N ? x[2 * (r + 1) + 1] : 0.0;
            double wr, wi;
            if (r == j) { wr = -2.0 + 2.0 * (xr * pr - xi * pi); wi = 2.0 * (xr * pi + xi * pr); }
            else if (j == r + 1) { wr = 1.0 + (xr * xr - xi * xi); wi = 2.0 * xr * xi; }
            else { wr = 1.0; wi = 0.0; }
            const double d = fmax(fabs(nz[2 * p] - wr), fabs(nz[2 * p + 1] - wi));
            if (!(d <= worst)) worst = d;
        }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(nzd); free(nz); free(x); free(colptr); free(rowval); free(colors);
    return report(fdtype == FD_FORWARD ? "complex_x/fwd" : "complex_x/cen", worst, fdtype == FD_FORWARD ? 4e-6 : 4e-8, calls, fdtype == FD_FORWARD ? 4 : 6);
}

/* shim: make_plan(ctx, J::Tridiagonal{Complex{T}} | BandedMatrix{Complex{T}}, ...; PlanOpts(complex_x = true)): complex-valued x on
   structured storage (round 4).  The outputs are Complex arrays as they lie in memory: dl / d / du of a Tridiagonal (three arrays),
   the (l + u + 1) x n data of a BandedMatrix (one).  kind 0: Tridiagonal, 1: BandedMatrix(1, 1). */
static int client_complex_structured(int kind, int fdtype)
{
    const int64_t N = 50003;
    int64_t *colors = cyclic_colors(N, 3);
    double *x = malloc(sizeof(double) * 2 * (size_t)N);
    for (int64_t j = 0; j < N; ++j) { x[2 * j] = 0.5 + 0.25 * sin((double)(j + 1)); x[2 * j + 1] = 0.3 * cos(0.7 * (double)(j + 1)); }
    double *xd = to_dev(x, sizeof(double) * 2 * (size_t)N);
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = fdtype; o.flags = FD_PLAN_COMPLEX_X;
    if (kind == 0) CHECK(fd_plan_create_tridiagonal(g_ctx, N, colors, 8, &o, &plan));
    else CHECK(fd_plan_create_banded(g_ctx, N, N, 1, 1, colors, 8, &o, &plan));
    int64_t nouts = 0, len[3] = {0, 0, 0};
    CHECK(fd_plan_info(plan, FD_INFO_NOUTS, &nouts));
    void *outs[3] = {NULL, NULL, NULL};
    for (int k = 0; k < nouts; ++k) { CHECK(fd_plan_info(plan, FD_INFO_OUT0_LEN + k, &len[k])); outs[k] = dev_nan((size_t)len[k]); }
    const int64_t want0 = kind == 0 ? 2 * (N - 1) : 2 * 3 * N;
    if (nouts != (kind == 0 ? 3 : 1) || len[0] != want0) { printf("complex_structured: %lld outputs, first %lld  FAILED\n", (long long)nouts, (long long)len[0]); return 3; }
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *h[3] = {NULL, NULL, NULL};
    for (int k = 0; k < nouts; ++k) { h[k] = malloc(sizeof(double) * (size_t)len[k]); from_dev(h[k], outs[k], sizeof(double) * (size_t)len[k]); }
    /* analytic entries of f_i = x[i-1] - 2x[i] + x[i+1] + x[i]^2 x[i+1]: (r, j) for j = r - 1, r, r + 1 */
    double worst = 0;
    for (int64_t j = 0; j < N; ++j)
        for (int64_t r = j > 0 ? j - 1 : 0; r <= (j + 1 < N ? j + 1 : N - 1); ++r) {
            const double xr = x[2 * r], xi = x[2 * r + 1];
            const double pr = r + 1 < N ? x[2 * (r + 1)] : 0.0, pi = r + 1 < N ? x[2 * (r + 1) + 1] : 0.0;
            double wr, wi;
            if (r == j) { wr = -2.0 + 2.0 * (xr * pr - xi * pi); wi = 2.0 * (xr * pi + xi * pr); }
            else if (j == r + 1) { wr = 1.0 + (xr * xr - xi * xi); wi = 2.0 * xr * xi; }
            else { wr = 1.0; wi = 0.0; }
            const double *v;
            if (kind == 0) v = r == j + 1 ? h[0] + 2 * j : (r == j ? h[1] + 2 * j : h[2] + 2 * (j - 1));     /* dl[j], d[j], du[j-1] */
            else v = h[0] + 2 * ((1 + r - j) + 3 * j);                                                          /* data[u + r - j, j]   */
            const double d = fmax(fabs(v[0] - wr), fabs(v[1] - wi));
            if (!(d <= worst)) worst = d;
        }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    for (int k = 0; k < nouts; ++k) { hipFree(outs[k]); free(h[k]); }
    hipFree(xd); free(x); free(colors);
    char name[64];
    snprintf(name, sizeof name, "complex_%s/%s", kind == 0 ? "tridiagonal" : "banded", fdtype == FD_FORWARD ? "fwd" : "cen");
    return report(name, worst, fdtype == FD_FORWARD ? 4e-6 : 4e-8, calls, fdtype == FD_FORWARD ? 4 : 6);
}

/* shim: FiniteDiff.finite_difference_jacobian(f, x, cache; jac_prototype) (src/jacobians.jl:277-429, the out-of-place method): J is
   allocated like the prototype NEXT TO x -- on the device -- and filled by the in-place call; the caller gets the new J back. */
static int client_out_of_place(void)
{
    const int64_t N = 40001;
    int64_t *colptr, *rowval, *colors = cyclic_colors(N, 3);
    tridiag_csc(N, &colptr, &rowval);
    const int64_t nnz = colptr[N] - 1;
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N);
    fd_f_launch f; void *fctx; fd_plan *plan;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_FORWARD;
    CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &plan));
    double *nzd = NULL;                                         /* similar_J(jac_prototype, x): a fresh nzval of the prototype's length */
    if (hipMalloc((void **)&nzd, sizeof(double) * (size_t)nnz) != 0) return 4;
    void *outs[3] = {nzd, NULL, NULL};
    CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
    CHECK(fd_ctx_synchronize(g_ctx));
    double *nz = malloc(sizeof(double) * (size_t)nnz);
    from_dev(nz, nzd, sizeof(double) * (size_t)nnz);
    double worst = 0;
    for (int64_t j = 0; j < N; ++j)
        for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
            const int64_t r = rowval[p] - 1;
            const double xn = r + 1 < N ? x[r + 1] : 0.0;
            const double w = r == j ? -2.0 + 2.0 * x[r] * xn : (j == r + 1 ? 1.0 + x[r] * x[r] : 1.0);
            const double d = fabs(nz[p] - w);
            if (!(d <= worst)) worst = d;
        }
    const int64_t calls = f_points(fctx);
    CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(nzd); free(nz); free(x); free(colptr); free(rowval); free(colors);
    return report("out_of_place", worst, 4e-6, calls, 4);
}

/* shim: Base.resize!(cache, i) (src/jacobians.jl:655-661) followed by the next finite_difference_jacobian!: the shim's plans are
   keyed on the identity (address, LENGTH) of the arrays plus J's shape and fdtype, so the resized cache (colorvec = 1:i, new lengths) simply
   compiles a new plan and the old one is released -- plan, call, destroy, plan for the new size, call.  Dense arm, as resize!
   sets colorvec = 1:i. */
static int client_resize(void)
{
    int bad = 0;
    const int64_t sizes[2] = {700, 1300};
    for (int s = 0; s < 2; ++s) {
        const int64_t N = sizes[s];
        double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *Jd = dev_nan((size_t)(N * N));
        fd_f_launch f; void *fctx; fd_plan *plan;
        const int64_t prm[1] = {N};
        CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
        fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = FD_FORWARD;
        CHECK(fd_plan_create_dense(g_ctx, N, N, N, &o, &plan));
        void *outs[3] = {Jd, NULL, NULL};
        CHECK(fd_jacobian_async(plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
        CHECK(fd_ctx_synchronize(g_ctx));
        double *J = malloc(sizeof(double) * (size_t)(N * N));
        from_dev(J, Jd, sizeof(double) * (size_t)(N * N));
        double worst = 0;
        for (int64_t c = 0; c < N; ++c)
            for (int64_t r = 0; r < N; ++r) {
                const double d = fabs(J[r + N * c] - tridiag_nl_J(x, N, r, c));
                if (!(d <= worst)) worst = d;
            }
        const int64_t calls = f_points(fctx);
        CHECK(fd_plan_destroy(plan)); CHECK(fd_builtin_f_destroy(fctx));
        hipFree(xd); hipFree(Jd); free(J); free(x);
        bad |= report(s == 0 ? "resize/700" : "resize/1300", worst, 2e-6, calls, N + 1);
    }
    return bad;
}

/* shim: plan_for(cache, J, x, f, sparsity, colorvec, fdtype) + the drop-in method -- the cache -> plan lookup EVERY call goes
   through, then fd_jacobian_async.  The lookup is O(1): the key is the identity (address + length) of colptr / rowval / colorvec,
   J's shape, the fdtype and f; a changed key compiles a new plan.  An in-place edit of colorvec does not change the key: the
   caller says invalidate!(cache) -- or runs with PATTERN_CHECK[] = :content, where plans carry fingerprints
   (FD_PLAN_FINGERPRINT) and every call asks fd_plan_matches (content compared by the library: host threads for host arrays,
   kernels for device arrays).  Both policies are executed here, on a host-resident (Int64) and on a device-resident (Int32)
   pattern, and the per-call cost of the lookup is measured against calling fd_jacobian_async on a plan held in a variable. */
#include <time.h>
static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}
typedef struct shim_cache {       /* PLANS[cache]: the plan and the key it was compiled for */
    fd_plan *plan;
    const void *colptr, *rowval, *colorvec, *fctx;
    int64_t len_colptr, len_rowval, len_colorvec, M, N;
    int fdtype, content, device;
    int plans_built;
} shim_cache;
static int dropin_call(shim_cache *c, int64_t M, int64_t N, const void *colptr, int64_t len_cp, const void *rowval, int64_t len_rv,
                       int idx_bytes, const void *colorvec, int64_t len_cv, int color_bytes, int device_pattern, int fdtype,
                       fd_f_launch f, void *fctx, const void *xd, void *nzd)
{
    int hit = c->plan && c->colptr == colptr && c->rowval == rowval && c->colorvec == colorvec && c->len_colptr == len_cp &&
              c->len_rowval == len_rv && c->len_colorvec == len_cv && c->M == M && c->N == N && c->fdtype == fdtype && c->fctx == fctx &&
              c->device == device_pattern;
    if (hit && c->content) {      /* PATTERN_CHECK[] = :content */
        fd_pattern_arrays pa;
        memset(&pa, 0, sizeof pa);
        pa.idx_a = colptr; pa.len_a = len_cp; pa.idx_b = rowval; pa.len_b = len_rv; pa.colorvec = colorvec; pa.len_color = len_cv;
        pa.idx_bytes = idx_bytes; pa.idx_base = 1; pa.color_bytes = color_bytes; pa.memkind = device_pattern ? FD_DEVICE : FD_HOST;
        int m = 0;
        CHECK(fd_plan_matches(c->plan, &pa, &m));
        hit = m;
    }
    if (!hit) {
        if (c->plan) CHECK(fd_plan_destroy(c->plan));     /* (the shim: the old Plan's finalizer) */
        c->plan = NULL;
        fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = fdtype; o.flags = c->content ? FD_PLAN_FINGERPRINT : 0;
        if (device_pattern) CHECK(fd_plan_create_csc_device(g_ctx, M, N, colptr, rowval, idx_bytes, 1, colorvec, color_bytes, &o, &c->plan));
        else CHECK(fd_plan_create_csc(g_ctx, M, N, colptr, rowval, idx_bytes, 1, colorvec, color_bytes, &o, &c->plan));
        CHECK(install_lazy(c->plan, fctx));
        c->colptr = colptr; c->rowval = rowval; c->colorvec = colorvec; c->fctx = fctx; c->len_colptr = len_cp; c->len_rowval = len_rv;
        c->len_colorvec = len_cv; c->M = M; c->N = N; c->fdtype = fdtype; c->device = device_pattern;
        ++c->plans_built;
    }
    void *outs[3] = {nzd, NULL, NULL};
    return fd_jacobian_async(c->plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs);
}
static void dropin_invalidate(shim_cache *c)    /* invalidate!(cache) */
{
    if (c->plan) fd_plan_destroy(c->plan);
    c->plan = NULL;
}
static int client_dropin(int64_t N, int reps)
{
    int bad = 0;
    int64_t *colptr, *rowval, *colors = cyclic_colors(N, 3);
    tridiag_csc(N, &colptr, &rowval);
    const int64_t nnz = colptr[N] - 1;
    int32_t *cp32 = malloc(sizeof(int32_t) * (size_t)(N + 1)), *rv32 = malloc(sizeof(int32_t) * (size_t)nnz), *cv32 = malloc(sizeof(int32_t) * (size_t)N);
    for (int64_t j = 0; j <= N; ++j) cp32[j] = (int32_t)colptr[j];
    for (int64_t p = 0; p < nnz; ++p) rv32[p] = (int32_t)rowval[p];
    for (int64_t j = 0; j < N; ++j) cv32[j] = (int32_t)colors[j];
    void *cpd = to_dev(cp32, sizeof(int32_t) * (size_t)(N + 1)), *rvd = to_dev(rv32, sizeof(int32_t) * (size_t)nnz), *cvd = to_dev(cv32, sizeof(int32_t) * (size_t)N);
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *nzd = dev_nan((size_t)nnz);
    double *nz = malloc(sizeof(double) * (size_t)nnz);
    fd_f_launch f; void *fctx;
    const int64_t prm[1] = {N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &f, &fctx));
    const int64_t jz = N / 2;     /* the column whose colour the in-place edit removes: its stored values become 0 (fill_matrix!) */
    for (int device = 0; device < 2; ++device)
        for (int content = 0; content < 2; ++content) {
            shim_cache c;
            memset(&c, 0, sizeof c);
            c.content = content;
            const void *cp = device ? cpd : (void *)colptr, *rv = device ? rvd : (void *)rowval, *cv = device ? cvd : (void *)colors;
            const int ib = device ? 4 : 8;
#define DROPIN() dropin_call(&c, N, N, cp, N + 1, rv, nnz, ib, cv, N, ib, device, FD_FORWARD, f, fctx, xd, nzd)
            CHECK(DROPIN());
            CHECK(DROPIN());                                   /* the second call finds the plan */
            CHECK(fd_ctx_synchronize(g_ctx));
            int ok = c.plans_built == 1;
            from_dev(nz, nzd, sizeof(double) * (size_t)nnz);
            double worst = 0;
            for (int64_t j = 0; j < N; ++j)
                for (int64_t p = colptr[j] - 1; p < colptr[j + 1] - 1; ++p) {
                    const double d = fabs(nz[p] - tridiag_nl_J(x, N, rowval[p] - 1, j));
                    if (!(d <= worst)) worst = d;
                }
            ok = ok && worst <= 2e-6;
            /* in-place edit of colorvec: column jz loses its colour */
            if (device) { const int32_t zero = 0; hipMemcpy((char *)cvd + 4 * (size_t)jz, &zero, 4, 1); }
            else colors[jz] = 0;
            CHECK(DROPIN());
            CHECK(fd_ctx_synchronize(g_ctx));
            from_dev(nz, nzd, sizeof(double) * (size_t)nnz);
            const double djz = nz[colptr[jz] - 1 + 1];         /* the diagonal entry of column jz */
            if (content) ok = ok && c.plans_built == 2 && djz == 0.0;      /* the library saw the edit */
            else {
                ok = ok && c.plans_built == 1 && djz != 0.0;               /* identity: the snapshot stands ... */
                dropin_invalidate(&c);                                     /* ... until invalidate!(cache) */
                CHECK(DROPIN());
                CHECK(fd_ctx_synchronize(g_ctx));
                from_dev(nz, nzd, sizeof(double) * (size_t)nnz);
                ok = ok && c.plans_built == 2 && nz[colptr[jz] - 1 + 1] == 0.0;
            }
            /* restore the colouring (another in-place edit), then time the lookup */
            if (device) { const int32_t v = (int32_t)(jz % 3 + 1); hipMemcpy((char *)cvd + 4 * (size_t)jz, &v, 4, 1); }
            else colors[jz] = jz % 3 + 1;
            dropin_invalidate(&c);
            for (int k = 0; k < 3; ++k) CHECK(DROPIN());
            CHECK(fd_ctx_synchronize(g_ctx));
            const int built = c.plans_built;
            double t0 = now_ms();
            for (int k = 0; k < reps; ++k) CHECK(DROPIN());
            CHECK(fd_ctx_synchronize(g_ctx));
            const double ms_dropin = (now_ms() - t0) / reps;
            void *outs[3] = {nzd, NULL, NULL};
            t0 = now_ms();
            for (int k = 0; k < reps; ++k) CHECK(fd_jacobian_async(c.plan, f, fctx, xd, NULL, -1.0, -1.0, 1.0, outs));
            CHECK(fd_ctx_synchronize(g_ctx));
            const double ms_direct = (now_ms() - t0) / reps;
            ok = ok && c.plans_built == built;
            printf("dropin N=%lld pattern=%s check=%s  ms_per_call: dropin %.4f direct %.4f ratio %.3f  plans_built %d  %s\n", (long long)N,
                   device ? "device(Int32)" : "host(Int64)", content ? "content" : "identity", ms_dropin, ms_direct, ms_dropin / ms_direct,
                   c.plans_built, ok ? "ok" : "FAILED");
            bad |= ok ? 0 : 3;
            dropin_invalidate(&c);
#undef DROPIN
        }
    CHECK(fd_builtin_f_destroy(fctx));
    hipFree(xd); hipFree(nzd); hipFree(cpd); hipFree(rvd); hipFree(cvd);
    free(nz); free(x); free(colptr); free(rowval); free(colors); free(cp32); free(rv32); free(cv32);
    return bad;
}

/* A row function handed over as LLVM BITCODE (fd_f_link_rows_bitcode): the file named on the command line was produced OFFLINE by
   `hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fgpu-rdc -emit-llvm --offload-device-only -c tests/bitcode_user_tridiag_nl.hip`
   (what AMDGPU.jl / GPUCompiler emit for a Julia closure takes the same road).  Forward, central and complex step, each as an opaque f!,
   through the column store and -- forward / central -- the band store: the bits of the built-in family; and what the routes cost. */
static int client_bitcode(const char *path, int64_t N, int reps)
{
    FILE *fh = fopen(path, "rb");
    if (!fh) { fprintf(stderr, "cannot open %s\n", path); return 3; }
    fseek(fh, 0, SEEK_END);
    const long nb = ftell(fh);
    fseek(fh, 0, SEEK_SET);
    char *bc = malloc((size_t)nb);
    if (fread(bc, 1, (size_t)nb, fh) != (size_t)nb) { fclose(fh); return 3; }
    fclose(fh);
    int64_t *colptr, *rowval, *colors = cyclic_colors(N, 3);
    tridiag_csc(N, &colptr, &rowval);
    const int64_t nnz = colptr[N] - 1;
    double *x = make_x(N), *xd = to_dev(x, sizeof(double) * (size_t)N), *outd = dev_nan((size_t)nnz), *refd = dev_nan((size_t)nnz);
    double *out = malloc(sizeof(double) * (size_t)nnz), *ref = malloc(sizeof(double) * (size_t)nnz);
    fd_f_launch fb, fu; void *fbctx, *fuctx; fd_f_launch_lazy lz = NULL; int caps = 0;
    const int64_t prm[1] = {N};
    const long long params[1] = {(long long)N};
    CHECK(new_f(FD_F_TRIDIAG_NL, prm, 1, &fb, &fbctx));
    int rc = fd_f_link_rows_bitcode(g_ctx, bc, nb, params, sizeof params, N, N, 8, &fu, &lz, &caps, &fuctx);
    if (rc != FD_OK) { fprintf(stderr, "fd_f_link_rows_bitcode -> %d: %s\n%s\n", rc, fd_last_error(), fd_f_compile_log()); return 3; }
    int bad = 0;
    double t_route[3][3];
    const char *fdn[3] = {"forward", "central", "complex"};
    for (int fdt = 0; fdt < 3; ++fdt) {
        fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = fdt;
        fd_plan *pb, *pl[3];
        CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &pb));
        CHECK(install_lazy(pb, fbctx));
        CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &pl[0]));          /* opaque: no lazy launcher */
        CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &pl[2]));          /* the band store */
        CHECK(fd_plan_set_lazy_f(pl[2], lz));
        CHECK(fd_plan_set_lazy_caps(pl[2], caps));
        o.flags = FD_PLAN_STORE_CSC | FD_PLAN_STORE_CSC_ALWAYS;
        CHECK(fd_plan_create_csc(g_ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &pl[1]));          /* the column store */
        CHECK(fd_plan_set_lazy_f(pl[1], lz));
        CHECK(fd_plan_set_lazy_caps(pl[1], caps & ~FD_LAZY_CAP_STORE));
        void *outr[3] = {refd, NULL, NULL}, *outs[3] = {outd, NULL, NULL};
        CHECK(fd_jacobian_async(pb, fb, fbctx, xd, NULL, -1.0, -1.0, 1.0, outr));
        CHECK(fd_ctx_synchronize(g_ctx));
        from_dev(ref, refd, sizeof(double) * (size_t)nnz);
        for (int k = 0; k < 3; ++k) {
            t_route[fdt][k] = -1;
            if (k == 2 && fdt == 2) continue;                       /* (the complex step of a band goes through the column store) */
            hipMemset(outd, 0xFF, sizeof(double) * (size_t)nnz);
            for (int r = 0; r < 3; ++r) CHECK(fd_jacobian_async(pl[k], fu, fuctx, xd, NULL, -1.0, -1.0, 1.0, outs));
            CHECK(fd_ctx_synchronize(g_ctx));
            from_dev(out, outd, sizeof(double) * (size_t)nnz);
            if (memcmp(out, ref, sizeof(double) * (size_t)nnz) != 0) { printf("bitcode      %s route %d: bits differ from the built-in family  FAILED\n", fdn[fdt], k); bad = 1; }
            const double t0 = now_ms();
            for (int r = 0; r < reps; ++r) CHECK(fd_jacobian_async(pl[k], fu, fuctx, xd, NULL, -1.0, -1.0, 1.0, outs));
            CHECK(fd_ctx_synchronize(g_ctx));
            t_route[fdt][k] = (now_ms() - t0) / reps;
        }
        CHECK(fd_plan_destroy(pb));
        for (int k = 0; k < 3; ++k) CHECK(fd_plan_destroy(pl[k]));
    }
    for (int fdt = 0; fdt < 3; ++fdt)
        printf("bitcode N=%lld %s: linked row function as an opaque f! %.4f ms | column store (one launch) %.4f ms | band store (one launch) %.4f ms\n", (long long)N,
               fdn[fdt], t_route[fdt][0], t_route[fdt][1], t_route[fdt][2]);
    CHECK(fd_builtin_f_destroy(fbctx)); CHECK(fd_f_compiled_destroy(fuctx));
    hipFree(xd); hipFree(outd); hipFree(refd); free(out); free(ref); free(x); free(colptr); free(rowval); free(colors); free(bc);
    if (bad) return 3;
    printf("bitcode      forward / central / complex x opaque / column store / band store: the built-in family's bits  ok\n");
    return 0;
}

int main(int argc, char **argv)
{
    const char *which = argc > 1 ? argv[1] : "all";
    if (hipStreamCreate(&g_stream) != 0) {
        /* no device: say so through the library's own status (FD_ERR_NODEVICE), as the shim would */
        int rc = fd_ctx_create(0, NULL, &g_ctx);
        fprintf(stderr, "fd_ctx_create -> %d: %s\n", rc, fd_last_error());
        return rc == FD_OK ? 1 : rc;
    }
    CHECK(fd_ctx_create(0, g_stream, &g_ctx));      /* Context(device; stream = AMDGPU.stream()) */
    int bad = 0, ran = 0;
#define RUN(name, call) if (!strcmp(which, "all") || !strcmp(which, name)) { bad |= (call); ++ran; }
    RUN("csc", client_csc(FD_FORWARD) | client_csc(FD_CENTRAL) | client_csc(FD_COMPLEX))
    RUN("csc_store_cols", client_csc_store_cols(FD_FORWARD) | client_csc_store_cols(FD_CENTRAL) | client_csc_store_cols(FD_COMPLEX))
    RUN("csc_device", client_csc_device())
    RUN("csc_dense", client_csc_dense())
    RUN("coo_dense", client_coo_dense())
    RUN("entries", client_entries())
    RUN("dense", client_dense())
    RUN("tridiagonal", client_tridiagonal())
    RUN("banded", client_banded())
    RUN("blockbanded", client_blockbanded())
    RUN("bandedblockbanded", client_bandedblockbanded(FD_CENTRAL) | client_bandedblockbanded(FD_FORWARD))
    RUN("csc_f32", client_csc_f32())
    RUN("jvp", client_jvp())
    RUN("solve", client_solve())
    RUN("bandsolve", client_bandsolve())
    RUN("host", client_host())
    RUN("complex_x", client_complex_x(FD_FORWARD) | client_complex_x(FD_CENTRAL))
    RUN("complex_structured", client_complex_structured(0, FD_FORWARD) | client_complex_structured(0, FD_CENTRAL) |
                              client_complex_structured(1, FD_FORWARD) | client_complex_structured(1, FD_CENTRAL))
    RUN("out_of_place", client_out_of_place())
    RUN("resize", client_resize())
    RUN("bitcode", argc > 2 ? client_bitcode(argv[2], argc > 3 ? atoll(argv[3]) : 300007, argc > 4 ? atoi(argv[4]) : 20) : 0)
    RUN("terms", client_terms(argc > 2 ? atoll(argv[2]) : 100003, FD_FORWARD) | client_terms(argc > 2 ? atoll(argv[2]) : 100003, FD_CENTRAL))
    RUN("jit", client_jit(argc > 2 ? atoll(argv[2]) : 300007, argc > 3 ? atoi(argv[3]) : 20))
    RUN("dropin", client_dropin(argc > 2 ? atoll(argv[2]) : 300007, argc > 3 ? atoi(argv[3]) : 20))
    CHECK(fd_ctx_destroy(g_ctx));
    hipStreamDestroy(g_stream);
    if (!ran) { fprintf(stderr, "unknown client %s\n", which); return 2; }
    printf("%s\n", bad ? "SOME CLIENTS FAILED" : "all clients ok");
    return bad ? 3 : 0;
}
