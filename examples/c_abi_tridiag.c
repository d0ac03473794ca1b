/*
 * A plain-C client of libfdjac (include/fdjac.h): the call sequence a Julia `ccall` shim performs for
 *     FiniteDiff.finite_difference_jacobian!(J::SparseMatrixCSC, f!, x; colorvec = repeat(1:3, N/3))
 * with Julia's own arrays (1-based Int64 colptr / rowval, host memory) -- no HIP, no Python on the client side.
 *
 *   gcc -O2 -Iinclude examples/c_abi_tridiag.c -o c_abi_tridiag -Lfinitediff.jl_amd/lib -lfdjac \
 *       -Wl,-rpath,$PWD/finitediff.jl_amd/lib
 *   ./c_abi_tridiag 30        # prints the max deviation from the exact second-difference stencil (-2 / 1)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "fdjac.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != FD_OK) {                                                            \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, fd_last_error());            \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 30;
    if (N < 2) return 2;
    const int64_t nnz = 3 * N - 2;
    int64_t *colptr = malloc(sizeof(int64_t) * (size_t)(N + 1));
    int64_t *rowval = malloc(sizeof(int64_t) * (size_t)nnz);
    int64_t *colorvec = malloc(sizeof(int64_t) * (size_t)N);
    double *x = malloc(sizeof(double) * (size_t)N), *nzval = malloc(sizeof(double) * (size_t)nnz);
    /* sparse(Tridiagonal(...)) pattern, 1-based as Julia stores it; colorvec = mod1(j, 3) */
    int64_t p = 0;
    for (int64_t j = 1; j <= N; ++j) {
        colptr[j - 1] = p + 1;
        for (int64_t r = j - 1; r <= j + 1; ++r)
            if (r >= 1 && r <= N) rowval[p++] = r;
        colorvec[j - 1] = (j - 1) % 3 + 1;
        x[j - 1] = 0.5 + 0.25 * sin((double)j);
    }
    colptr[N] = p + 1;

    fd_ctx *ctx = NULL;
    fd_plan *plan = NULL;
    fd_f_launch f = NULL;
    void *fctx = NULL;
    CHECK(fd_ctx_create(0, NULL, &ctx));
    const int64_t prm[1] = {N};
    CHECK(fd_builtin_f_create(ctx, FD_F_TRIDIAG, prm, 1, &f, &fctx));   /* f!(dx, x): test/coloring_tests.jl:5-13 */
    fd_plan_opts opts = {0};
    opts.fdtype = FD_FORWARD;
    CHECK(fd_plan_create_csc(ctx, N, N, colptr, rowval, 8, 1, colorvec, 8, &opts, &plan));
    void *outs[3] = {nzval, NULL, NULL};
    CHECK(fd_jacobian(plan, f, fctx, x, FD_HOST, NULL, FD_HOST, -1.0, -1.0, 1.0, outs, FD_HOST));

    int64_t launches = 0, points = 0;
    CHECK(fd_builtin_f_counts(fctx, &launches, &points));
    double worst = 0.0;
    p = 0;
    for (int64_t j = 1; j <= N; ++j)
        for (int64_t r = j - 1; r <= j + 1; ++r)
            if (r >= 1 && r <= N) {
                const double exact = r == j ? -2.0 : 1.0, d = fabs(nzval[p++] - exact);
                if (d > worst) worst = d;
            }
    printf("N=%lld nnz=%lld f!_evaluations=%lld max|J - exact|=%.3e\n", (long long)N, (long long)nnz, (long long)points, worst);
    CHECK(fd_plan_destroy(plan));
    CHECK(fd_builtin_f_destroy(fctx));
    CHECK(fd_ctx_destroy(ctx));
    free(colptr); free(rowval); free(colorvec); free(x); free(nzval);
    return worst < 1e-6 ? 0 : 3;
}
