/*
 * Drives examples/user_terms_store.hip through the C ABI: a user's SEPARABLE residual on the nine-point (Moore) stencil whose launch
 * stores the Jacobian ROW BY ROW through the plan's row lists (FD_PLAN_STORE_CSC | FD_PLAN_STORE_CSC_ROWS, fd_plan_row_lists,
 * fd_csc_store_rows).  Checks: the lists exist; the bound plan takes the row-wise launch, another plan of the same pattern the column
 * store; both are bit-identical to the user's PLAIN launcher + the library's decompression; the analytic Jacobian; the f! counts.
 *
 *   gcc -O2 -Iinclude examples/user_terms_client.c -o user_terms_client -L. -luser_tm -Lfinitediff.jl_amd/lib -lfdjac \
 *       -L/opt/rocm/lib -lamdhip64 -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fdjac.h"

extern int hipMalloc(void **ptr, size_t size);
extern int hipFree(void *ptr);
extern int hipMemcpy(void *dst, const void *src, size_t size, int kind);
extern int hipStreamCreate(void **stream);
extern int hipStreamDestroy(void *stream);

extern void user_tm_bind(int64_t M, const void *row_ptr_dev, const void *row_col_dev, int64_t entries, uint64_t plan_serial);
extern int64_t user_tm_points(void);
extern int64_t user_tm_row_stores(void);
extern int user_tm_launch(void *, void *, const void *, int64_t, int64_t, int64_t, int64_t, int64_t, int, void *);
extern int user_tm_launch_lazy(void *, void *, const fd_lazy_points *, int64_t, int64_t, int64_t, void *);

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, fd_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int64_t nx = argc > 1 ? atoll(argv[1]) : 61, ny = argc > 2 ? atoll(argv[2]) : 47, N = nx * ny;
    void *stream = NULL;
    fd_ctx *ctx = NULL;
    if (hipStreamCreate(&stream) != 0) { fprintf(stderr, "no device\n"); return 7; }
    CHECK(fd_ctx_create(0, stream, &ctx));
    /* the CSC pattern (1-based Int64, as Julia holds it), rows ascending within a column; colours (i mod 3) + 3 (j mod 3) + 1 */
    int64_t *colptr = malloc(sizeof(int64_t) * (size_t)(N + 1)), *rowval = malloc(sizeof(int64_t) * (size_t)(9 * N)), *colors = malloc(sizeof(int64_t) * (size_t)N);
    int64_t p = 0;
    for (int64_t k = 0; k < N; ++k) {
        const int64_t j = k / nx, i = k % nx;
        colptr[k] = p + 1;
        colors[k] = (i % 3) + 3 * (j % 3) + 1;
        for (int dj = -1; dj <= 1; ++dj)
            for (int di = -1; di <= 1; ++di)
                if (i + di >= 0 && i + di < nx && j + dj >= 0 && j + dj < ny) rowval[p++] = (j + dj) * nx + (i + di) + 1;
    }
    colptr[N] = p + 1;
    const int64_t nnz = p;
    double *x = malloc(sizeof(double) * (size_t)N);
    for (int64_t k = 0; k < N; ++k) x[k] = 0.5 + 0.25 * sin((double)(k + 1));
    void *xd = NULL, *o1 = NULL, *o2 = NULL, *o3 = NULL;
    hipMalloc(&xd, sizeof(double) * (size_t)N); hipMalloc(&o1, sizeof(double) * (size_t)nnz); hipMalloc(&o2, sizeof(double) * (size_t)nnz); hipMalloc(&o3, sizeof(double) * (size_t)nnz);
    hipMemcpy(xd, x, sizeof(double) * (size_t)N, 1);
    double *a = malloc(sizeof(double) * (size_t)nnz), *b = malloc(sizeof(double) * (size_t)nnz), *c = malloc(sizeof(double) * (size_t)nnz);
    int bad = 0;
    for (int fdtype = FD_FORWARD; fdtype <= FD_CENTRAL; ++fdtype) {
        fd_plan_opts o; memset(&o, 0, sizeof o); o.fdtype = fdtype; o.flags = FD_PLAN_STORE_CSC | FD_PLAN_STORE_CSC_ROWS;
        fd_plan *pr = NULL, *pc = NULL, *ph = NULL;
        CHECK(fd_plan_create_csc(ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &pr));       /* the plan the functor is bound to: row by row */
        const void *row_ptr = NULL, *row_col = NULL, *row_slot = NULL; int64_t entries = 0; uint64_t serial = 0;
        CHECK(fd_plan_row_lists(pr, &row_ptr, &row_col, &row_slot, &entries, &serial));
        user_tm_bind(N, row_ptr, row_col, entries, serial);
        CHECK(fd_plan_set_lazy_f(pr, user_tm_launch_lazy));
        CHECK(fd_plan_set_lazy_caps(pr, FD_LAZY_CAP_STORE_CSC | FD_LAZY_CAP_STORE_CSC_BASE));
        o.flags = FD_PLAN_STORE_CSC;
        CHECK(fd_plan_create_csc(ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &pc));       /* another plan of the same pattern: column by column */
        CHECK(fd_plan_set_lazy_f(pc, user_tm_launch_lazy));
        CHECK(fd_plan_set_lazy_caps(pc, FD_LAZY_CAP_STORE_CSC | FD_LAZY_CAP_STORE_CSC_BASE));
        o.flags = 0;
        CHECK(fd_plan_create_csc(ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &ph));       /* plain launcher + the library's decompression */
        void *outs1[3] = {o1, NULL, NULL}, *outs2[3] = {o2, NULL, NULL}, *outs3[3] = {o3, NULL, NULL};
        const int64_t n0 = user_tm_points(), r0 = user_tm_row_stores();
        CHECK(fd_jacobian_async(pr, user_tm_launch, NULL, xd, NULL, -1.0, -1.0, 1.0, outs1));
        CHECK(fd_ctx_synchronize(ctx));
        const int64_t n1 = user_tm_points(), r1 = user_tm_row_stores();
        CHECK(fd_jacobian_async(pc, user_tm_launch, NULL, xd, NULL, -1.0, -1.0, 1.0, outs2));
        CHECK(fd_ctx_synchronize(ctx));
        const int64_t n2 = user_tm_points(), r2 = user_tm_row_stores();
        CHECK(fd_jacobian_async(ph, user_tm_launch, NULL, xd, NULL, -1.0, -1.0, 1.0, outs3));
        CHECK(fd_ctx_synchronize(ctx));
        const int64_t n3 = user_tm_points();
        hipMemcpy(a, o1, sizeof(double) * (size_t)nnz, 2); hipMemcpy(b, o2, sizeof(double) * (size_t)nnz, 2); hipMemcpy(c, o3, sizeof(double) * (size_t)nnz, 2);
        double worst = 0;
        for (int64_t k = 0; k < N; ++k)
            for (int64_t q = colptr[k] - 1; q < colptr[k + 1] - 1; ++q) {
                const double want = rowval[q] - 1 == k ? -4.0 + 3.0 * x[k] * x[k] : 0.5;
                const double d = fabs(a[q] - want);
                if (!(d <= worst)) worst = d;
            }
        const int same = memcmp(a, c, sizeof(double) * (size_t)nnz) == 0 && memcmp(b, c, sizeof(double) * (size_t)nnz) == 0;
        const int64_t want_calls = fdtype == FD_FORWARD ? 10 : 18;
        /* (a grid narrower than three columns has rows that reach further than the row-wise kernel's window allows for: nx ny < 2 is declined) */
        const int rows_expected = N >= 2;
        const int ok = entries == nnz && (r1 - r0) == rows_expected && (r2 - r1) == 0 && same && worst <= (fdtype == FD_FORWARD ? 2e-6 : 2e-8) && n1 - n0 == want_calls &&
                       n2 - n1 == want_calls && n3 - n2 == want_calls;
        printf("user_terms %s: row lists %lld of %lld entries, row-wise launches %lld (bound plan) / %lld (another plan), bit-identical to the hand-over path %s, "
               "max|J - analytic| %.3e, f! evaluations %lld / %lld / %lld (expected %lld)  %s\n", fdtype == FD_FORWARD ? "forward" : "central", (long long)entries,
               (long long)nnz, (long long)(r1 - r0), (long long)(r2 - r1), same ? "yes" : "NO", worst, (long long)(n1 - n0), (long long)(n2 - n1), (long long)(n3 - n2),
               (long long)want_calls, ok ? "ok" : "FAILED");
        bad |= !ok;
        CHECK(fd_plan_destroy(pr)); CHECK(fd_plan_destroy(pc)); CHECK(fd_plan_destroy(ph));
    }
    hipFree(xd); hipFree(o1); hipFree(o2); hipFree(o3);
    free(a); free(b); free(c); free(x); free(colptr); free(rowval); free(colors);
    CHECK(fd_ctx_destroy(ctx));
    hipStreamDestroy(stream);
    printf("%s\n", bad ? "user_terms_client FAILED" : "user_terms_client ok");
    return bad ? 3 : 0;
}
