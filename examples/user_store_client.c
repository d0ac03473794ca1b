/*
 * Plain-C client for examples/user_f_store.hip: a USER's own HIP f! (its own shared library, built with hipcc apart from
 * libfdjac) stores a tridiagonal Jacobian itself through include/fdjac_device.h -- into SparseMatrixCSC nzval, BandedMatrix
 * data and Tridiagonal dl / d / du -- and the result is checked against the analytic Jacobian, against the same plan driven
 * WITHOUT the storing launcher (materialised points + the library's decompression: must be the same bits) and by the number
 * of f! evaluations the reference performs (1 + C forward, 2C central; test/coloring_tests.jl:36,42).
 *
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -Iinclude examples/user_f_store.hip -o libuser_f.so
 *   gcc -O2 -Iinclude examples/user_store_client.c -o user_store_client -L. -luser_f -Lfinitediff.jl_amd/lib -lfdjac \
 *       -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD -Wl,-rpath,$PWD/finitediff.jl_amd/lib -Wl,-rpath,/opt/rocm/lib
 *   ./user_store_client [N] [dump.bin]      (dump.bin: x and the CSC nzval of the forward run, for the oracle comparison)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fdjac.h"

extern int hipMalloc(void **ptr, size_t size);
extern int hipFree(void *ptr);
extern int hipMemcpy(void *dst, const void *src, size_t size, int kind); /* 1 = host->device, 2 = device->host */
extern int hipStreamCreate(void **stream);
extern int hipStreamSynchronize(void *stream);

/* the user's library (examples/user_f_store.hip) */
extern void user_f_init(int64_t n, int mode);
extern void user_f_set_mode(int mode);
extern int64_t user_f_points(void);
extern int user_f_launch(void *, void *, const void *, int64_t, int64_t, int64_t, int64_t, int64_t, int, void *);
extern int user_f_launch_lazy(void *, void *, const fd_lazy_points *, int64_t, int64_t, int64_t, void *);

#define CHECK(call)                                                                                \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        if (rc_ != 0) {                                                                            \
            fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, fd_last_error()); \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

static double J_exact(const double *x, int64_t N, int64_t r, int64_t c)
{
    const double xm = r > 0 ? x[r - 1] : 0.0, xp = r + 1 < N ? x[r + 1] : 0.0;
    if (c == r - 1) return 1.0 - 0.25 * x[r];
    if (c == r) return -2.0 + 0.25 * (xp - xm);
    if (c == r + 1) return 1.0 + 0.25 * x[r];
    return 0.0;
}

enum { ST_CSC, ST_BANDED, ST_TRIDIAGONAL };

/* one storage type, one fdtype: plan, three runs (user kernel row-centric / wave, no storing launcher), checks */
static int run_case(fd_ctx *ctx, void *stream, int storage, int fdtype, int64_t N, const double *x, const double *x_dev,
                    const int64_t *colors, const char *dump)
{
    fd_plan_opts o;
    memset(&o, 0, sizeof o);
    o.fdtype = fdtype;
    fd_plan *plan = NULL;
    int64_t len[3] = {0, 0, 0};
    int nouts = 1;
    if (storage == ST_CSC) {
        int64_t *colptr = malloc(sizeof(int64_t) * (size_t)(N + 1)), *rowval = malloc(sizeof(int64_t) * (size_t)(3 * N));
        int64_t p = 0;
        for (int64_t j = 1; j <= N; ++j) {
            colptr[j - 1] = p + 1;
            for (int64_t r = j - 1; r <= j + 1; ++r) if (r >= 1 && r <= N) rowval[p++] = r;
        }
        colptr[N] = p + 1;
        CHECK(fd_plan_create_csc(ctx, N, N, colptr, rowval, 8, 1, colors, 8, &o, &plan));
        free(colptr); free(rowval);
    } else if (storage == ST_BANDED) {
        CHECK(fd_plan_create_banded(ctx, N, N, 1, 1, colors, 8, &o, &plan));
    } else {
        CHECK(fd_plan_create_tridiagonal(ctx, N, colors, 8, &o, &plan));
    }
    int64_t v = 0;
    CHECK(fd_plan_info(plan, FD_INFO_NOUTS, &v)); nouts = (int)v;
    for (int k = 0; k < nouts; ++k) CHECK(fd_plan_info(plan, FD_INFO_OUT0_LEN + k, &len[k]));
    double *res[3][3];      /* [run][output] host copies */
    void *outs[3] = {NULL, NULL, NULL};
    for (int k = 0; k < nouts; ++k) if (hipMalloc(&outs[k], sizeof(double) * (size_t)(len[k] + 2)) != 0) return 1;
    const int64_t want_calls = fdtype == FD_FORWARD ? 4 : 6;
    int bad = 0;
    for (int run = 0; run < 3; ++run) {
        /* run 0: row-centric user kernel (fd_band_emit); run 1: wave kernel (fd_band_emit_wave); run 2: no storing launcher */
        user_f_init(N, run == 1 ? 1 : 0);
        CHECK(fd_plan_set_lazy_f(plan, run < 2 ? user_f_launch_lazy : NULL));
        if (run < 2) CHECK(fd_plan_set_lazy_caps(plan, FD_LAZY_CAP_STORE));
        CHECK(fd_plan_info(plan, FD_INFO_LAZY_STORE, &v));
        if (v != (run < 2 ? 1 : 0)) { printf("FD_INFO_LAZY_STORE = %lld in run %d  FAILED\n", (long long)v, run); bad = 1; }
        for (int k = 0; k < nouts; ++k) {     /* NaN-fill: a value nobody wrote cannot pass */
            double *h = malloc(sizeof(double) * (size_t)(len[k] + 2));
            for (int64_t i = 0; i < len[k] + 2; ++i) h[i] = NAN;
            hipMemcpy(outs[k], h, sizeof(double) * (size_t)(len[k] + 2), 1);
            free(h);
        }
        CHECK(fd_jacobian_async(plan, user_f_launch, NULL, x_dev, NULL, -1.0, -1.0, 1.0, outs));
        hipStreamSynchronize(stream);
        for (int k = 0; k < nouts; ++k) {
            res[run][k] = malloc(sizeof(double) * (size_t)(len[k] + 2));
            hipMemcpy(res[run][k], outs[k], sizeof(double) * (size_t)(len[k] + 2), 2);
            if (!isnan(res[run][k][len[k]]) || !isnan(res[run][k][len[k] + 1])) { printf("wrote past the end of output %d  FAILED\n", k); bad = 1; }
        }
        if (user_f_points() != want_calls) { printf("run %d: %lld f! evaluations, expected %lld  FAILED\n", run, (long long)user_f_points(), (long long)want_calls); bad = 1; }
    }
    /* every stored value against the analytic Jacobian; the three runs bit for bit */
    const double tol = fdtype == FD_FORWARD ? 2e-6 : 2e-9;
    double worst = 0.0;
    int64_t diffs = 0;
    for (int k = 0; k < nouts; ++k)
        for (int64_t i = 0; i < len[k]; ++i) {
            if (memcmp(&res[0][k][i], &res[2][k][i], 8) != 0 || memcmp(&res[1][k][i], &res[2][k][i], 8) != 0) ++diffs;
            int64_t r, c;
            int slot_outside = 0;
            if (storage == ST_CSC) { c = (i + 1) / 3; r = c - 1 + ((i + 1) - 3 * c); }
            else if (storage == ST_BANDED) { c = i / 3; r = c - 1 + (i - 3 * c); slot_outside = r < 0 || r >= N; }
            else { c = k == 2 ? i + 1 : i; r = k == 0 ? i + 1 : (k == 1 ? i : i); }
            const double want = slot_outside ? 0.0 : J_exact(x, N, r, c);
            const double e = fabs(res[1][k][i] - want);
            if (!(e <= worst)) worst = e;     /* (NaN-propagating) */
        }
    const char *names[3] = {"csc", "banded", "tridiagonal"};
    const int ok = !bad && diffs == 0 && worst <= tol;
    printf("user kernel -> %-11s %-7s max|J - analytic| = %.3e (tol %.0e), %lld bitwise differences between the row-centric, wave and "
           "library-decompressed runs  %s\n", names[storage], fdtype == FD_FORWARD ? "forward" : "central", worst, tol, (long long)diffs,
           ok ? "ok" : "FAILED");
    if (dump && storage == ST_CSC && fdtype == FD_FORWARD) {
        FILE *fh = fopen(dump, "wb");
        if (fh) { fwrite(&N, 8, 1, fh); fwrite(x, 8, (size_t)N, fh); fwrite(res[1][0], 8, (size_t)len[0], fh); fclose(fh); }
    }
    for (int run = 0; run < 3; ++run) for (int k = 0; k < nouts; ++k) free(res[run][k]);
    for (int k = 0; k < nouts; ++k) hipFree(outs[k]);
    CHECK(fd_plan_destroy(plan));
    return ok ? 0 : 3;
}

int main(int argc, char **argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : 200003;
    const char *dump = argc > 2 ? argv[2] : NULL;
    void *stream = NULL;
    if (hipStreamCreate(&stream) != 0) { fprintf(stderr, "no HIP device\n"); return 7; }
    fd_ctx *ctx = NULL;
    CHECK(fd_ctx_create(0, stream, &ctx));
    double *x = malloc(sizeof(double) * (size_t)N);
    int64_t *colors = malloc(sizeof(int64_t) * (size_t)N);
    for (int64_t j = 0; j < N; ++j) { x[j] = 0.5 + 0.25 * sin(0.37 * (double)(j + 1)); colors[j] = j % 3 + 1; }
    void *x_dev = NULL;
    if (hipMalloc(&x_dev, sizeof(double) * (size_t)N) != 0) return 1;
    hipMemcpy(x_dev, x, sizeof(double) * (size_t)N, 1);
    int rc = 0;
    for (int storage = ST_CSC; storage <= ST_TRIDIAGONAL; ++storage)
        for (int fdtype = FD_FORWARD; fdtype <= FD_CENTRAL; ++fdtype)
            rc |= run_case(ctx, stream, storage, fdtype, N, x, (const double *)x_dev, colors, dump);
    hipFree(x_dev);
    CHECK(fd_ctx_destroy(ctx));
    printf(rc ? "user_store_client FAILED\n" : "user_store_client ok\n");
    return rc;
}
