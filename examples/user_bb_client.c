/*
 * Plain-C client for examples/user_bb_store.hip: a USER's own HIP f! with a block-tridiagonal Jacobian of dense blocks stores it
 * itself (complex step) through include/fdjac_device.h's fd_colrange_store / fd_colrange_emit into BlockBandedMatrix data; checked
 * against the analytic Jacobian, against the same plan driven WITHOUT the storing launcher (materialised complex points through the
 * user's plain launcher + the library's decompression) and by the number of f! evaluations the reference performs (one per colour:
 * src/jacobians.jl:624-637).
 *
 *   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -Iinclude examples/user_bb_store.hip -o libuser_bb.so
 *   gcc -O2 -Iinclude examples/user_bb_client.c -o user_bb_client -L. -luser_bb -Lfinitediff.jl_amd/lib -lfdjac \
 *       -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD -Wl,-rpath,$PWD/finitediff.jl_amd/lib -Wl,-rpath,/opt/rocm/lib
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

extern int user_bb_init(int64_t nb, int bs);
extern int64_t user_bb_points(void);
extern int user_bb_launch(void *, void *, const void *, int64_t, int64_t, int64_t, int64_t, int64_t, int, void *);
extern int user_bb_launch_lazy(void *, void *, const fd_lazy_points *, int64_t, int64_t, int64_t, void *);

#define CHECK(call)                                                                                \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        if (rc_ != 0) {                                                                            \
            fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, fd_last_error()); \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

int main(int argc, char **argv)
{
    const int64_t nb = argc > 1 ? atoll(argv[1]) : 150, bs = argc > 2 ? atoll(argv[2]) : 12, N = nb * bs, bl = 1, bu = 1, w = 3;
    void *stream = NULL;
    if (hipStreamCreate(&stream) != 0) { fprintf(stderr, "no HIP device\n"); return 7; }
    fd_ctx *ctx = NULL;
    CHECK(fd_ctx_create(0, stream, &ctx));
    /* BlockSkylineSizes-style layout arrays, 1-based starts (as the Julia shim passes them), colours 1 .. 3 bs */
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
    double *x = malloc(sizeof(double) * (size_t)N);
    for (int64_t j = 0; j < N; ++j) x[j] = 0.4 + 0.3 * sin(0.61 * (double)(j + 1));
    void *xd = NULL, *dd = NULL;
    if (hipMalloc(&xd, sizeof(double) * (size_t)N) != 0 || hipMalloc(&dd, sizeof(double) * (size_t)(len + 2)) != 0) return 1;
    hipMemcpy(xd, x, sizeof(double) * (size_t)N, 1);
    fd_plan_opts o;
    memset(&o, 0, sizeof o);
    o.fdtype = FD_COMPLEX;
    fd_plan *plan = NULL;
    CHECK(fd_plan_create_blockbanded(ctx, nb, sizes, bl, bu, starts, strides, 8, 1, colors, 8, &o, &plan));
    if (user_bb_init(nb, (int)bs) != 0) return 1;
    double *res[2];
    int bad = 0;
    for (int run = 0; run < 2; ++run) {            /* run 0: the user's storing launcher; run 1: materialised points + decompression */
        user_bb_init(nb, (int)bs);
        CHECK(fd_plan_set_lazy_f(plan, run == 0 ? user_bb_launch_lazy : NULL));
        if (run == 0) CHECK(fd_plan_set_lazy_caps(plan, FD_LAZY_CAP_STORE));
        int64_t v = -1;
        CHECK(fd_plan_info(plan, FD_INFO_LAZY_STORE, &v));
        if (v != (run == 0 ? 1 : 0)) { printf("FD_INFO_LAZY_STORE = %lld in run %d  FAILED\n", (long long)v, run); bad = 1; }
        double *h = malloc(sizeof(double) * (size_t)(len + 2));
        for (int64_t i = 0; i < len + 2; ++i) h[i] = NAN;
        hipMemcpy(dd, h, sizeof(double) * (size_t)(len + 2), 1);
        void *outs[3] = {dd, NULL, NULL};
        CHECK(fd_jacobian_async(plan, user_bb_launch, NULL, xd, NULL, -1.0, -1.0, 1.0, outs));
        hipStreamSynchronize(stream);
        hipMemcpy(h, dd, sizeof(double) * (size_t)(len + 2), 2);
        if (!isnan(h[len]) || !isnan(h[len + 1])) { printf("wrote past the end in run %d  FAILED\n", run); bad = 1; }
        const int64_t ncol = (nb < 3 ? nb : 3) * bs;        /* colours in use: one f! evaluation each */
        if (user_bb_points() != ncol) { printf("run %d: %lld f! evaluations, expected %lld  FAILED\n", run, (long long)user_bb_points(), (long long)ncol); bad = 1; }
        res[run] = h;
    }
    /* dF_k/dx_j = x_k w_j (blocks |b(k) - b(j)| <= 1)  +  [k == j] (S_{b(k)} + cos x_k),  S_b = sum of w_j x_j over blocks b-1 .. b+1 */
    double *S = calloc((size_t)nb, sizeof(double));
    for (int64_t b = 0; b < nb; ++b)
        for (int64_t j = (b > 0 ? b - 1 : 0) * bs; j < (b + 1 < nb ? b + 2 : nb) * bs; ++j) S[b] += x[j] / (1.0 + (double)(j % bs));
    double worst = 0, worst_ab = 0;
    for (int64_t J = 0; J < nb; ++J) {
        const int64_t K0 = J - bu > 0 ? J - bu : 0, K1 = J + bl < nb - 1 ? J + bl : nb - 1;
        for (int64_t j = 0; j < bs; ++j)
            for (int64_t K = K0; K <= K1; ++K)
                for (int64_t k = 0; k < bs; ++k) {
                    const int64_t col = J * bs + j, row = K * bs + k;
                    double want = x[row] / (1.0 + (double)(col % bs));
                    if (row == col) want += S[K] + cos(x[row]);
                    const int64_t at = starts[(bu + K - J) + w * J] - 1 + j * strides[J] + k;
                    const double e = fabs(res[0][at] - want) / fmax(1.0, fabs(want)), d = fabs(res[0][at] - res[1][at]) / fmax(1.0, fabs(want));
                    if (!(e <= worst)) worst = e;
                    if (!(d <= worst_ab)) worst_ab = d;
                }
    }
    const int ok = !bad && worst <= 1e-12 && worst_ab <= 1e-12;
    printf("user kernel -> blockbanded complex step: max rel |J - analytic| = %.3e, storing launch vs decompression %.3e (tol 1e-12)  %s\n",
           worst, worst_ab, ok ? "ok" : "FAILED");
    CHECK(fd_plan_destroy(plan));
    hipFree(xd); hipFree(dd);
    CHECK(fd_ctx_destroy(ctx));
    printf(ok ? "user_bb_client ok\n" : "user_bb_client FAILED\n");
    return ok ? 0 : 3;
}
