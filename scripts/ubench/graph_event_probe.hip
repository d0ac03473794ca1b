// Do events recorded inside a captured HIP graph give usable hipEventElapsedTime values on replay?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); } } while (0)
__global__ void spin(double* p, int n) { double a = p[threadIdx.x]; for (int i = 0; i < n; ++i) a = a * 1.0000001 + 1e-9; p[threadIdx.x] = a; }
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double* p; CK(hipMalloc(&p, 8 * 256)); CK(hipMemset(p, 0, 8 * 256));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(e0, s));
    spin<<<1, 256, 0, s>>>(p, 200000);
    CK(hipEventRecord(e1, s));
    spin<<<1, 256, 0, s>>>(p, 600000);
    CK(hipEventRecord(e2, s));
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < 3; ++it) {
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        float a = -1, b = -1;
        hipError_t r1 = hipEventElapsedTime(&a, e0, e1), r2 = hipEventElapsedTime(&b, e1, e2);
        printf("replay %d: e0->e1 %.3f ms (%s), e1->e2 %.3f ms (%s)\n", it, a, hipGetErrorString(r1), b, hipGetErrorString(r2));
    }
    // reference: direct launches
    CK(hipEventRecord(e0, s)); spin<<<1, 256, 0, s>>>(p, 200000); CK(hipEventRecord(e1, s)); spin<<<1, 256, 0, s>>>(p, 600000); CK(hipEventRecord(e2, s));
    CK(hipStreamSynchronize(s));
    float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
    printf("direct   : e0->e1 %.3f ms, e1->e2 %.3f ms\n", a, b);
    // launch overhead: 4 tiny kernels, direct vs graph, 200 repetitions
    hipGraph_t g2; hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < 4; ++k) spin<<<1, 256, 0, s>>>(p, 10);
    CK(hipStreamEndCapture(s, &g2)); CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge2, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); for (int r = 0; r < 200; ++r) CK(hipGraphLaunch(ge2, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&a, e0, e1));
    CK(hipEventRecord(e0, s)); for (int r = 0; r < 200; ++r) for (int k = 0; k < 4; ++k) spin<<<1, 256, 0, s>>>(p, 10); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&b, e0, e1));
    printf("4 tiny kernels per step: graph %.2f us/step, direct %.2f us/step\n", a * 1e3 / 200, b * 1e3 / 200);
    return 0;
}
