// Does __builtin_amdgcn_global_load_lds (16 B per lane, LDS destination = wave-uniform base + lane*16) do what the
// row-window kernel needs: per-lane global source addresses, linear LDS image, values readable after a barrier?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
__global__ void __launch_bounds__(256) k(const double* __restrict__ src, double* __restrict__ out, int nitems, int npairs, int rot)
{
    extern __shared__ double s[];   // nitems chunks of 128 doubles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int it = wave; it < nitems; it += 4) {
        int i = it * 64 + lane;                 // pair index
        if (i >= npairs) i = npairs - 1;
        const double* g = src + 2 * ((i + rot) % npairs);       // per-lane source (rotated: not linear)
        double* l = s + (size_t)it * 128;                       // wave-uniform LDS base
        __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)l, 16, 0, 0);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * npairs; i += 256) out[i] = s[i] * 2.0;
}
int main() {
    const int npairs = 343, nitems = (npairs + 63) / 64, rot = 7;
    std::vector<double> h(2 * npairs);
    for (int i = 0; i < 2 * npairs; ++i) h[i] = i + 0.25;
    double *d, *o; hipMalloc(&d, h.size() * 8); hipMalloc(&o, h.size() * 8);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), nitems * 128 * 8, 0, d, o, nitems, npairs, rot);
    std::vector<double> r(h.size());
    hipError_t e = hipMemcpy(r.data(), o, h.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < npairs; ++i) {
        const int src = (i + rot) % npairs;
        if (r[2 * i] != 2 * h[2 * src] || r[2 * i + 1] != 2 * h[2 * src + 1]) ++bad;
    }
    printf("glds probe: %s, %d mismatches of %d pairs\n", hipGetErrorString(e), bad, npairs);
    return 0;
}
