// how long does one dependent v_fma_f32 take (ns), as a function of how busy the chip is?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float *out, float a, float b, int n, unsigned long long *t) {
    float x = threadIdx.x * 1e-9f;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) x = __builtin_fmaf(x, a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
int main() {
    float *out; unsigned long long *t;
    hipMalloc(&out, 4096 * 1024 * 4); hipMalloc(&t, 4096 * 8);
    unsigned long long h[4096];
    for (int wgs : {1, 12, 256, 1024, 4096}) for (int threads : {64, 256, 1024}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(chain, dim3(wgs), dim3(threads), 0, 0, out, 0.999f, 0.001f, 200, t);
        hipDeviceSynchronize();
        hipMemcpy(h, t, wgs * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < wgs; ++i) avg += h[i]; avg /= wgs;
        printf("wgs %4d threads %4d: %.2f ns per dependent fma (wave 0 of each WG, 12800 fmas)\n", wgs, threads, avg * 10.0 / (200 * 64));
    }
    return 0;
}
