// How fast does ONE wave walk a dependent fp32 add chain whose operands come from LDS (the LayerNorm statistics walk, the logsum lane)?
// Variants: (a) the C++ ping-pong loop of ln_stats_resident_kernel, 16 active lanes, row pitch dim + 4; (b) the same with 64 active lanes;
// prints shader cycles per element (s_memtime) and the shader clock (s_memtime / s_memrealtime x 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out, unsigned long long *cyc, int dim, int active) {
    extern __shared__ __attribute__((aligned(16))) float rows[];
    const int tid = threadIdx.x, pitch = dim + 4;
    for (int i = tid; i < 64 * pitch; i += blockDim.x) rows[i] = 1.0f + (i & 7) * 1e-3f;
    __syncthreads();
    float acc = 0.f;
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    if (tid < active) {
        const float *rowp = rows + (size_t)tid * pitch;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        float4 a[8], b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = *reinterpret_cast<const float4 *>(rowp + 4 * q);
        for (int k = 0; k < dim; k += 64) {
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = *reinterpret_cast<const float4 *>(rowp + k + 32 + 4 * q);
#pragma unroll
            for (int q = 0; q < 8; ++q) { acc = acc + a[q].x; acc = acc + a[q].y; acc = acc + a[q].z; acc = acc + a[q].w; }
            const int kn = k + 64 < dim ? k + 64 : k;
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = *reinterpret_cast<const float4 *>(rowp + kn + 4 * q);
#pragma unroll
            for (int q = 0; q < 8; ++q) { acc = acc + b[q].x; acc = acc + b[q].y; acc = acc + b[q].z; acc = acc + b[q].w; }
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    }
    if (tid == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
    out[tid] = acc;
}
int main() {
    float *o; unsigned long long *c, h[2];
    hipMalloc(&o, 1024); hipMalloc(&c, 16);
    const int dim = 1280;
    const size_t lds = (size_t)64 * (dim + 4) * 4;      // 328 KB would not fit: 16 rows only for the 16-lane case, 64 rows need a smaller dim
    for (int active : {16, 64}) {
        const int d = active == 16 ? 1280 : 512;
        const size_t bytes = (size_t)64 * (d + 4) * 4;
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(256), bytes, 0, o, c, d, active); hipMemcpy(h, c, 16, hipMemcpyDeviceToHost); }
        printf("%d active lanes, %d elements: %.2f shader cycles per element, clock %.0f MHz\n", active, d, (double)h[0] / d, (double)h[0] / h[1] * 100.0);
    }
    (void)lds;
    return 0;
}
