// Do the matrix pipe and the VALU overlap across waves of one SIMD?  One workgroup of 8 waves: waves 0..3 (one per SIMD) issue dependent v_mfma_f32_32x32x2_f32, waves 4..7
// (their SIMD partners) issue v_fma_f64 or v_fma_f32 chains; each role alone and both together, s_memrealtime around the loops.
// hipcc --offload-arch=gfx950 -O3 -o overlap scratch/mfma/overlap.hip && ./overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
template <int MODE>      // bit 0: fp32 MFMA role runs, bit 1: fp64 role runs, bit 2: fp32 role runs, bit 3: the MFMA role issues f16 MFMAs (v_mfma_f32_32x32x16_f16) instead
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *t, int iters) {
    const int role = threadIdx.x >> 8;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (role == 0) {
        if (MODE & 8) {
            v16f a0 = {0}, a1 = {0};
            v8h x, y;
            for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 1e-3f + e); y[e] = (_Float16)(1.0f + e * 0.25f); }
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int s = 0; s < 5; ++s) { a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0); }
            }
            out[threadIdx.x] = a0[0] + a1[3];
        }
        if (MODE & 1) {
            v16f a0 = {0}, a1 = {0};
            float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int s = 0; s < 5; ++s) { a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0); }
            }
            out[threadIdx.x] = a0[0] + a1[3];
        }
    } else {
        if (MODE & 2) {
            double d0 = threadIdx.x * 1e-3, d1 = 1.0, d2 = 2.0, d3 = 3.0;
            const double m = 1.0000001, c = 1e-9;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int s = 0; s < 40; ++s) { d0 = __fma_rn(d0, m, c); d1 = __fma_rn(d1, m, c); d2 = __fma_rn(d2, m, c); d3 = __fma_rn(d3, m, c); }
            }
            out[threadIdx.x] = (float)(d0 + d1 + d2 + d3);
        }
        if (MODE & 4) {
            float d0 = threadIdx.x * 1e-3f, d1 = 1.0f, d2 = 2.0f, d3 = 3.0f;
            const float m = 1.0000001f, c = 1e-9f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int s = 0; s < 40; ++s) { d0 = __fmaf_rn(d0, m, c); d1 = __fmaf_rn(d1, m, c); d2 = __fmaf_rn(d2, m, c); d3 = __fmaf_rn(d3, m, c); }
            }
            out[threadIdx.x] = d0 + d1 + d2 + d3;
        }
        if (MODE & 16) {      // the same four fp32 chains as single v_fma_f32 (the compiler packs the plain form into v_pk_fma_f32)
            float d0 = threadIdx.x * 1e-3f, d1 = 1.0f, d2 = 2.0f, d3 = 3.0f;
            const float m = 1.0000001f, c = 1e-9f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int s = 0; s < 40; ++s) {
                    asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(m), "v"(c));
                }
            }
            out[threadIdx.x] = d0 + d1 + d2 + d3;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) t[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE> void run(const char *name, float *out, unsigned long long *t) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(512), 0, 0, out, t, iters); hipDeviceSynchronize(); }
    unsigned long long h[8];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-28s  MFMA waves %.1f us (%.0f cycles per 10 MFMAs)   VALU waves %.1f us (%.0f cycles per 160 fma)\n", name, h[0] / 100.0, h[0] * 24.0 / iters, h[4] / 100.0, h[4] * 24.0 / iters);
}
int main() {
    float *out; unsigned long long *t;
    hipMalloc(&out, 4096); hipMalloc(&t, 64);
    run<1>("MFMA alone", out, t);
    run<2>("fp64 fma alone", out, t);
    run<4>("fp32 fma alone", out, t);
    run<3>("MFMA + fp64 fma", out, t);
    run<5>("MFMA + fp32 fma", out, t);
    run<8>("f16 MFMA alone", out, t);
    run<10>("f16 MFMA + fp64 fma", out, t);
    run<12>("f16 MFMA + fp32 fma", out, t);
    run<16>("single v_fma_f32 alone", out, t);
    run<17>("MFMA + single v_fma_f32", out, t);
    run<24>("f16 MFMA + single v_fma_f32", out, t);
    return 0;
}
