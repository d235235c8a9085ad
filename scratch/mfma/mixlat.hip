// scratch/mfma/mixlat.hip -- dependent-chain cost of v_fma_mix_f32 (fp16 operand folded into the fma) against v_cvt_f32_f16 + v_fma_f32 (conversion off the chain), one wave per
// SIMD, 32-link blocks like the decode attention's walk.   hipcc --offload-arch=gfx950 -O3 mixlat.hip -o mixlat && ./mixlat
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(64) void chain(const unsigned *__restrict__ hv, const float *__restrict__ pv, float *out, unsigned long long *cycles, int iters) {
    unsigned w[16]; float p[32];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = hv[threadIdx.x * 16 + i];
#pragma unroll
    for (int i = 0; i < 32; ++i) p[i] = pv[i];
    float o = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const unsigned wk = w[k >> 1];
            if (MODE == 0) {      // the form the compiler folds the walk's conversion into
                if (k & 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(o) : "v"(p[k]), "v"(wk));
                else asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(o) : "v"(p[k]), "v"(wk));
            } else {              // conversion kept as its own instruction (independent of the chain), plain v_fma_f32 on the chain
                float f;
                if (k & 1) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "v"(wk));
                else asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(f) : "v"(wk));
                o = __fmaf_rn(p[k], f, o);
            }
        }
        asm volatile("" : "+v"(o));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = o;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned *hv; float *pv, *out; unsigned long long *cy;
    CK(hipMalloc(&hv, 64 * 16 * 4)); CK(hipMalloc(&pv, 32 * 4)); CK(hipMalloc(&out, 1024 * 64 * 4)); CK(hipMalloc(&cy, 1024 * 8));
    CK(hipMemset(hv, 0x3c, 64 * 16 * 4)); CK(hipMemset(pv, 0, 32 * 4));
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(24), dim3(64), 0, 0, hv, pv, out, cy, iters);
            else hipLaunchKernelGGL(chain<1>, dim3(24), dim3(64), 0, 0, hv, pv, out, cy, iters);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long c; CK(hipMemcpy(&c, cy, 8, hipMemcpyDeviceToHost));
            printf("%s: %.1f ns per link (event), %.2f s_memtime ticks per link\n", mode ? "v_cvt_f32_f16 + v_fma_f32" : "v_fma_mix_f32           ", ms * 1e6 / (iters * 32.0), (double)c / (iters * 32.0));
        }
    return 0;
}
