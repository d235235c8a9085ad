// Dependent-chain cost of the candidates for the decode walk's step o = fma(p_j, v_j, o), one wave alone on its SIMD (s_memtime = shader cycles):
//   A  v_fma_mix_f32 (fp16 value operand converted inside the instruction)      -- what the round-1 walker issues
//   B  v_fma_f32 on fp32 operands
//   C  v_fma_f32 with an independent v_cvt_f32_f16 between two steps (the conversion of a later key)
//   D  v_add_f32 (the logsum lane)
//   E  v_mfma_f32_16x16x4_f32 chained through its accumulator (four keys per instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void k(float *out, unsigned long long *cyc, float p, unsigned h) {
    float o = out[threadIdx.x], c = 0.f;
    unsigned long long t0, t1;
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP16(REP16("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]\n\t")) : "+v"(o) : "v"(p), "v"(h));
    asm volatile("s_nop 0" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP16(REP16("v_fma_f32 %0, %1, %2, %0\n\t")) : "+v"(o) : "v"(p), "v"(p));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[1] = t1 - t0;
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP16(REP16("v_fma_f32 %0, %2, %1, %0\n\tv_cvt_f32_f16 %1, %3\n\t")) : "+v"(o), "+v"(c) : "v"(p), "v"(h));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[2] = t1 - t0;
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP16(REP16("v_add_f32 %0, %1, %0\n\t")) : "+v"(o) : "v"(p));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[3] = t1 - t0;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f acc = {o, o, o, o};
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 64; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p, c, acc, 0, 0, 0);
    asm volatile("s_nop 7\n\ts_nop 7" :: "v"(acc));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[4] = t1 - t0;
    // F: two dependent fma chains interleaved in one wave (two dims per lane)
    float o2 = o + 1.0f;
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP16(REP16("v_fma_f32 %0, %2, %2, %0\n\tv_fma_f32 %1, %2, %2, %1\n\t")) : "+v"(o), "+v"(o2) : "v"(p));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[5] = t1 - t0;
    // G: v_pk_fma_f32 chain (two dims per instruction)
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f oo = {o, o2}, pp = {p, p};
    t0 = __builtin_amdgcn_s_memtime();
    asm volatile(REP16(REP16("v_pk_fma_f32 %0, %1, %1, %0\n\t")) : "+v"(oo) : "v"(pp));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[6] = t1 - t0;
    out[threadIdx.x] = o + c + acc[0] + oo[0] + oo[1];
}
int main() {
    float *o; unsigned long long *c, h[8];
    hipMalloc(&o, 256); hipMalloc(&c, 64); hipMemset(o, 0, 256);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 1.0000001f, 0x3c003c00u);
        hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
    }
    printf("cycles per step (one wave alone): fma_mix %.1f | fma_f32 %.1f | fma_f32 + independent cvt %.1f (per pair) | add_f32 %.1f | mfma16x16x4 %.1f per instr = %.1f per key | two interleaved fma chains %.1f per pair | pk_fma_f32 %.1f\n",
           h[0] / 256.0, h[1] / 256.0, h[2] / 256.0, h[3] / 256.0, h[4] / 64.0, h[4] / 256.0, h[5] / 256.0, h[6] / 256.0);
    return 0;
}
