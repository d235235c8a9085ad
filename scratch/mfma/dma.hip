// LDS-DMA staging test: contiguous run of bytes -> LDS at (static + dynamic) offsets, read back by the same wave after vmcnt(0)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst_in) {
    unsigned keep;
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_in);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int STATIC>
__global__ void k(const unsigned char *src, unsigned char *out, int run, int wave_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[8];
    if (STATIC && threadIdx.x < 8) red[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *stage = smem + 1920 + (size_t)wid * wave_bytes;
    const unsigned st0 = (unsigned)(size_t)stage;
    const unsigned char *s = src + (size_t)wid * run;
    for (int o = 0; o < run; o += 1024) {
        const int off = o + lane * 16;
        glds16(s + (off < run ? off : 0), st0 + (unsigned)o);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane * 16; i < run; i += 1024) *reinterpret_cast<uint4 *>(out + (size_t)wid * run + i) = *reinterpret_cast<const uint4 *>(stage + i);
    if (STATIC && threadIdx.x == 0) out[0] += (unsigned char)(red[3] * 0);
}
int main() {
    const int run = 4320, waves = 7, wb = 16064;
    std::vector<unsigned char> h(run * waves), o(run * waves);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char)(i * 131 + (i >> 8));
    unsigned char *d, *dout;
    hipMalloc(&d, h.size()); hipMalloc(&dout, h.size());
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    const size_t lds = 1920 + (size_t)waves * wb;
    hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int st = 0; st < 2; ++st) {
        hipMemset(dout, 0, h.size());
        if (st) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), lds, 0, d, dout, run, wb);
        else hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), lds, 0, d, dout, run, wb);
        hipDeviceSynchronize();
        hipMemcpy(o.data(), dout, h.size(), hipMemcpyDeviceToHost);
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < h.size(); ++i) if (h[i] != o[i]) { if (!bad) first = i; ++bad; }
        printf("static %d: %zu bad bytes of %zu (first at %zu)\n", st, bad, h.size(), first);
    }
    return 0;
}
