// scratch/mfma/dma_off.hip -- does the instruction offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned *src, unsigned *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds;
    const unsigned voff = threadIdx.x * 16u;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(src) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;          // dword i holds i
    unsigned *src, *out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, 2048 * 4);
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, src, out);
    std::vector<unsigned> o(2048);
    hipMemcpy(o.data(), out, 2048 * 4, hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int i = 0; i < 2048; ++i) if (o[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
    printf("LDS dwords written: [%d, %d]; LDS[%d] = source dword %u (offset:1024 = 256 dwords)\n", first, last, first, first >= 0 ? o[first] : 0);
    return 0;
}
