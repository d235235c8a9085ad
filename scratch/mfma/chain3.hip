// dependent fp32 add chain fed from LDS (ping-pong ds_read_b128), as in ln_stats: ns per add vs active lanes and row pitch
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LANES>
__global__ void k(float *out, int dim, int pitch, unsigned long long *t) {
    extern __shared__ __attribute__((aligned(16))) float buf[];
    for (int i = threadIdx.x; i < 16 * pitch; i += blockDim.x) buf[i] = 1e-3f * (i & 255);
    __syncthreads();
    if (threadIdx.x >= LANES) return;
    const float *rowp = buf + (threadIdx.x & 15) * pitch;
    float acc = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float4 A[4], B[4];
    auto ld = [&](float4 (&v)[4], int kk) { for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4 *>(rowp + kk + 4 * q); };
    auto proc = [&](const float4 (&v)[4]) { for (int q = 0; q < 4; ++q) { acc = acc + v[q].x; acc = acc + v[q].y; acc = acc + v[q].z; acc = acc + v[q].w; } };
    int k = 0;
    ld(A, 0);
    for (; k + 32 <= dim; k += 32) { ld(B, k + 16); proc(A); if (k + 48 <= dim) ld(A, k + 32); proc(B); }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
int main() {
    float *out; unsigned long long *t; hipMalloc(&out, 1 << 20); hipMalloc(&t, 8 * 1024);
    unsigned long long h[1024];
    const int dim = 1280;
    for (int pitch : {1284, 1280, 1281}) {
        const size_t lds = 16 * pitch * 4;
        hipFuncSetAttribute((const void *)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipFuncSetAttribute((const void *)k<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        for (int lanes : {16, 64}) {
            for (int rep = 0; rep < 3; ++rep) { if (lanes == 16) hipLaunchKernelGGL(k<16>, dim3(64), dim3(256), lds, 0, out, dim, pitch, t); else hipLaunchKernelGGL(k<64>, dim3(64), dim3(256), lds, 0, out, dim, pitch, t); }
            hipDeviceSynchronize();
            hipMemcpy(h, t, 64 * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < 64; ++i) avg += h[i]; avg /= 64;
            printf("pitch %d lanes %d: %.2f ns per add (%.2f us per pass)\n", pitch, lanes, avg * 10.0 / dim, avg * 10.0 / 1000);
        }
    }
    return 0;
}
