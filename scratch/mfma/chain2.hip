// dependent-chain cost of v_fma_f32 vs v_fma_mix_f32 (fp16 operand) in short and long runs; 12 WGs x 1024 threads, waves 0-1 work
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MIX>
__global__ void chain(float *out, const unsigned *hv, float a, int n, unsigned long long *t) {
    float x = threadIdx.x * 1e-9f;
    unsigned w[8];
    for (int i = 0; i < 8; ++i) w[i] = hv[threadIdx.x * 8 + i];
    float f[16];
    for (int i = 0; i < 16; ++i) f[i] = (float)((const _Float16 *)w)[i];
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x < 128)
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (MIX) x = __builtin_fmaf(a, (float)((const _Float16 *)w)[k], x);
            else x = __builtin_fmaf(a, f[k], x);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
int main() {
    float *out; unsigned long long *t; unsigned *hv;
    hipMalloc(&out, 4096 * 1024 * 4); hipMalloc(&t, 4096 * 8); hipMalloc(&hv, 1024 * 8 * 4);
    hipMemset(hv, 0x3c, 1024 * 8 * 4);
    unsigned long long h[4096];
    for (int n : {8, 32, 1024}) for (int mix = 0; mix < 2; ++mix) {
        for (int rep = 0; rep < 5; ++rep) {
            if (mix) hipLaunchKernelGGL(chain<1>, dim3(12), dim3(1024), 0, 0, out, hv, 0.999f, n, t);
            else hipLaunchKernelGGL(chain<0>, dim3(12), dim3(1024), 0, 0, out, hv, 0.999f, n, t);
        }
        hipDeviceSynchronize();
        hipMemcpy(h, t, 12 * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 12; ++i) avg += h[i]; avg /= 12;
        printf("n %4d (%5d fmas) mix %d: %.2f ns per dependent fma\n", n, n * 16, mix, avg * 10.0 / (n * 16));
    }
    return 0;
}
