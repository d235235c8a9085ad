// Does v_mfma_f32_32x32x2_f32 accumulate as fma(a1,b1, fma(a0,b0,c)) (sequential fp32 fma chain in k order)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(const float *A, const float *B, float *C, int K) {   // A [32][K], B [K][32] -> C [32][32]
    const int lane = threadIdx.x, col = lane & 31, h = lane >> 5;
    v16f acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[col * K + k0 + h], B[(k0 + h) * 32 + col], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + col] = acc[i];
}
int main() {
    const int K = 64;
    std::vector<float> A(32 * K), B(K * 32), C(1024), R1(1024), R2(1024);
    srand(1);
    for (auto &v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 3.0f;
    for (auto &v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 3.0f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad_seq = 0, bad_pair = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        float s = 0.0f; for (int kk = 0; kk < K; ++kk) s = fmaf(A[m * K + kk], B[kk * 32 + n], s);           // sequential fma chain
        float p = 0.0f; for (int kk = 0; kk < K; kk += 2) p = p + (A[m * K + kk] * B[kk * 32 + n] + A[m * K + kk + 1] * B[(kk + 1) * 32 + n]);
        if (s != C[m * 32 + n]) bad_seq++;
        if (p != C[m * 32 + n]) bad_pair++;
    }
    printf("mfma_f32_32x32x2: mismatches vs sequential fma chain %d / 1024, vs pairwise %d / 1024\n", bad_seq, bad_pair);
    return 0;
}
