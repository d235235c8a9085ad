// Does v_mfma_f32_16x16x4_f32 accumulate as a sequential fp32 fma chain in k order (k = 0..3 inside one instruction)?
// Operand layout (CDNA3 ISA, 16x16x4 f32): A: lane l holds A[row = l & 15][k = l >> 4]; B: lane l holds B[k = l >> 4][col = l & 15];
// C/D: 4 VGPRs, lane l, register i -> C[row = 4 * (l >> 4) + i][col = l & 15].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, float *C, int K) {   // A [16][K], B [K][16] -> C [16][16]
    const int lane = threadIdx.x, c16 = lane & 15, q = lane >> 4;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c16 * K + k0 + q], B[(k0 + q) * 16 + c16], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) C[(4 * q + i) * 16 + c16] = acc[i];
}
int main() {
    const int K = 64;
    std::vector<float> A(16 * K), B(K * 16), C(256);
    srand(1);
    for (auto &v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 3.0f;
    for (auto &v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 3.0f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    int bad_seq = 0, bad_plain = 0, bad_tree = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        float s = 0.0f; for (int kk = 0; kk < K; ++kk) s = fmaf(A[m * K + kk], B[kk * 16 + n], s);
        float t = 0.0f; for (int kk = 0; kk < K; kk += 4) {
            float p0 = A[m * K + kk] * B[kk * 16 + n], p1 = A[m * K + kk + 1] * B[(kk + 1) * 16 + n], p2 = A[m * K + kk + 2] * B[(kk + 2) * 16 + n], p3 = A[m * K + kk + 3] * B[(kk + 3) * 16 + n];
            t = t + ((p0 + p1) + (p2 + p3)); }
        double d = 0.0; for (int kk = 0; kk < K; ++kk) d += (double)A[m * K + kk] * B[kk * 16 + n];
        if (s != C[m * 16 + n]) bad_seq++;
        if (t != C[m * 16 + n]) bad_tree++;
        if ((float)d != C[m * 16 + n]) bad_plain++;
    }
    printf("mfma_f32_16x16x4: mismatches vs sequential fma chain %d / 256, vs per-instruction tree %d / 256, vs double-rounded %d / 256\n", bad_seq, bad_tree, bad_plain);
    return 0;
}
