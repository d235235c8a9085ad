// What clock does a lightly loaded kernel run at?  s_memtime (clock64, shader clock) against s_memrealtime (wall_clock64, 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float *out, long long *t, int n) {
    float a = threadIdx.x * 1e-9f, b = 1.0000001f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) a = __builtin_fmaf(a, b, 1e-9f);
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
    float *o; long long *t, h[2];
    hipMalloc(&o, 4 << 20 << 4); hipMalloc(&t, 16);
    int n = 200000;
    struct { int g, b; const char *name; } cfg[] = {{1, 64, "1 wave"}, {256, 64, "1 wave/CU"}, {256, 256, "4 waves/CU"}, {256 * 8, 256, "32 waves/CU"}, {256 * 8, 1024, "full 8 wg x16 waves"}};
    for (auto &c : cfg)
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); chain<<<c.g, c.b>>>(o, t, n); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            printf("%-22s rep %d: %.3f ms  clock64 %lld  wall(100MHz) %lld -> %.0f MHz  ; %.2f ns/fma  %.2f clk/fma\n", c.name, rep, ms, h[0], h[1],
                   100.0 * h[0] / h[1], 10.0 * h[1] / n, (double)h[0] / n);
        }
    // short kernels back to back (decode-like): 2000 launches of a 5 us chain
    n = 1500;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int i = 0; i < 2000; ++i) chain<<<256, 256>>>(o, t, n);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("2000 short launches rep %d: %.3f us/launch  clock64 %lld wall %lld -> %.0f MHz  %.2f ns/fma\n", rep, ms * 1e3 / 2000, h[0], h[1], 100.0 * h[0] / h[1], 10.0 * h[1] / n);
    }
    return 0;
}
