// Does a kernel's CODE SIZE cost time at every dependent launch?  K kernels of `LINES` 64-byte code lines each (every wave touches every line once: a branch over 15 nops),
// launched round-robin as a dependent chain inside one captured hipGraph, 256 workgroups x 256 threads like a decode GEMV.  If the instruction cache (64 KiB per two CUs on
// CDNA3) keeps what it fetched, a rotation whose code fits stays hot; one that does not re-fetches every kernel from L2 at every launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int TAG, int LINES>
__global__ __launch_bounds__(256) void bigcode(float *out) {
    if (LINES == 64) asm volatile(".rept 64\n s_branch 15\n .fill 15, 4, 0xBF800000\n .endr" ::: "memory");
    if (LINES == 256) asm volatile(".rept 256\n s_branch 15\n .fill 15, 4, 0xBF800000\n .endr" ::: "memory");
    if (LINES == 384) asm volatile(".rept 384\n s_branch 15\n .fill 15, 4, 0xBF800000\n .endr" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x + TAG] = (float)TAG;
}
typedef void (*kern_t)(float *);
template <int LINES> static void run(const char *label, int nk, float *o) {
    kern_t ks[6] = {bigcode<0, LINES>, bigcode<1, LINES>, bigcode<2, LINES>, bigcode<3, LINES>, bigcode<4, LINES>, bigcode<5, LINES>};
    hipStream_t st; hipStreamCreate(&st);
    const int per = 140;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < per; ++i) hipLaunchKernelGGL(ks[i % nk], dim3(256), dim3(256), 0, st, o);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 5; ++w) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, st);
        for (int w = 0; w < 50; ++w) hipGraphLaunch(ge, st);
        hipEventRecord(e1, st); hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s rotation of %d kernels x %3d KiB: %.3f us per launch\n", label, nk, LINES * 64 / 1024, ms * 1e3 / (50 * per));
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
}
int main() {
    float *o; hipMalloc(&o, (256 * 256 + 16) * 4);
    run<64>("4 KiB kernels", 1, o); run<64>("4 KiB kernels", 6, o);
    run<256>("16 KiB kernels", 1, o); run<256>("16 KiB kernels", 2, o); run<256>("16 KiB kernels", 3, o); run<256>("16 KiB kernels", 4, o); run<256>("16 KiB kernels", 5, o); run<256>("16 KiB kernels", 6, o);
    run<384>("24 KiB kernels", 1, o); run<384>("24 KiB kernels", 2, o); run<384>("24 KiB kernels", 3, o); run<384>("24 KiB kernels", 5, o);
    return 0;
}
