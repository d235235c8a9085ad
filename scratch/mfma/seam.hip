// scratch/mfma/seam.hip -- what does an all-to-all hand-off between two decode GEMVs cost on this MI355X: a kernel boundary (hipGraph of dependent launches) or a
// device-wide barrier inside one persistent launch?  Every phase is the skeleton of a fused decode kernel: each of G workgroups reads the WHOLE 1536-float vector
// the previous phase produced (the RMSNorm / Q8_K prologue needs all of it), does a token amount of arithmetic, and writes its own slice of the next vector.
//   hipcc --offload-arch=gfx950 -O3 seam.hip -o seam && ./seam
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int VEC = 1536, NT = 256;

__device__ __forceinline__ float phase_body(const float *in, float *out, int G, int wg, float *red) {
    // the whole vector, summed (order irrelevant here), then this workgroup's slice of the next one
    float s = 0.0f;
    for (int i = threadIdx.x; i < VEC; i += NT) s += in[i];
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const int per = VEC / G;
    if ((int)threadIdx.x < per) out[wg * per + threadIdx.x] = tot * 1e-3f + (float)threadIdx.x;
    __syncthreads();
    return tot;
}
__global__ __launch_bounds__(NT) void phase_kernel(const float *in, float *out, int G) {
    __shared__ float red[4];
    phase_body(in, out, G, blockIdx.x, red);
}
// one launch, `phases` phases, a device-wide barrier (agent-scope release / acquire around a counter) between them; the vector ping-pongs between two buffers.
// A spin that exceeds `limit` polls sets *err and every workgroup leaves (no hang).
__global__ __launch_bounds__(NT) void persistent_kernel(float *a, float *b, int G, int phases, unsigned *counter, int *err, int limit) {
    __shared__ float red[4];
    __shared__ int bail;
    if (threadIdx.x == 0) bail = 0;
    __syncthreads();
    for (int p = 0; p < phases; ++p) {
        const float *in = (p & 1) ? b : a;
        float *out = (p & 1) ? a : b;
        phase_body(in, out, G, blockIdx.x, red);
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(p + 1) * (unsigned)G;
            int polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bail = 1; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (bail) return;
    }
}
// the same with a two-level barrier: the workgroups of an XCD (blockIdx.x % 8 under the dispatcher's round-robin) meet on their own counter, the last of each XCD
// bumps the global one that everybody polls -- 8 + G/8 arrivals on a line instead of G
__global__ __launch_bounds__(NT) void persistent2_kernel(float *a, float *b, int G, int phases, unsigned *cx, unsigned *cg, int *err, int limit) {
    __shared__ float red[4];
    __shared__ int bail;
    if (threadIdx.x == 0) bail = 0;
    __syncthreads();
    const int xcd = blockIdx.x & 7, per = G / 8;
    for (int p = 0; p < phases; ++p) {
        const float *in = (p & 1) ? b : a;
        float *out = (p & 1) ? a : b;
        phase_body(in, out, G, blockIdx.x, red);
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned old = __hip_atomic_fetch_add(cx + 32 * xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)(p + 1) * per - 1) __hip_atomic_fetch_add(cg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(p + 1) * 8u;
            int polls = 0;
            while (__hip_atomic_load(cg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++polls > limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bail = 1; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (bail) return;
    }
}
int main() {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    float *a, *b; unsigned *counter, *cx; int *err;
    CK(hipMalloc(&a, VEC * 4)); CK(hipMalloc(&b, VEC * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&cx, 8 * 128));
    std::vector<float> h(VEC, 1.0f);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int phases = 140;      // a decode token has 142 launches
    for (int G : {64, 128, 256}) {
        if (G > ncu) continue;
        // (A) a captured graph of `phases` dependent launches, replayed
        CK(hipMemcpy(a, h.data(), VEC * 4, hipMemcpyHostToDevice));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(NT), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, G);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const int reps = 20;
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float msA = 0; CK(hipEventElapsedTime(&msA, e0, e1));
        // (A') the same launches enqueued on the stream, no graph
        for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(NT), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, G);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r)
            for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(NT), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, G);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float msS = 0; CK(hipEventElapsedTime(&msS, e0, e1));
        // (B) one persistent launch with in-launch barriers
        float msB = 0; int herr = 0;
        for (int r = 0; r < 3 + reps; ++r) {
            CK(hipMemsetAsync(counter, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st));
            if (r == 3) CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(persistent_kernel, dim3(G), dim3(NT), 0, st, a, b, G, phases, counter, err, 200000);
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&msB, e0, e1));
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        float msC = 0; int herr2 = 0;
        for (int r = 0; r < 3 + reps; ++r) {
            CK(hipMemsetAsync(counter, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(cx, 0, 8 * 128, st));
            if (r == 3) CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(persistent2_kernel, dim3(G), dim3(NT), 0, st, a, b, G, phases, cx, counter, err, 200000);
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&msC, e0, e1));
        CK(hipMemcpy(&herr2, err, 4, hipMemcpyDeviceToHost));
        printf("G %3d workgroups: stream launches without a graph %.2f us per phase | ", G, msS * 1e3 / (reps * phases));
        printf("G %3d workgroups: graph of dependent launches %.2f us per phase | persistent, one counter %.2f%s | persistent, per-XCD + global counters %.2f%s\n", G,
               msA * 1e3 / (reps * phases), msB * 1e3 / (reps * phases), herr ? " (TIMED OUT: invalid)" : "", msC * 1e3 / (reps * phases), herr2 ? " (TIMED OUT: invalid)" : "");
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
