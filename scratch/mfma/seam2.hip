// scratch/mfma/seam2.hip -- decomposition of the dependent-launch floor: graphs of 140 launches of (a) an empty kernel, (b) a kernel whose lanes each load one
// float of the predecessor's vector and store it back, (c) the read-reduce-write skeleton of seam.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int VEC = 1536, NT = 256;
__global__ __launch_bounds__(NT) void k_empty(const float *, float *, int) {}
__global__ __launch_bounds__(NT) void k_copy(const float *in, float *out, int G) {
    const int per = VEC / G;
    if ((int)threadIdx.x < per) out[blockIdx.x * per + threadIdx.x] = in[blockIdx.x * per + threadIdx.x] + 1.0f;
}
__global__ __launch_bounds__(NT) void k_full(const float *in, float *out, int G) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int i = threadIdx.x; i < VEC; i += NT) s += in[i];
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const int per = VEC / G;
    if ((int)threadIdx.x < per) out[blockIdx.x * per + threadIdx.x] = tot * 1e-3f + (float)threadIdx.x;
}
// (d) the same as (c) with the vector fetched as two float4 per thread issued together (the form the decode kernels use)
__global__ __launch_bounds__(NT) void k_full4(const float *in, float *out, int G) {
    __shared__ float red[4];
    const float4 *in4 = reinterpret_cast<const float4 *>(in);
    const float4 v0 = in4[threadIdx.x];
    const float4 v1 = threadIdx.x < VEC / 4 - NT ? in4[NT + threadIdx.x] : make_float4(0, 0, 0, 0);
    float s = ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w));
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const int per = VEC / G;
    if ((int)threadIdx.x < per) out[blockIdx.x * per + threadIdx.x] = tot * 1e-3f + (float)threadIdx.x;
}
// (e) (d) without the cross-wave part: every wave reads the whole vector itself (six float4 per lane) and reduces it with shuffles only -- no LDS, no barrier
__global__ __launch_bounds__(NT) void k_wave(const float *in, float *out, int G) {
    const float4 *in4 = reinterpret_cast<const float4 *>(in);
    const int lane = threadIdx.x & 63;
    float s = 0.0f;
    float4 v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = in4[lane + 64 * i];
#pragma unroll
    for (int i = 0; i < 6; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    const int per = VEC / G;
    if ((int)threadIdx.x < per) out[blockIdx.x * per + threadIdx.x] = s * 1e-3f + (float)threadIdx.x;
}
template <typename K> static int run(K kern, const char *name, float *a, float *b, int G, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    const int phases = 140, reps = 20;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(kern, dim3(G), dim3(NT), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, G);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-44s %.2f us per launch\n", name, ms * 1e3 / (reps * phases));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return 0;
}
int main() {
    float *a, *b;
    CK(hipMalloc(&a, VEC * 4)); CK(hipMalloc(&b, VEC * 4)); CK(hipMemset(a, 0, VEC * 4)); CK(hipMemset(b, 0, VEC * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int G : {24, 256}) {
        printf("G = %d workgroups of 256 threads, 140 dependent launches in one hipGraph:\n", G);
        if (run(k_empty, "(a) empty kernel", a, b, G, st, e0, e1)) return 1;
        if (run(k_copy, "(b) each workgroup copies its slice", a, b, G == 24 ? 24 : 256, st, e0, e1)) return 1;
        if (run(k_full, "(c) read whole vector, reduce, write slice", a, b, G == 24 ? 24 : 256, st, e0, e1)) return 1;
        if (run(k_full4, "(d) the same, two float4 per thread at once", a, b, G == 24 ? 24 : 256, st, e0, e1)) return 1;
        if (run(k_wave, "(e) per-wave read + shuffle reduce, no barrier", a, b, G == 24 ? 24 : 256, st, e0, e1)) return 1;
    }
    return 0;
}
