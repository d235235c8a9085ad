// scratch/mfma/seam4.hip -- can the launch of decode kernel k+1 and its weight stream hide behind kernel k?  The phases of seam3 (every workgroup needs the whole vector of the
// previous phase), each now also streaming a private slice of "weights" (WB bytes per workgroup) that does not depend on the vector:
//   (A) 140 dependent launches on one stream, captured in a graph (the decode engine's form): launch boundary -> weights + vector fetched -> body;
//   (B) the same kernels alternating on TWO streams captured into one graph: kernel k+1 is ordered after k-1 only, starts while k runs, issues its weight loads and then polls the
//       vector k produces -- handed over as {value, epoch} pairs (seam3: agent-scope relaxed 64-bit atomics, the data is the flag).  Every (phase) buffer is written once per replay,
//       the epoch is the replay number kept in device memory and advanced by a last kernel behind the join.
// Bounded spins: a time-out sets *err and the workgroup leaves.
//   hipcc --offload-arch=gfx950 -O3 seam4.hip -o seam4 && ./seam4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int VEC = 1536, NT = 256, J = VEC / NT;
typedef unsigned long long u64;
__device__ __forceinline__ u64 pack(float v, unsigned e) { return ((u64)e << 32) | (u64)__float_as_uint(v); }

template <int NW>      // uint4 loads per thread: the workgroup streams NW * 4 KB of weights
__device__ __forceinline__ unsigned weight_stream(const uint4 *w, int wg) {
    uint4 r[NW];
    const uint4 *p = w + (size_t)wg * NW * NT + threadIdx.x;
#pragma unroll
    for (int i = 0; i < NW; ++i) r[i] = p[i * NT];
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) x ^= r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
    return x;      // all-zero weights: contributes 0.0f without letting the loads be dropped
}

template <int NW>
__global__ __launch_bounds__(NT) void phase_plain(const float *in, float *out, const uint4 *w, int G) {
    __shared__ float red[4];
    const unsigned wx = weight_stream<NW>(w, blockIdx.x);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < J; ++j) s += in[threadIdx.x + j * NT];
    s += __uint_as_float(wx & 0x007fffffu);      // zero for zero weights
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const int per = VEC / G;
    if ((int)threadIdx.x < per) out[blockIdx.x * per + threadIdx.x] = tot * 1e-3f + (float)threadIdx.x;
}

template <int NW>
__global__ __launch_bounds__(NT) void phase_flag(const u64 *in, u64 *out, const uint4 *w, int G, const unsigned *epoch_dev, int first, int *err, int limit) {
    __shared__ float red[4];
    // the weight loads are in flight before the first poll
    uint4 r[NW];
    const uint4 *p = w + (size_t)blockIdx.x * NW * NT + threadIdx.x;
#pragma unroll
    for (int i = 0; i < NW; ++i) r[i] = p[i * NT];
    const unsigned epoch = *epoch_dev;
    u64 v[J];
    if (first) {
#pragma unroll
        for (int j = 0; j < J; ++j) v[j] = pack(1.0f, epoch);
    } else {
        int polls = 0;
        bool ok;
        do {
#pragma unroll
            for (int j = 0; j < J; ++j) v[j] = __hip_atomic_load(in + threadIdx.x + j * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = true;
#pragma unroll
            for (int j = 0; j < J; ++j) ok &= (unsigned)(v[j] >> 32) == epoch;
            if (!ok && ++polls > limit) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        } while (!ok);
    }
    unsigned wx = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) wx ^= r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < J; ++j) s += __uint_as_float((unsigned)v[j]);
    s += __uint_as_float(wx & 0x007fffffu);
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const int per = VEC / G;
    if ((int)threadIdx.x < per)
        __hip_atomic_store(out + blockIdx.x * per + threadIdx.x, pack(tot * 1e-3f + (float)threadIdx.x, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void bump(unsigned *e) { *e += 1; }
__global__ void fill_plain(float *a) { a[blockIdx.x * blockDim.x + threadIdx.x] = 1.0f; }

template <int NW>
int run(int G, int ncu) {
    const int phases = 140, reps = 20;
    float *bufA; u64 *bufB; uint4 *w; int *err; unsigned *epoch;
    const size_t wbytes = (size_t)G * NW * NT * 16;
    CK(hipMalloc(&bufA, (size_t)(phases + 1) * VEC * 4)); CK(hipMalloc(&bufB, (size_t)(phases + 1) * VEC * 8)); CK(hipMalloc(&w, wbytes * 2)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&epoch, 4));
    CK(hipMemset(w, 0, wbytes * 2)); CK(hipMemset(bufB, 0, (size_t)(phases + 1) * VEC * 8)); CK(hipMemset(err, 0, 4));
    unsigned one = 1; CK(hipMemcpy(epoch, &one, 4, hipMemcpyHostToDevice));
    hipStream_t st, s2; CK(hipStreamCreate(&st)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    // odd / even phases read different halves of the weight buffer so that no phase finds its slice in L2 left by the previous one
    // (A)
    hipGraph_t gA; hipGraphExec_t geA;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(fill_plain, dim3(VEC / 256), dim3(256), 0, st, bufA);
    for (int p = 0; p < phases; ++p)
        hipLaunchKernelGGL(phase_plain<NW>, dim3(G), dim3(NT), 0, st, bufA + (size_t)p * VEC, bufA + (size_t)(p + 1) * VEC, w + (p & 1) * (wbytes / 16), G);
    CK(hipStreamEndCapture(st, &gA));
    CK(hipGraphInstantiate(&geA, gA, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(geA, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(geA, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float msA = 0; CK(hipEventElapsedTime(&msA, e0, e1));
    std::vector<float> ra(VEC); CK(hipMemcpy(ra.data(), bufA + (size_t)phases * VEC, VEC * 4, hipMemcpyDeviceToHost));
    // (B)
    hipGraph_t gB; hipGraphExec_t geB;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    CK(hipEventRecord(fork, st)); CK(hipStreamWaitEvent(s2, fork, 0));
    for (int p = 0; p < phases; ++p)
        hipLaunchKernelGGL(phase_flag<NW>, dim3(G), dim3(NT), 0, (p & 1) ? s2 : st, bufB + (size_t)p * VEC, bufB + (size_t)(p + 1) * VEC, w + (p & 1) * (wbytes / 16), G, epoch, p == 0 ? 1 : 0, err, 20000);
    CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(st, join, 0));
    hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, st, epoch);
    CK(hipStreamEndCapture(st, &gB));
    CK(hipGraphInstantiate(&geB, gB, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(geB, st));
    CK(hipStreamSynchronize(st));
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    float msB = 0;
    if (!herr) {
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(geB, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&msB, e0, e1));
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    }
    std::vector<u64> rb(VEC); CK(hipMemcpy(rb.data(), bufB + (size_t)phases * VEC, VEC * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < VEC; ++i) { float v; unsigned u = (unsigned)rb[i]; memcpy(&v, &u, 4); if (v != ra[i]) ++bad; }
    printf("G %3d workgroups x %3d KB of weights each (%.1f MB per phase): (A) dependent launches %.2f us per phase (%.2f TB/s) | (B) two streams, data-as-flag %.2f us per phase (%.2f TB/s)%s%s\n",
           G, NW * 4, wbytes / 1e6, msA * 1e3 / (reps * phases), wbytes / (msA * 1e-3 / (reps * phases)) / 1e12, msB * 1e3 / (reps * phases),
           msB > 0 ? wbytes / (msB * 1e-3 / (reps * phases)) / 1e12 : 0.0, herr ? " TIMED OUT (invalid)" : "", bad ? " MISMATCH" : " results equal");
    fflush(stdout);
    CK(hipGraphExecDestroy(geA)); CK(hipGraphDestroy(gA)); CK(hipGraphExecDestroy(geB)); CK(hipGraphDestroy(gB));
    CK(hipFree(bufA)); CK(hipFree(bufB)); CK(hipFree(w)); CK(hipFree(err)); CK(hipFree(epoch));
    CK(hipStreamDestroy(st)); CK(hipStreamDestroy(s2));
    return herr ? 2 : 0;
}

int main() {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    for (int G : {96, 256}) {
        int rc;
        if ((rc = run<1>(G, ncu))) return rc;       // 4 KB per workgroup: o / q|k|v class (1.0 MB per phase at 256)
        if ((rc = run<4>(G, ncu))) return rc;       // 16 KB
        if ((rc = run<8>(G, ncu))) return rc;       // 32 KB: down class (8.4 MB per phase at 256)
        if ((rc = run<16>(G, ncu))) return rc;      // 64 KB: gate|up class (16.8 MB per phase at 256)
    }
    return 0;
}
