// scratch/mfma/seam3.hip -- the all-to-all seam once more, with the DATA as the flag: every element of the handed-over vector travels as a 64-bit {value, epoch} pair
// written with one agent-scope relaxed atomic store and polled with agent-scope relaxed atomic loads.  No counter (no G serialised read-modify-writes on one line), no
// release / acquire fences (no L2 write-back / invalidate): a consumer spins on the elements themselves until each carries this phase's epoch.  Two buffers by phase parity
// (a workgroup can only be one phase ahead of the slowest: it needs everybody's phase-p data before it writes phase p+1).  Spins are bounded; a time-out sets *err and
// every workgroup leaves.
//   hipcc --offload-arch=gfx950 -O3 seam3.hip -o seam3 && ./seam3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int VEC = 1536, NT = 256;
typedef unsigned long long u64;

__device__ __forceinline__ u64 pack(float v, unsigned e) { return ((u64)e << 32) | (u64)__float_as_uint(v); }

// reference: the same arithmetic as dependent launches (plain floats, float4 x 2 per thread issued together = body (d) of seam2)
__global__ __launch_bounds__(NT) void phase_kernel(const float *in, float *out, int G) {
    __shared__ float red[4];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < VEC / NT; ++j) s += in[threadIdx.x + j * NT];
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const int per = VEC / G;
    if ((int)threadIdx.x < per) out[blockIdx.x * per + threadIdx.x] = tot * 1e-3f + (float)threadIdx.x;
}

// persistent: phase p reads buffer p & 1 (elements tagged p), writes buffer (p + 1) & 1 tagged p + 1.  Buffer 0 is initialised by the host with epoch 0.
template <int MODE>      // 0: workgroup-wide (6 pairs per thread, LDS reduction, one barrier)   1: every wave reads the whole vector itself (24 pairs per lane, shuffles only)
__global__ __launch_bounds__(NT) void persistent_kernel(u64 *buf0, u64 *buf1, int G, int phases, int *err, int limit, unsigned long long *spins) {
    __shared__ float red[2][4];
    const int per = VEC / G;
    unsigned long long my_spins = 0;
    for (int p = 0; p < phases; ++p) {
        const u64 *in = (p & 1) ? buf1 : buf0;
        u64 *out = (p & 1) ? buf0 : buf1;
        float tot;
        if (MODE == 0) {
            constexpr int J = VEC / NT;
            u64 v[J];
            int polls = 0;
            bool ok;
            do {
#pragma unroll
                for (int j = 0; j < J; ++j) v[j] = __hip_atomic_load(in + threadIdx.x + j * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = true;
#pragma unroll
                for (int j = 0; j < J; ++j) ok &= (unsigned)(v[j] >> 32) == (unsigned)p;
                if (!ok && ++polls > limit) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }      // bounded: everybody who waits on this workgroup times out as well
            } while (!ok);
            my_spins += polls;
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < J; ++j) s += __uint_as_float((unsigned)v[j]);
            for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
            if ((threadIdx.x & 63) == 0) red[p & 1][threadIdx.x >> 6] = s;
            __syncthreads();
            tot = red[p & 1][0] + red[p & 1][1] + red[p & 1][2] + red[p & 1][3];
        } else {
            constexpr int J = VEC / 64;
            const int lane = threadIdx.x & 63;
            u64 v[J];
            int polls = 0;
            bool ok;
            do {
#pragma unroll
                for (int j = 0; j < J; ++j) v[j] = __hip_atomic_load(in + lane + j * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = true;
#pragma unroll
                for (int j = 0; j < J; ++j) ok &= (unsigned)(v[j] >> 32) == (unsigned)p;
                if (!ok && ++polls > limit) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }      // bounded: everybody who waits on this workgroup times out as well
            } while (!ok);
            my_spins += polls;
            // the same association as MODE 0 / the launched kernel: element i belongs to "thread" i % 256; wave w' of that kernel = i % 256 / 64
            float s4[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < J; ++j) s4[j & 3] += __uint_as_float((unsigned)v[j]);      // j & 3 == (lane + 64 j) % 256 / 64
#pragma unroll
            for (int k = 0; k < 4; ++k)
                for (int o = 32; o; o >>= 1) s4[k] += __shfl_xor(s4[k], o);
            tot = s4[0] + s4[1] + s4[2] + s4[3];
        }
        if ((int)threadIdx.x < per)
            __hip_atomic_store(out + blockIdx.x * per + threadIdx.x, pack(tot * 1e-3f + (float)threadIdx.x, (unsigned)(p + 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0 && spins) atomicAdd(spins, my_spins);
}

int main() {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    float *a, *b; u64 *p0, *p1; int *err; unsigned long long *spins;
    CK(hipMalloc(&a, VEC * 4)); CK(hipMalloc(&b, VEC * 4)); CK(hipMalloc(&p0, VEC * 8)); CK(hipMalloc(&p1, VEC * 8)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&spins, 8));
    std::vector<float> h(VEC, 1.0f), ra(VEC), rp(VEC);
    std::vector<u64> hp(VEC), hq(VEC);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int phases = 140, reps = 20;
    for (int G : {64, 128, 256}) {
        if (G > ncu) continue;
        CK(hipMemcpy(a, h.data(), VEC * 4, hipMemcpyHostToDevice));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(NT), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, G);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(ra.data(), a, VEC * 4, hipMemcpyDeviceToHost));      // phases is even: the last phase wrote a
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float msA = 0; CK(hipEventElapsedTime(&msA, e0, e1));
        float msP[2] = {0, 0}; int herr[2] = {0, 0}; int bad[2] = {0, 0}; double sp[2] = {0, 0};
        for (int mode = 0; mode < 2; ++mode) {
            for (int i = 0; i < VEC; ++i) { hp[i] = ((u64)0 << 32) | 0x3f800000ull; hq[i] = 0xffffffff00000000ull; }
            for (int r = 0; r < 3 + reps; ++r) {
                CK(hipMemcpyAsync(p0, hp.data(), VEC * 8, hipMemcpyHostToDevice, st));
                CK(hipMemcpyAsync(p1, hq.data(), VEC * 8, hipMemcpyHostToDevice, st));
                CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(spins, 0, 8, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                if (mode == 0) hipLaunchKernelGGL(persistent_kernel<0>, dim3(G), dim3(NT), 0, st, p0, p1, G, phases, err, 20000, spins);
                else hipLaunchKernelGGL(persistent_kernel<1>, dim3(G), dim3(NT), 0, st, p0, p1, G, phases, err, 20000, spins);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 3) msP[mode] += ms;
            }
            CK(hipMemcpy(&herr[mode], err, 4, hipMemcpyDeviceToHost));
            unsigned long long hs = 0; CK(hipMemcpy(&hs, spins, 8, hipMemcpyDeviceToHost));
            sp[mode] = (double)hs / ((double)G * NT * phases);
            CK(hipMemcpy(hp.data(), p0, VEC * 8, hipMemcpyDeviceToHost));      // phases even: the last phase wrote buffer 0 with epoch `phases`
            for (int i = 0; i < VEC; ++i) {
                float v; unsigned u = (unsigned)hp[i]; memcpy(&v, &u, 4);
                if (v != ra[i] || (unsigned)(hp[i] >> 32) != (unsigned)phases) ++bad[mode];
            }
        }
        printf("G %3d workgroups: graph of dependent launches %.2f us per phase | data-as-flag persistent, workgroup-wide %.2f us (one launch %.1f us; %s%s; %.2f re-polls per thread and phase) | "
               "every wave reads the vector %.2f us (%s%s; %.2f re-polls)\n", G, msA * 1e3 / (reps * phases), msP[0] * 1e3 / (reps * phases), msP[0] * 1e3 / reps,
               herr[0] ? "TIMED OUT " : "", bad[0] ? "MISMATCH" : "equal to the launches", sp[0], msP[1] * 1e3 / (reps * phases), herr[1] ? "TIMED OUT " : "", bad[1] ? "MISMATCH" : "equal", sp[1]);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
