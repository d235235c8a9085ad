// mllm_amd/csrc/runtime.hip -- device context, memory and error plumbing of libmllm_hip.
// Replaces Backend::{alloc_device,free_device,copy_from_host,copy_to_host} (mllm/Backend.hpp:60-73); structural precedent
// mllm/backends/opencl/OpenCLBackend.cpp:670-787 (the only discrete-device backend of the reference).
#include <cstdio>
#include <cstring>
#include <mutex>

#include <cstdarg>

#include "common.h"

namespace mllm_hip {
static thread_local char g_err[512] = "";

void set_error(const char *what, hipError_t e, const char *file, int line) {
    snprintf(g_err, sizeof(g_err), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
}
void set_error_msg(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int check_launch(const char *what, const char *file, int line) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(what, e, file, line);
        return MLLM_HIP_ERR_HIP;
    }
    return MLLM_HIP_OK;
}
struct OptionTable {
    int v[OPT_COUNT];
    OptionTable() { for (int &x : v) x = -1; }      // every Option unset, however many there are
};
static OptionTable g_option_table;
static int *const g_options = g_option_table.v;
static const char *const g_option_names[OPT_COUNT] = {"vision_batch", "time_layers", "no_gub", "no_pjb", "pjb_min_ns", "attn_flags", "attn_ds", "head_wpc", "gemm_order", "no_lnf", "merge_o", "chain_cont", "gu_persist", "qkv_persist"};
int option(Option o) { return g_options[o]; }
}  // namespace mllm_hip

using namespace mllm_hip;

extern "C" int mllm_hip_set_option(const char *name, int value) {
    if (!name) return MLLM_HIP_ERR_ARG;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(name, g_option_names[i]) == 0) { g_options[i] = value; return MLLM_HIP_OK; }
    snprintf(g_err, sizeof(g_err), "mllm_hip_set_option: unknown option '%s'", name);
    return MLLM_HIP_ERR_ARG;
}
extern "C" int mllm_hip_get_option(const char *name, int *value) {
    if (!name || !value) return MLLM_HIP_ERR_ARG;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(name, g_option_names[i]) == 0) { *value = g_options[i]; return MLLM_HIP_OK; }
    return MLLM_HIP_ERR_ARG;
}

extern "C" const char *mllm_hip_last_error(void) { return g_err; }

extern "C" int mllm_hip_init(int device) {
    int n = 0;
    MH_CHECK(hipGetDeviceCount(&n));
    if (n <= 0 || device < 0 || device >= n) {
        snprintf(g_err, sizeof(g_err), "mllm_hip_init: no such HIP device %d (count %d)", device, n);
        return MLLM_HIP_ERR_ARG;
    }
    MH_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    MH_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof(g_err), "mllm_hip_init: device arch %s is not gfx950 (this library is built for MI355X only)",
                 prop.gcnArchName);
        return MLLM_HIP_ERR_ARG;
    }
    return MLLM_HIP_OK;
}

extern "C" int mllm_hip_alloc(void **dptr, size_t nbytes) {
    if (!dptr) return MLLM_HIP_ERR_ARG;
    MH_CHECK(hipMalloc(dptr, nbytes ? nbytes : 16));
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_free(void *dptr) {
    if (dptr) MH_CHECK(hipFree(dptr));
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_h2d(void *dst, const void *src, size_t nbytes, void *stream) {
    MH_CHECK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice, as_stream(stream)));
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_d2h(void *dst, const void *src, size_t nbytes, void *stream) {
    MH_CHECK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, as_stream(stream)));
    MH_CHECK(hipStreamSynchronize(as_stream(stream)));  // copy_to_host is a sync point (SURVEY §8b Threading)
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_sync(void *stream) {
    MH_CHECK(hipStreamSynchronize(as_stream(stream)));
    return MLLM_HIP_OK;
}

// ---- one in-order stream per backend instance (SURVEY §8b Threading: "one in-order stream; sync only at the end of the outermost forward and in
// copy_to_host"); precedent: the command queue of mllm/backends/opencl/OpenCLBackend.cpp:476-477 --------------------------------------------------
extern "C" int mllm_hip_stream_create(void **stream) {
    if (!stream) return MLLM_HIP_ERR_ARG;
    hipStream_t s;
    MH_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_stream_destroy(void *stream) {
    if (stream) MH_CHECK(hipStreamDestroy(as_stream(stream)));
    return MLLM_HIP_OK;
}

// ---- stream-ordered activation pool: Backend::alloc_device / free_device of per-call Op outputs (the device analogue of the
// reference's MemoryPoolManager, mllm/memory/MemoryPoolManager.hpp:15-214).  hipMallocAsync on the device's default pool with the release
// threshold lifted, so a forward's buffers are recycled by the next forward without touching the driver; a block freed on `stream` may be handed
// out again to work enqueued later on the same stream only. --------------------------------------------------------------------------------------
static std::once_flag g_pool_once;
static hipError_t g_pool_err = hipSuccess;
extern "C" int mllm_hip_pool_alloc(void **dptr, size_t nbytes, void *stream) {
    if (!dptr) return MLLM_HIP_ERR_ARG;
    std::call_once(g_pool_once, [] {
        int dev = 0;
        hipMemPool_t pool;
        uint64_t keep = UINT64_MAX;
        g_pool_err = hipGetDevice(&dev);
        if (g_pool_err == hipSuccess) g_pool_err = hipDeviceGetDefaultMemPool(&pool, dev);
        if (g_pool_err == hipSuccess) g_pool_err = hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    });
    MH_CHECK(g_pool_err);
    MH_CHECK(hipMallocAsync(dptr, nbytes ? nbytes : 16, as_stream(stream)));
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_pool_free(void *dptr, void *stream) {
    if (dptr) MH_CHECK(hipFreeAsync(dptr, as_stream(stream)));
    return MLLM_HIP_OK;
}

// ---- Backend::load_from_file fast path (mllm/Backend.hpp:118; OpenCL precedent OpenCLBackend.cpp:928-980 maps the buffer and freads into it): host bytes
// (the ParamLoader's mmap of the .mllm, or any pageable memory) -> HBM through two pinned staging buffers, the memcpy of chunk i + 1 overlapping the DMA of
// chunk i.  Returns once `src` has been consumed (the caller may unmap it); the last DMAs may still be in flight on `stream`. ---------------------------
namespace {
constexpr size_t kStage = (size_t)32 << 20;
struct Staging {
    void *buf[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int next = 0;
    std::mutex mu;
} g_stage;
}  // namespace
extern "C" int mllm_hip_upload(void *dst, const void *src_host, size_t nbytes, void *stream) {
    if (nbytes == 0) return MLLM_HIP_OK;
    if (!dst || !src_host) return MLLM_HIP_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_stage.mu);
    for (int i = 0; i < 2; ++i)
        if (!g_stage.buf[i]) {
            MH_CHECK(hipHostMalloc(&g_stage.buf[i], kStage, hipHostMallocDefault));
            MH_CHECK(hipEventCreateWithFlags(&g_stage.done[i], hipEventDisableTiming));
        }
    for (size_t off = 0; off < nbytes; off += kStage) {
        const int s = g_stage.next;
        g_stage.next ^= 1;
        const size_t n = nbytes - off < kStage ? nbytes - off : kStage;
        if (g_stage.busy[s]) MH_CHECK(hipEventSynchronize(g_stage.done[s]));      // the DMA that last read this staging buffer has finished
        memcpy(g_stage.buf[s], (const char *)src_host + off, n);
        MH_CHECK(hipMemcpyAsync((char *)dst + off, g_stage.buf[s], n, hipMemcpyHostToDevice, as_stream(stream)));
        MH_CHECK(hipEventRecord(g_stage.done[s], as_stream(stream)));
        g_stage.busy[s] = true;
    }
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_host_register(void *host, size_t nbytes) {
    if (!host || !nbytes) return MLLM_HIP_ERR_ARG;
    MH_CHECK(hipHostRegister(host, nbytes, hipHostRegisterDefault));
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_host_unregister(void *host) {
    if (!host) return MLLM_HIP_ERR_ARG;
    MH_CHECK(hipHostUnregister(host));
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_upload_release(void) {
    std::lock_guard<std::mutex> lock(g_stage.mu);
    for (int i = 0; i < 2; ++i) {
        if (g_stage.busy[i]) MH_CHECK(hipEventSynchronize(g_stage.done[i]));
        if (g_stage.buf[i]) { MH_CHECK(hipHostFree(g_stage.buf[i])); MH_CHECK(hipEventDestroy(g_stage.done[i])); }
        g_stage.buf[i] = nullptr; g_stage.done[i] = nullptr; g_stage.busy[i] = false;
    }
    return MLLM_HIP_OK;
}
