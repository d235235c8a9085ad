// mllm_amd/csrc/runtime.hip -- device context, memory and error plumbing of libmllm_hip.
// Replaces Backend::{alloc_device,free_device,copy_from_host,copy_to_host} (mllm/Backend.hpp:60-73); structural precedent
// mllm/backends/opencl/OpenCLBackend.cpp:670-787 (the only discrete-device backend of the reference).
#include <cstdio>
#include <cstring>
#include <mutex>

#include "common.h"

namespace mllm_hip {
static thread_local char g_err[512] = "";

void set_error(const char *what, hipError_t e, const char *file, int line) {
    snprintf(g_err, sizeof(g_err), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
}
int check_launch(const char *what, const char *file, int line) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(what, e, file, line);
        return MLLM_HIP_ERR_HIP;
    }
    return MLLM_HIP_OK;
}
}  // namespace mllm_hip

using namespace mllm_hip;

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
