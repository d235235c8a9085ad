// mllm_amd/csrc/kernels_sample.hip -- SURVEY N2: candidate selection of the sampled generation methods on the device, and the visual-token
// exchange of the sharded vision prefill (RCCL) behind the C ABI.
//
//   _LlmTextGenerateToppSamplingMethod::generate (mllm/Generate.cpp:93-142) sorts the whole (probability, index) row descending (std::sort :99) and keeps
//   the prefix whose running sum reaches p: here the sort is one device radix sort (keys descending; the sort is stable and the indices start ascending, so
//   equal probabilities keep ascending index order -- std::sort leaves ties unspecified), and only the prefix crosses PCIe.
//   The draw of both sampling methods, _sample_element (Generate.hpp:38-44), is a std::discrete_distribution over the float probabilities: the host helper
//   below is its inverse-CDF form on a caller-supplied uniform number (the reference seeds from std::random_device and cannot be replayed).
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>

#include "common.h"

using namespace mllm_hip;

namespace {
__global__ void iota_kernel(int *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
size_t sort_temp_bytes(int n) {
    size_t tb = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb, (const float *)nullptr, (float *)nullptr, (const int *)nullptr, (int *)nullptr, n);
    return tb;
}
}  // namespace

extern "C" size_t mllm_hip_sort_desc_workspace_bytes(int n) { return n <= 0 ? 0 : align256((size_t)n * 4) + align256(sort_temp_bytes(n)); }

extern "C" int mllm_hip_sort_desc(const float *x, int n, float *val_sorted, int *idx_sorted, void *workspace, size_t workspace_bytes, void *stream) {
    if (n <= 0 || !x || !val_sorted || !idx_sorted) return MLLM_HIP_ERR_ARG;
    if (!workspace || workspace_bytes < mllm_hip_sort_desc_workspace_bytes(n)) return MLLM_HIP_ERR_ARG;
    hipStream_t st = as_stream(stream);
    int *iota = (int *)workspace;
    void *temp = (uint8_t *)workspace + align256((size_t)n * 4);
    size_t tb = workspace_bytes - align256((size_t)n * 4);
    hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, st, iota, n);
    int rc = MH_LAUNCH_OK("iota");
    if (rc) return rc;
    MH_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(temp, tb, x, val_sorted, (const int *)iota, idx_sorted, n, 0, 32, st));
    return MLLM_HIP_OK;
}

extern "C" int mllm_hip_sample_index_host(const float *probs, int k, float u01) {
    if (!probs || k <= 0) return 0;
    double sum = 0.0;
    for (int i = 0; i < k; ++i) sum += probs[i];
    double acc = 0.0;
    for (int i = 0; i < k; ++i) {
        acc += probs[i] / sum;
        if ((double)u01 < acc) return i;
    }
    return k - 1;
}

// ---- the one exchange step of the sharded vision prefill ------------------------------------------------------------------------------
extern "C" int mllm_hip_comm_unique_id(void *id128) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!id128) return MLLM_HIP_ERR_ARG;
    return ncclGetUniqueId((ncclUniqueId *)id128) == ncclSuccess ? MLLM_HIP_OK : MLLM_HIP_ERR_HIP;
}
extern "C" int mllm_hip_comm_create(const void *id128, int world, int rank, void **comm) {
    if (!id128 || !comm || world <= 0 || rank < 0 || rank >= world) return MLLM_HIP_ERR_ARG;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c;
    if (ncclCommInitRank(&c, world, id, rank) != ncclSuccess) return MLLM_HIP_ERR_HIP;
    *comm = (void *)c;
    return MLLM_HIP_OK;
}
extern "C" int mllm_hip_comm_destroy(void *comm) {
    if (!comm) return MLLM_HIP_ERR_ARG;
    return ncclCommDestroy((ncclComm_t)comm) == ncclSuccess ? MLLM_HIP_OK : MLLM_HIP_ERR_HIP;
}
extern "C" int mllm_hip_all_gather_rows(void *comm, const float *local_dev, float *all_dev, int64_t rows_per_rank, int cols, void *stream) {
    if (!comm || !local_dev || !all_dev || rows_per_rank < 0 || cols <= 0) return MLLM_HIP_ERR_ARG;
    if (rows_per_rank == 0) return MLLM_HIP_OK;
    return ncclAllGather(local_dev, all_dev, (size_t)rows_per_rank * cols, ncclFloat, (ncclComm_t)comm, as_stream(stream)) == ncclSuccess ? MLLM_HIP_OK : MLLM_HIP_ERR_HIP;
}
