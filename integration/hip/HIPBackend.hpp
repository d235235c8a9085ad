// integration/hip/HIPBackend.hpp -- the reference-side adapter: mllm's Backend / Op registry on top of the C ABI of libmllm_hip.so.
//
// This is what a maintainer of yirongjie/mllm adds as mllm/backends/hip/ (structural twin of mllm/backends/opencl/OpenCLBackend.{hpp,cpp}).  It includes only
// the reference's own headers (mllm/Backend.hpp:32-130, mllm/Op.hpp:20-148, mllm/Tensor.hpp, mllm/Module.hpp) and include/mllm_hip.h, and is compiled in this
// repository ONLY as test infrastructure: oracle/Makefile.ref builds it against /root/reference/mllm where that tree exists (tests/test_integration_build.py),
// to prove that every signature below matches the interface it plugs into.  Nothing reference-built enters the product library.
//
// BackendType: the enum (mllm/Types.hpp:34-39) has no free value at this snapshot and the core gates device tensors on MLLM_OPENCL (mllm/TensorImpl.hpp:88,106).
// Upstream, a maintainer adds MLLM_HIP and widens the two tests; compiled out of tree the adapter takes its slot from MLLM_HIP_BACKEND_TYPE (default: the
// OpenCL slot, the one the core already treats as "on device").
#ifndef MLLM_HIP_BACKEND_HPP
#define MLLM_HIP_BACKEND_HPP

#include <any>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "Backend.hpp"
#include "Op.hpp"
#include "Tensor.hpp"
#include "Types.hpp"

#include "mllm_hip.h"

#ifndef MLLM_HIP_BACKEND_TYPE
#define MLLM_HIP_BACKEND_TYPE MLLM_OPENCL
#endif

namespace mllm {

class Module;
class Layer;

class HIPBackend final : public Backend {
public:
    using OpCreator = std::function<Op *(HIPBackend *, const OpParam &, const std::string &)>;

    explicit HIPBackend(int device = 0);
    ~HIPBackend() override = default;

    // ---- device memory (mllm/Backend.hpp:60-73): DeviceMemory{handle, MEM_TYPE_GENERIC, size_in_bytes} (mllm/TensorImpl.hpp:23-45) ----
    void alloc_device(DeviceMemory &mem, DataType dtype) override;
    void free_device(DeviceMemory &mem) override;
    void copy_from_host(const DeviceMemory &dest, const void *src) override;
    void copy_to_host(void *dest, const DeviceMemory &src) override;
    void convert_fp_data(Tensor *src, Tensor *dest) override;
    bool load_from_file(Tensor *tensor, ParamLoader *loader) override;      // mllm/Backend.hpp:118

    // ---- op registry and call protocol (mllm/Backend.hpp:82-110) ----
    Op *opCreate(const OpParam &op_param, std::string name = "", int threadCount = 4) override;      // nullptr => the framework falls back to CPU (mllm/Layer.hpp:214-218)
    TensorFunction *funcCreate(TensorFuncType type) override;
    std::vector<Tensor> runLayer(Layer *layer, std::vector<Tensor> inputs, int N) override;
    std::vector<Tensor> runOp(Op *op, std::vector<Tensor> input, std::vector<std::string> out_names, bool in_place) override;
    std::vector<Tensor> runForward(Module *module, std::vector<Tensor> inputs, std::vector<std::any> args) override;
    void registerOps() override;
    void registerFuncs() override {}

    void *stream() const { return stream_; }        // one in-order stream; synchronised only at the end of the outermost forward and in copy_to_host
    void sync();
    // device scratch that grows on demand (activation quantisation planes, packed GEMM operands); valid until the next call on the same slot
    void *scratch(int slot, size_t bytes);
    const uint16_t *gelu_lut();
    const uint16_t *quickgelu_lut();

    static void check(int rc, const char *what);

private:
    std::map<OpType, OpCreator> creators_;
    void *stream_ = nullptr;
    void *scratch_[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t scratch_bytes_[4] = {0, 0, 0, 0};
    void *lut_gelu_ = nullptr, *lut_qgelu_ = nullptr;
};

class HIPBackendCreator : public BackendCreator {
public:
    Backend *create(BackendConfig config) override { return new HIPBackend(0); }
};
// InsertBackendCreatorMap(<slot>, HIPBackendCreator) -- precedent registerOpenCLBackendCreator, mllm/backends/opencl/OpenCLBackend.cpp:982-984
void registerHIPBackendCreator();

// raw device pointer of a tensor that lives on this backend
inline void *dptr(const std::shared_ptr<Tensor> &t) { return t->device_memory().handle; }

}  // namespace mllm
#endif
