// integration/hip/HIPBackend.cpp -- see HIPBackend.hpp.  Call protocol after mllm/backends/opencl/OpenCLBackend.cpp:990-1097 and mllm/backends/cpu/CPUBackend.cpp:314-405.
#include "HIPBackend.hpp"

#include <cstdio>
#include <cstring>

#include "Module.hpp"
#include "ParamLoader.hpp"
#include "Timing.hpp"
#include "memory/SystemMemoryManager.hpp"

namespace mllm {

void HIPBackend::check(int rc, const char *what) {
    if (rc != MLLM_HIP_OK) throw std::runtime_error(std::string("mllm_hip: ") + what + " failed (" + std::to_string(rc) + "): " + mllm_hip_last_error());
}

HIPBackend::HIPBackend(int device) {
    type_ = MLLM_HIP_BACKEND_TYPE;
    mem_manager_ = std::make_shared<SystemMemoryManager>();       // host-side alloc/free of Backend (mllm/Backend.hpp:48-58)
    check(mllm_hip_init(device), "mllm_hip_init");
    registerOps();
    registerFuncs();
}

void HIPBackend::alloc_device(DeviceMemory &mem, DataType) {
    mem.type = MEM_TYPE_GENERIC;
    check(mllm_hip_alloc(&mem.handle, mem.size_in_bytes), "mllm_hip_alloc");
}
void HIPBackend::free_device(DeviceMemory &mem) {
    if (mem.handle) check(mllm_hip_free(mem.handle), "mllm_hip_free");
    mem.handle = nullptr;
}
void HIPBackend::copy_from_host(const DeviceMemory &dest, const void *src) { check(mllm_hip_h2d(dest.handle, src, dest.size_in_bytes, stream_), "mllm_hip_h2d"); }
void HIPBackend::copy_to_host(void *dest, const DeviceMemory &src) {
    check(mllm_hip_d2h(dest, src.handle, src.size_in_bytes, stream_), "mllm_hip_d2h");
    sync();
}
void HIPBackend::convert_fp_data(Tensor *, Tensor *) {
    // the HIP path keeps activations fp32 (the reference CPU backend's arithmetic type): nothing to convert
}
void HIPBackend::sync() { check(mllm_hip_sync(stream_), "mllm_hip_sync"); }

void *HIPBackend::scratch(int slot, size_t bytes) {
    if (bytes > scratch_bytes_[slot]) {
        sync();
        if (scratch_[slot]) check(mllm_hip_free(scratch_[slot]), "mllm_hip_free");
        check(mllm_hip_alloc(&scratch_[slot], bytes), "mllm_hip_alloc");
        scratch_bytes_[slot] = bytes;
    }
    return scratch_[slot];
}
static void upload_luts(HIPBackend *b, void **g, void **q) {
    std::vector<uint16_t> hg(65536), hq(65536);
    HIPBackend::check(mllm_hip_build_act_luts(hg.data(), hq.data()), "mllm_hip_build_act_luts");
    HIPBackend::check(mllm_hip_alloc(g, 65536 * 2), "mllm_hip_alloc");
    HIPBackend::check(mllm_hip_alloc(q, 65536 * 2), "mllm_hip_alloc");
    HIPBackend::check(mllm_hip_h2d(*g, hg.data(), 65536 * 2, b->stream()), "mllm_hip_h2d");
    HIPBackend::check(mllm_hip_h2d(*q, hq.data(), 65536 * 2, b->stream()), "mllm_hip_h2d");
    b->sync();
}
const uint16_t *HIPBackend::gelu_lut() { if (!lut_gelu_) upload_luts(this, &lut_gelu_, &lut_qgelu_); return (const uint16_t *)lut_gelu_; }
const uint16_t *HIPBackend::quickgelu_lut() { if (!lut_qgelu_) upload_luts(this, &lut_gelu_, &lut_qgelu_); return (const uint16_t *)lut_qgelu_; }

// Weights go file -> pinned-free host staging -> HBM without a host tensor staying alive (precedent OpenCLBackend.cpp:928-980: map, fread, unmap)
bool HIPBackend::load_from_file(Tensor *tensor, ParamLoader *loader) {
    ParamMetadata md = loader->getParamMetadata(tensor->name());
    if (md.size == 0) return true;
    if (tensor->device_memory().handle == nullptr) return false;      // load() before alloc(): let ParamLoader take its default path
    FILE *fp = loader->getInputStream();
    if (!fp) return false;
    std::vector<uint8_t> buf((size_t)md.size);
    fseek(fp, (long)md.offset, SEEK_SET);
    if (fread(buf.data(), 1, buf.size(), fp) != buf.size()) return false;
    check(mllm_hip_h2d(tensor->device_memory().handle, buf.data(), buf.size(), stream_), "mllm_hip_h2d");
    sync();
    tensor->forceResetHostPointer(nullptr);
    return true;
}

Op *HIPBackend::opCreate(const OpParam &op_param, std::string name, int) {
    auto it = creators_.find((OpType)(int)op_param.at("type"));
    return it == creators_.end() ? nullptr : it->second(this, op_param, name);
}
TensorFunction *HIPBackend::funcCreate(TensorFuncType) { throw std::runtime_error("HIPBackend: the legacy TensorFunction path is not used (OpenCLBackend throws too)"); }
std::vector<Tensor> HIPBackend::runLayer(Layer *, std::vector<Tensor>, int) { throw std::runtime_error("runLayer is the QNN path"); }

std::vector<Tensor> HIPBackend::runOp(Op *op, std::vector<Tensor> inputs, std::vector<std::string> out_names, bool in_place) {
    Module *module = inputs.empty() ? Module::llm_model_ptr : inputs[0].module();
    static map<string, shared_ptr<Tensor>> empty_activation_tensors;
    map<string, shared_ptr<Tensor>> &activation_tensors = module ? module->activation_tensors : empty_activation_tensors;
    if (module && module->doTrace) {      // trace / load pass: named activation tensors, setUp only (CPUBackend.cpp:318-349)
        if (module->tracedFlag) {
            vector<Tensor> results = {};
            for (auto &name : out_names) results.push_back(*activation_tensors[name]);
            return results;
        }
        for (auto &input : inputs) {
            if (input.shouldInGraphs() && activation_tensors.find(input.name()) == activation_tensors.end()) {
                activation_tensors[input.name()] = std::make_shared<Tensor>(op->backend());
                activation_tensors[input.name()]->setName(input.name());
                activation_tensors[input.name()]->setModule(module);
            }
        }
        for (const auto &out_name : out_names) {
            if (activation_tensors.find(out_name) == activation_tensors.end()) {
                activation_tensors[out_name] = std::make_shared<Tensor>(op->backend());
                activation_tensors[out_name]->setName(out_name);
                activation_tensors[out_name]->setModule(module);
            }
        }
        vector<shared_ptr<Tensor>> inPtrs;
        for (auto &input : inputs)
            inPtrs.push_back(input.shouldInGraphs() ? activation_tensors[input.name()] : std::shared_ptr<Tensor>(&input, [](Tensor *) {}));
        vector<shared_ptr<Tensor>> outPtrs = {};
        for (auto &name : out_names) outPtrs.push_back(activation_tensors[name]);
        op->setUp(inPtrs, outPtrs);
        vector<Tensor> results = {};
        for (auto &name : out_names) results.push_back(*activation_tensors[name]);
        return results;
    }
    // run pass: non-owning input handles, fresh output shells named out-<opname>, reshape -> setUp (allocates on the device) -> execute
    vector<shared_ptr<Tensor>> input_tensors;
    for (auto &input : inputs) input_tensors.push_back(std::shared_ptr<Tensor>(&input, [](Tensor *) {}));
    vector<shared_ptr<Tensor>> out_tensors;
    if (!in_place) {
        for (const auto &out_name : out_names) {
            auto out_tensor = std::make_shared<Tensor>(op->backend());
            out_tensor->setName(out_name);
            out_tensors.push_back(out_tensor);
        }
    } else {
        for (size_t i = 0; i < input_tensors.size() && i < out_names.size(); ++i) {
            input_tensors[i]->setName(out_names[i]);
            out_tensors.push_back(input_tensors[i]);
        }
    }
    op->reshape(input_tensors, out_tensors);
    op->setUp(input_tensors, out_tensors);
    op->execute(input_tensors, out_tensors);
    vector<Tensor> results;
    for (const auto &out_tensor : out_tensors) results.push_back(*out_tensor);
    return results;
}

std::vector<Tensor> HIPBackend::runForward(Module *module, std::vector<Tensor> inputs, std::vector<std::any> args) {
    if (Module::llm_model_ptr && (Module::llm_model_ptr->doLoad || Module::llm_model_ptr->doChangeBn)) return module->Forward(inputs, args);
    uint64_t time_start = 0;
    const bool outermost = inputs[0].ttype() == TensorType::INPUT_TENSOR;
    if (outermost) {
        for (auto &input : inputs) {
            input.setModule(module);
            input.setTtype(TensorType::NORMAL_TENSOR);
        }
        Module::llm_model_ptr = module;
        if (module->prefilling_token_size_ == 0) module->prefilling_token_size_ = inputs[0].sequence() * inputs[0].batch();
        else if (module->decoding_token_size_ == 0) module->decoding_token_size_ = inputs[0].sequence() * inputs[0].batch();
        time_start = mllm_time_us();
    }
    auto output = module->Forward(inputs, args);
    if (outermost) {
        sync();      // the launchers never synchronise: Module::profiling()'s numbers (mllm/Module.cpp:25-61) need the device to be done
        module->inference_times_.push_back((mllm_time_us() - time_start) / 1000.0F);
    }
    return output;
}

class HIPBackendCreatorReg {};
void registerHIPBackendCreator() { InsertBackendCreatorMap(MLLM_HIP_BACKEND_TYPE, std::make_shared<HIPBackendCreator>()); }

}  // namespace mllm
