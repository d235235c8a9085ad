// integration/hip/HIPBackend.hpp -- the reference-side adapter: mllm's Backend / Op registry on top of the C ABI of libmllm_hip.so.
//
// This is what a maintainer of yirongjie/mllm adds as mllm/backends/hip/ (structural twin of mllm/backends/opencl/OpenCLBackend.{hpp,cpp}).  It includes only
// the reference's own headers (mllm/Backend.hpp:32-130, mllm/Op.hpp:20-148, mllm/Tensor.hpp, mllm/Module.hpp) and include/mllm_hip.h, and is compiled in this
// repository ONLY as test infrastructure: oracle/Makefile.ref builds it against /root/reference/mllm where that tree exists, links it with drivers that run the
// reference's own Module graphs on it (oracle/ref_drivers/ref_hip_*.cpp) and the GPU suite executes those (tests/test_gpu_adapter.py).  Nothing reference-built
// enters the product library.
//
// BackendType: the enum (mllm/Types.hpp:34-39) has no free value at this snapshot and the core gates device tensors on MLLM_OPENCL (mllm/TensorImpl.hpp:88,106).
// Upstream, a maintainer adds MLLM_HIP and widens the two tests; compiled out of tree the adapter takes its slot from MLLM_HIP_BACKEND_TYPE (default: the
// OpenCL slot, the one the core already treats as "on device").
//
// Three facts about the frontend shape this file (all verified in the reference's source):
//  1. TensorImpl::to(CPU) calls Backend::free_device on whatever handle the tensor holds, owner or view (mllm/TensorImpl.hpp:171-189), and the destructor frees
//     what it "owns" (:134-144).  Device blocks are therefore reference-counted here: every TensorImpl that points into a block holds one reference (views are
//     handed out through view_of(), which retains), every Op that keeps a raw pointer holds one, and free_device() is "release".
//  2. Layer::run migrates every input to the layer's backend before runOp sees it (mllm/Layer.hpp:159-163), and the models read some of those tensors on the
//     host afterwards (`inputs[1].dataAt<float>` right behind `rot_pos_emb(inputs[1])`, models/qwen2_vl/modeling_qwen2_vl.hpp:179-182).  Ops whose inputs are such
//     host-side scalars say so through HIPOp::host_inputs(); runOp hands those tensors back to the CPU backend before anything else.
//  3. Small host-made tensors (token ids, position ids, grids) reach the device through copy_from_host; the backend keeps a host shadow of every small upload so an
//     Op that needs the values on the host (the rotary tables are libm sinf / cosf, F_WHERE's output shape depends on the data) reads them without a D2H sync.
#ifndef MLLM_HIP_BACKEND_HPP
#define MLLM_HIP_BACKEND_HPP

#include <any>
#include <atomic>
#include <functional>
#include <thread>
#include <tuple>
#include <type_traits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
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
class HIPBackend;

// base of every HIP Op: the backend accessor and the "my inputs are host-side scalars" flag (fact 2 above)
class HIPOp : public Op {
public:
    HIPOp(Backend *bn, const std::string &name) : Op(bn, name) {}
    virtual bool host_inputs() const { return false; }
    // the Op does not write its output on the device: a view of its input (F_VIEW, F_CLIP, F_TRANPOSE, F_FLATTEN, PARAMETER, KVCACHE) or a host upload whose shadow
    // execute() registers itself (F_WHERE).  runOp drops the host shadow of every other Op's output block, which the device has just overwritten.
    virtual bool keeps_shadow() const { return false; }

    // The reference's Op entry points take their tensor lists BY VALUE (mllm/Op.hpp:39,61,88): three calls per Op, two vector copies each -- a malloc and an atomic increment
    // per tensor, six times per Op, on the thread the whole frontend runs on.  The adapter's Ops implement the by-reference forms below; the by-value ones forward to them (a caller
    // that only knows `Op` still works), and HIPBackend::runOp calls the by-reference forms directly.
    using TL = const std::vector<std::shared_ptr<Tensor>> &;
    virtual ErrorCode reshape_(TL inputs, TL outputs) = 0;
    virtual ErrorCode setUp_(TL inputs, TL outputs) = 0;
    virtual ErrorCode execute_(TL, TL) { return MLLM_NO_ERROR; }      // views: nothing to launch (what Op::execute does, mllm/Op.hpp:88-96)
    ErrorCode reshape(std::vector<std::shared_ptr<Tensor>> inputs, std::vector<std::shared_ptr<Tensor>> outputs) final { return reshape_(inputs, outputs); }
    ErrorCode setUp(std::vector<std::shared_ptr<Tensor>> inputs, std::vector<std::shared_ptr<Tensor>> outputs) final { return setUp_(inputs, outputs); }
    ErrorCode execute(std::vector<std::shared_ptr<Tensor>> inputs, std::vector<std::shared_ptr<Tensor>> outputs) final { return execute_(inputs, outputs); }

protected:
    HIPBackend *hb() const;
};

// a Q4_0 table that two Ops share (EMBEDDING and the PARAMETER of the tied lm_head read the same `*.embed_tokens.weight`): the file's 18-byte blocks (`raw`, what a
// view of the Parameter points at) and the nibble / scale planes the kernels read (mllm_hip_repack_q40)
struct HIPQ40Table {
    Tensor raw;
    void *qs = nullptr, *d = nullptr;
    int rows = 0, cols = 0;
};

class HIPBackend final : public Backend {
public:
    using OpCreator = std::function<Op *(HIPBackend *, const OpParam &, const std::string &)>;

    explicit HIPBackend(int device = 0);
    ~HIPBackend() override;

    // ---- device memory (mllm/Backend.hpp:60-73): DeviceMemory{handle, MEM_TYPE_GENERIC, size_in_bytes} (mllm/TensorImpl.hpp:23-45) on a stream-ordered pool ----
    void alloc_device(DeviceMemory &mem, DataType dtype) override;
    void free_device(DeviceMemory &mem) override;      // releases ONE reference to the block `mem.handle` points into
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

    // ---- deferred launches ----
    // The frontend walks the model Op by Op on the caller's thread; what an Op's execute() costs there is dominated by the HIP launch itself (about 2.5 us of host time per
    // kernel, 480 kernels per decode step of the 2 B model -- more than the frontend's own work for the step).  So execute() does not launch: it hands the C-ABI call, with
    // its arguments already evaluated, to ONE worker thread that issues the calls in program order on the backend's stream, while the caller's thread is already inside the
    // next Op.  Everything that must see the device in order goes through the same queue (stream-ordered frees, small uploads); everything that reads results on the host
    // (copy_to_host, host_floats' D2H, sync) drains it first.  A failing call is remembered on the worker (with mllm_hip_last_error's text, which is thread-local there)
    // and thrown from the next drain().
    template <typename Fn, typename... A>
    void defer(const char *what, Fn fn, A... args) {
        if (!lazy_.empty() && !flushing_) flush_lazy();      // program order: what the window still holds goes first
        // the call travels as plain bytes inside its ring slot (function pointer + the argument tuple): no allocation on this thread, nothing to free on the worker's
        if (inline_launch_) { ++deferred_calls_; check(fn(args...), what); return; }      // MLLM_HIP_INLINE_LAUNCH=1: the caller's thread launches (A/B measurements, debugging)
        using Tup = std::tuple<A...>;
        static_assert(sizeof(Tup) <= sizeof(Deferred::args) && std::is_trivially_destructible<Tup>::value, "deferred C-ABI calls carry pointers and integers only");
        ++deferred_calls_;
        Deferred &d = claim_slot();
        new (d.args) Tup(args...);
        d.fn = reinterpret_cast<void (*)()>(fn);
        d.thunk = [](Deferred &x) -> int { return std::apply(reinterpret_cast<Fn>(x.fn), *reinterpret_cast<Tup *>(x.args)); };
        d.what = what;
        publish_slot();
    }
    void enqueue(std::function<int()> call, const char *what);      // the rare calls that own memory (small uploads carry their bytes)
    void drain();

    // ---- the lazy window: fusion across the Ops the frontend issues one at a time ----
    // A decode layer reaches this backend as 18 Ops on ONE activation row (RMSNORM, LINEAR q / k / v, ROPE x 2, KVCACHE x 2, F_FA2, LINEAR o, F_TTADD, RMSNORM, LINEAR gate, SILU,
    // LINEAR up, F_TTMUL, LINEAR down, F_TTADD), 17 launches of 2-5 us each -- the device, not the host, bounds the Op-by-Op path.  The Ops of that run do not launch: they describe
    // themselves (LazyOp) and the window emits, in program order, the longest runs the library has one launch for (mllm_hip_row_fused_launch, mllm_hip_rope2_store2) and everything
    // else as the Op's own call.  The window is emitted as soon as the next Op cannot extend the run it holds, or closes it (window_extends), so the device never waits for more than
    // the six Ops of one run.  A fused launch still writes EVERY Op's output tensor with the value that Op's own kernel computes, so nothing the frontend can observe changes;
    // runs are contiguous, so the order of all device work is the program's.  Anything that is not a LazyOp (another Op's defer, an upload, a drain) flushes the window first.
    // MLLM_HIP_NO_FUSE=1 emits every Op on its own (A/B measurements).
    struct LazyOp {
        enum Kind : int { NORM, LINEAR, SILU, MUL, ADD, ROPE, KVSTORE, FA2, NORM_M, LINEAR_M } kind;      // _M: the M >= 16 (prefill) forms: RMSNORM / LAYERNORM rows, packed-GEMM Linears
        const float *a = nullptr, *b = nullptr;      // inputs (b: second operand of ADD / MUL)
        float *out = nullptr;
        int64_t n = 0;                               // elements (SILU / MUL / ADD), row width (NORM, KVSTORE), out_features (LINEAR)
        const float *w = nullptr;                    // NORM weights / LINEAR bias
        float eps = 0;
        const void *W = nullptr;                     // LINEAR: raw Q4_K rows
        int K = 0;
        void *ws = nullptr;                          // LINEAR: the workspace of its own call
        const float *sin = nullptr, *cos = nullptr;  // ROPE
        int ld_tab = 0, S = 0, H = 0, D = 0;
        uint16_t *dst16 = nullptr;                   // KVSTORE: the slab rows to append to
        int M = 1;                                   // NORM_M / LINEAR_M: rows
        const void *Wpacked = nullptr;               // LINEAR_M: the packed GEMM operand (mllm_hip_q4k_prepack); ws = the packed-activation scratch of its own call
        int layer_norm = 0, unit_offset = 0;         // NORM_M: LAYERNORM (b = its bias, may be null) / RMSNorm with add_unit_offset
        const void *kp = nullptr, *vp = nullptr;     // FA2 (one query row): K / V views [Sk][Hkv * D]; a = q, H = Hq
        int Sk = 0, Hkv = 0, causal = 0, kvdt = 0;
    };
    void lazy(const LazyOp &op);
    void flush_lazy();
    long fused_launches() const { return fused_launches_; }
    long fused_ops() const { return fused_ops_; }
    long deferred_calls() const { return deferred_calls_; }      // C-ABI calls handed to the worker so far (what the window's unit test counts)

    // ---- reference-counted device blocks (fact 1) ----
    void *dev_alloc(size_t bytes);                   // Op-owned memory (weights' repacks, KV slabs, tables): one reference, dropped by dev_release
    void dev_release(void *p);
    void retain(void *p);                            // one more holder of the block p points into
    // makes `view` a tensor on this backend that points at `handle` (somewhere inside a live block) and holds its own reference
    void view_of(const std::shared_ptr<Tensor> &view, void *handle, size_t bytes);

    // device scratch that grows on demand (activation quantisation planes, packed GEMM operands); valid until the next call on the same slot
    void *scratch(int slot, size_t bytes);
    const uint16_t *gelu_lut();
    const uint16_t *quickgelu_lut();
    // host bytes -> device through the library's pinned staging (returns when `src` may be reused)
    void upload(void *dst, const void *src, size_t bytes);

    // ---- host shadows of small device tensors (fact 3) ----
    // the fp32 values of a small tensor that lives on this backend: the shadow of its upload, else one D2H (which synchronises)
    const std::vector<float> &host_floats(const std::shared_ptr<Tensor> &t);
    void remember_host(void *handle, const float *v, size_t n);

    // ---- state shared between Ops ----
    std::shared_ptr<HIPQ40Table> q40_table(AbstructLoader &loader, const std::string &name, int rows, int cols);
    std::shared_ptr<HIPQ40Table> q40_table_at(const void *raw_handle) const;
    // sin / cos device tables of the current M-RoPE position ids, built once per forward and shared by every MULTIMODALROPE Op (2 per layer)
    struct RopeTables { const float *sin = nullptr, *cos = nullptr; int S = 0, half = 0; };
    RopeTables mrope_tables(const std::shared_ptr<Tensor> &position_ids, float theta, int D, const std::vector<int> &section);
    // CPURoPE's static half-split table [2][max_pos][D] (sines, then cosines) on the device: ONE per (theta, D, max_pos) for the whole backend -- every ROPE Op of a model asks for
    // the same one (two per layer), and building it is max_pos * D libm calls on the caller's thread
    const float *rope_hf_tables(float theta, int D, int max_pos);
    // sin / cos of a VISIONROPE angle table, keyed by the table's device handle (set by the VISIONROPE Op that made it)
    void set_vision_tables(void *angles_handle, const float *sin, const float *cos, int N, int half);
    bool vision_tables(const void *angles_handle, RopeTables *out) const;

    // ---- bookkeeping the drivers print: which creators refused (=> CPU fallback), how many Ops ran here ----
    const std::vector<std::pair<int, std::string>> &refused() const { return refused_; }
    long ops_run() const { return ops_run_; }
    size_t live_blocks() const { return blocks_.size(); }

    static void check(int rc, const char *what);

private:
    struct Block { size_t size; int refs; bool pooled; };
    std::map<uintptr_t, Block>::iterator block_of(const void *p);
    void release(const void *p);

    std::map<OpType, OpCreator> creators_;
    void *stream_ = nullptr;
    std::map<uintptr_t, Block> blocks_;
    // activation blocks the frontend has released, kept by size: a decode step asks for the same few hundred sizes again, in the same order, on the same stream (reuse is
    // stream-ordered like the pool's own), so the step's allocations cost a map lookup instead of a hipMallocAsync / hipFreeAsync pair each
    std::unordered_map<size_t, std::vector<void *>> idle_;
    size_t idle_bytes_ = 0;
    static constexpr size_t kIdleLimit = (size_t)1 << 30;
    void drain_idle();
    void *scratch_[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t scratch_bytes_[6] = {0, 0, 0, 0, 0, 0};
    void *lut_gelu_ = nullptr, *lut_qgelu_ = nullptr;
    std::unordered_map<const void *, std::vector<float>> shadows_;
    std::map<std::string, std::shared_ptr<HIPQ40Table>> q40_by_name_;
    std::unordered_map<const void *, std::shared_ptr<HIPQ40Table>> q40_by_handle_;
    struct MropeKey { const void *pos = nullptr; long serial = -1; float theta = 0; int D = 0; std::vector<int> section; } mrope_key_;
    void *mrope_dev_ = nullptr;
    size_t mrope_bytes_ = 0;
    RopeTables mrope_;
    std::unordered_map<const void *, RopeTables> vision_;
    std::map<std::tuple<float, int, int>, float *> rope_hf_;
    std::unordered_map<std::string, std::pair<void *, size_t>> maps_;      // .mllm path -> read-only mmap (load_from_file)
    // single-producer / single-consumer ring of deferred calls
    struct Deferred {
        int (*thunk)(Deferred &) = nullptr;      // unpacks args and calls fn; nullptr: `call` below is the call
        void (*fn)() = nullptr;
        const char *what = "";
        alignas(16) unsigned char args[256];
        std::function<int()> call;
    };
    Deferred &claim_slot();
    void publish_slot();
    static constexpr size_t kRing = 4096;
    std::vector<Deferred> ring_;
    alignas(64) std::atomic<size_t> head_{0};      // the producer (the frontend's thread) writes at head_ ...
    alignas(64) std::atomic<size_t> tail_{0};      // ... the worker consumes at tail_ (own cache lines: the two threads do not bounce one line)
    alignas(64) size_t tail_seen_ = 0;             // producer's cached view of tail_
    std::atomic<bool> stop_{false}, failed_{false};
    bool inline_launch_ = false, no_fuse_ = false, flushing_ = false;
    std::vector<LazyOp> lazy_;
    long fused_launches_ = 0, fused_ops_ = 0, deferred_calls_ = 0;
    bool window_extends(LazyOp::Kind k) const;
    void emit_single(const LazyOp &op);
    size_t emit_group(size_t i);
    std::thread worker_;
    std::string failure_;
    void worker_loop();
    long forward_serial_ = 0, ops_run_ = 0;
    const char *dump_dir_ = nullptr;      // MLLM_HIP_DUMP_DIR: bring-up dumps of every Op's outputs (HIPBackend.cpp: dump_outputs)
    void dump_outputs(Op *op, const std::vector<std::shared_ptr<Tensor>> &outs);
    std::vector<std::pair<int, std::string>> refused_;
};

inline HIPBackend *HIPOp::hb() const { return static_cast<HIPBackend *>(backend_); }

class HIPBackendCreator : public BackendCreator {
public:
    Backend *create(BackendConfig config) override { return new HIPBackend(0); }
};
// InsertBackendCreatorMap(<slot>, HIPBackendCreator) -- precedent registerOpenCLBackendCreator, mllm/backends/opencl/OpenCLBackend.cpp:982-984
void registerHIPBackendCreator();
// what Module::initBackend's new `case MLLM_HIP` does (mllm/Module.hpp:158-188): put the singleton into Backend::global_backends; the out-of-tree drivers call it
// before model.to(<slot>) because initBackend's switch cannot be extended from outside
HIPBackend *installHIPBackend(int device = 0);

// raw device pointer of a tensor that lives on this backend
inline void *dptr(const std::shared_ptr<Tensor> &t) { return t->device_memory().handle; }

}  // namespace mllm
#endif
