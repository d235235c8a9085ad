// integration/hip/HIPBackend.cpp -- see HIPBackend.hpp.  Call protocol after mllm/backends/opencl/OpenCLBackend.cpp:990-1097 and mllm/backends/cpu/CPUBackend.cpp:314-405.
#include "HIPBackend.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
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
    check(mllm_hip_stream_create(&stream_), "mllm_hip_stream_create");
    registerOps();
    registerFuncs();
    dump_dir_ = getenv("MLLM_HIP_DUMP_DIR");
    inline_launch_ = getenv("MLLM_HIP_INLINE_LAUNCH") != nullptr;
    no_fuse_ = getenv("MLLM_HIP_NO_FUSE") != nullptr;
    lazy_.reserve(64);
    ring_.resize(kRing);
    worker_ = std::thread([this] { worker_loop(); });
}

// ---- deferred launches (see HIPBackend.hpp) ---------------------------------------------------------------------------------------------------------------------
void HIPBackend::worker_loop() {
    int idle = 0;
    for (;;) {
        const size_t t = tail_.load(std::memory_order_relaxed);
        if (t == head_.load(std::memory_order_acquire)) {
            if (stop_.load(std::memory_order_acquire)) return;
            // inside a forward the next call is a microsecond away: spin.  After about half a millisecond without work (between generations, at the prompt) fall back to
            // polling every 100 us -- the producer never signals, so publishing a call costs it one plain store
            if (++idle < 12000) { __builtin_ia32_pause(); continue; }
            std::this_thread::sleep_for(std::chrono::microseconds(100));
            continue;
        }
        idle = 0;
        Deferred &d = ring_[t % kRing];
        if (!failed_.load(std::memory_order_relaxed)) {      // after a failure the rest of the queue is dropped: the caller's next drain() throws
            int rc = MLLM_HIP_OK;
            std::string msg;
            try { rc = d.thunk ? d.thunk(d) : d.call(); } catch (const std::exception &e) { rc = -1; msg = e.what(); }
            if (rc != MLLM_HIP_OK) {
                failure_ = std::string("mllm_hip: ") + d.what + " failed (" + std::to_string(rc) + "): " + (msg.empty() ? mllm_hip_last_error() : msg.c_str());
                failed_.store(true, std::memory_order_release);
            }
        }
        if (!d.thunk) d.call = nullptr;
        tail_.store(t + 1, std::memory_order_release);
    }
}
HIPBackend::Deferred &HIPBackend::claim_slot() {
    const size_t h = head_.load(std::memory_order_relaxed);
    if (h - tail_seen_ >= kRing)      // ring full by the cached view: look at the worker's real position (it is behind by 4096 calls only if something is stuck)
        while (h - (tail_seen_ = tail_.load(std::memory_order_acquire)) >= kRing) __builtin_ia32_pause();
    return ring_[h % kRing];
}
void HIPBackend::publish_slot() {
    head_.store(head_.load(std::memory_order_relaxed) + 1, std::memory_order_release);
}
void HIPBackend::enqueue(std::function<int()> call, const char *what) {
    if (!lazy_.empty() && !flushing_) flush_lazy();
    if (inline_launch_) { check(call(), what); return; }
    Deferred &d = claim_slot();
    d.thunk = nullptr;
    d.call = std::move(call);
    d.what = what;
    publish_slot();
}
void HIPBackend::drain() {
    if (!lazy_.empty() && !flushing_) flush_lazy();
    const size_t h = head_.load(std::memory_order_relaxed);
    while (tail_.load(std::memory_order_acquire) != h) __builtin_ia32_pause();
    if (failed_.load(std::memory_order_acquire)) {
        const std::string msg = failure_;
        failed_.store(false, std::memory_order_release);
        throw std::runtime_error(msg);
    }
}

// ---- reference-counted blocks ---------------------------------------------------------------------------------------------------------------------------------
std::map<uintptr_t, HIPBackend::Block>::iterator HIPBackend::block_of(const void *p) {
    auto it = blocks_.upper_bound((uintptr_t)p);
    if (it == blocks_.begin()) return blocks_.end();
    --it;
    return (uintptr_t)p < it->first + it->second.size ? it : blocks_.end();
}
void HIPBackend::retain(void *p) {
    auto it = block_of(p);
    if (it == blocks_.end()) throw std::runtime_error("HIPBackend::retain: pointer is not inside a live device block");
    ++it->second.refs;
}
void HIPBackend::release(const void *p) {
    auto it = block_of(p);
    if (it == blocks_.end()) return;      // not ours (or already gone): nothing to do
    if (--it->second.refs > 0) return;
    const uintptr_t lo = it->first, hi = lo + it->second.size;
    auto inside = [&](const void *q) { return (uintptr_t)q >= lo && (uintptr_t)q < hi; };
    for (auto s = shadows_.begin(); s != shadows_.end();) s = inside(s->first) ? shadows_.erase(s) : std::next(s);
    for (auto s = vision_.begin(); s != vision_.end();) s = inside(s->first) ? vision_.erase(s) : std::next(s);
    if (mrope_key_.pos && inside(mrope_key_.pos)) mrope_key_ = MropeKey();
    if (it->second.pooled) {
        if (idle_bytes_ + it->second.size <= kIdleLimit) { idle_[it->second.size].push_back((void *)lo); idle_bytes_ += it->second.size; }
        else defer("mllm_hip_pool_free", mllm_hip_pool_free, (void *)lo, stream_);      // stream-ordered behind the launches that still use the block
    } else { drain(); check(mllm_hip_free((void *)lo), "mllm_hip_free"); }
    blocks_.erase(it);
}
void HIPBackend::drain_idle() {
    try { drain(); } catch (...) {}
    for (auto &kv : idle_)
        for (void *p : kv.second) (void)mllm_hip_pool_free(p, stream_);
    idle_.clear();
    idle_bytes_ = 0;
}
// ---- the lazy window (see HIPBackend.hpp) ------------------------------------------------------------------------------------------------------------------------
// The runs one launch covers, as strings over the Ops' kinds (N RMSNORM, L LINEAR, S SILU, M F_TTMUL, A F_TTADD, R ROPE, K KVCACHE append, F F_FA2).  The window only ever holds a
// prefix of one of them: an Op that cannot extend the prefix sends it to the device first, an Op that completes a run nothing can extend sends the run at once.
// (lower case: the prefill forms -- n = RMSNORM / LAYERNORM on M >= 16 rows, l = a Linear on them: the norm writes the packed Q8_K operand once and every Linear behind it
// reads that, instead of one quantiser pass per Linear)
static const char *const kRuns[] = {"NLLL", "NLSLM", "ANLLL", "ANLSLM", "LA", "LLL", "RRKKF", "nlll", "nlSlM", "lll"};
static char kind_char(HIPBackend::LazyOp::Kind k) { return "NLSMARKFnl"[(int)k]; }
bool HIPBackend::window_extends(LazyOp::Kind k) const {
    char w[16];
    size_t n = 0;
    for (const LazyOp &o : lazy_) w[n++] = kind_char(o.kind);
    w[n++] = kind_char(k);
    for (const char *run : kRuns)
        if (strlen(run) >= n && strncmp(run, w, n) == 0) return true;
    return false;
}
void HIPBackend::lazy(const LazyOp &op) {
    if (no_fuse_) { emit_single(op); return; }
    if (!lazy_.empty() && (lazy_.size() >= 8 || !window_extends(op.kind))) flush_lazy();
    lazy_.push_back(op);
    char w[16];
    size_t n = 0;
    for (const LazyOp &o : lazy_) w[n++] = kind_char(o.kind);
    bool open = false;      // can anything still follow?
    for (const char *run : kRuns)
        if (strlen(run) > n && strncmp(run, w, n) == 0) open = true;
    if (!open) flush_lazy();
}
void HIPBackend::flush_lazy() {
    if (flushing_) return;
    flushing_ = true;
    try {
        for (size_t i = 0; i < lazy_.size();) i += emit_group(i);
    } catch (...) { lazy_.clear(); flushing_ = false; throw; }
    lazy_.clear();
    flushing_ = false;
}
// the Op's own call, exactly as its execute() used to issue it
void HIPBackend::emit_single(const LazyOp &o) {
    const bool was = flushing_;
    flushing_ = true;      // defer() must not come back into the window
    switch (o.kind) {
    case LazyOp::NORM: defer("mllm_hip_rmsnorm", mllm_hip_rmsnorm, o.a, o.w, o.out, (int8_t *)nullptr, (float *)nullptr, (int16_t *)nullptr, 1, (int)o.n, o.eps, 0, stream_); break;
    case LazyOp::LINEAR: defer("mllm_hip_linear", mllm_hip_linear, o.W, (int)MLLM_HIP_Q4_K, o.w, o.a, (void *)o.out, (int)MLLM_HIP_F32, (int64_t)o.n, 1, (int)o.n, o.K, o.ws, stream_); break;
    case LazyOp::SILU: defer("mllm_hip_silu", mllm_hip_silu, o.a, o.out, o.n, stream_); break;
    case LazyOp::MUL: defer("mllm_hip_mul", mllm_hip_mul, o.a, o.b, o.out, o.n, stream_); break;
    case LazyOp::ADD: defer("mllm_hip_add", mllm_hip_add, o.a, o.b, o.out, o.n, stream_); break;
    case LazyOp::ROPE: defer("mllm_hip_rope_apply", mllm_hip_rope_apply, o.a, (int64_t)o.H * o.D, o.sin, o.cos, o.ld_tab, (void *)o.out, (int)MLLM_HIP_F32, (int64_t)o.H * o.D, o.S, o.H, o.D, stream_); break;
    case LazyOp::KVSTORE: defer("mllm_hip_store_f16", mllm_hip_store_f16, o.a, (int64_t)o.n, o.dst16, (int64_t)o.n, o.S, (int)o.n, stream_); break;
    case LazyOp::NORM_M:
        if (o.layer_norm) defer("mllm_hip_layernorm", mllm_hip_layernorm, o.a, o.w, o.b, o.out, (int8_t *)nullptr, (float *)nullptr, (int16_t *)nullptr, o.M, (int)o.n, o.eps, stream_);
        else defer("mllm_hip_rmsnorm", mllm_hip_rmsnorm, o.a, o.w, o.out, (int8_t *)nullptr, (float *)nullptr, (int16_t *)nullptr, o.M, (int)o.n, o.eps, o.unit_offset, stream_);
        break;
    case LazyOp::LINEAR_M:
        defer("mllm_hip_quantize_q8k_packed", mllm_hip_quantize_q8k_packed, o.a, o.ws, o.M, o.K, stream_);
        defer("mllm_hip_linear_q4kp_packed", mllm_hip_linear_q4kp_packed, o.Wpacked, o.w, (const void *)o.ws, (void *)o.out, (int)MLLM_HIP_F32, (int64_t)o.n, (const float *)nullptr, o.M,
              (int)o.n, o.K, stream_);
        break;
    case LazyOp::FA2:
        defer("mllm_hip_fa2", mllm_hip_fa2, o.a, (int64_t)o.H * o.D, o.kp, (int64_t)o.Hkv * o.D, o.vp, (int64_t)o.Hkv * o.D, o.kvdt, o.out, (int64_t)o.H * o.D, 1, o.Sk, o.H, o.Hkv, o.D,
              o.causal, (const int *)nullptr, (void *)nullptr, stream_);
        break;
    }
    flushing_ = was;
}
// emits the longest run starting at lazy_[i] that one launch covers (else lazy_[i] alone); returns how many Ops it consumed
size_t HIPBackend::emit_group(size_t i) {
    const std::vector<LazyOp> &L = lazy_;
    const size_t n = L.size();
    auto fusable = [](const LazyOp &o) { return o.kind == LazyOp::LINEAR && o.K > 0 && o.K % 256 == 0 && (o.K / 256 + 7) / 8 <= 5; };
    auto launch = [&](const mllm_hip_row_fused &a, size_t ops) -> size_t {
        defer("mllm_hip_row_fused_launch", +[](mllm_hip_row_fused args, void *st) -> int { return mllm_hip_row_fused_launch(&args, st); }, a, stream_);
        ++fused_launches_;
        fused_ops_ += (long)ops;
        return ops;
    };
    // One launch stands for a run of Ops only if no workgroup can overwrite what another still reads: the frontend hands a block it has released to the next tensor of that size
    // (the ADD's operand becomes the norm's output, the gate's block the up projection's), which is harmless between launches and a race inside one.  So: no output of the launch
    // may overlap an input every workgroup reads (xa, xb) or another row's post_add.  (Outputs that alias EACH OTHER are written in the Ops' order by one thread: gate then up.)
    auto safe = [](const mllm_hip_row_fused &a) -> bool {
        struct R { const void *p; size_t n; };
        R in[5], out[10];
        int ni = 0, no = 0;
        const size_t kb = (size_t)a.K * 4;
        in[ni++] = {a.xa, kb};
        if (a.xb) in[ni++] = {a.xb, kb};
        if (a.sum_out) out[no++] = {a.sum_out, kb};
        if (a.norm_out) out[no++] = {a.norm_out, kb};
        for (int i = 0; i < a.nseg; ++i) {
            const size_t nbytes = (size_t)a.seg[i].N * 4;
            out[no++] = {a.seg[i].y, nbytes};
            if (a.seg[i].post_out) { out[no++] = {a.seg[i].post_out, nbytes}; in[ni++] = {a.seg[i].post_add, nbytes}; }
        }
        if (a.silu_out) out[no++] = {a.silu_out, (size_t)a.seg[0].N * 4};
        if (a.mul_out) out[no++] = {a.mul_out, (size_t)a.seg[0].N * 4};
        for (int o = 0; o < no; ++o)
            for (int i = 0; i < ni; ++i)
                if ((const char *)out[o].p < (const char *)in[i].p + in[i].n && (const char *)in[i].p < (const char *)out[o].p + out[o].n) return false;
        return true;
    };
    auto seg_of = [](mllm_hip_row_seg &s, const LazyOp &lin) { s.W = lin.W; s.bias = lin.w; s.y = lin.out; s.N = (int)lin.n; s.post_add = nullptr; s.post_out = nullptr; s.wg0_ = 0; };
    const LazyOp &o = L[i];
    // [F_TTADD ->] RMSNORM -> LINEAR x 1..3 on the normalised row   |   ... -> LINEAR gate -> SILU -> LINEAR up -> F_TTMUL
    {
        size_t j = i;
        const LazyOp *add = nullptr, *norm = nullptr;
        if (L[j].kind == LazyOp::ADD && j + 1 < n && L[j + 1].kind == LazyOp::NORM && L[j + 1].a == L[j].out && L[j + 1].n == L[j].n) add = &L[j++];
        if (L[j].kind == LazyOp::NORM) norm = &L[j++];
        if (norm && j < n && fusable(L[j]) && L[j].a == norm->out && L[j].K == (int)norm->n) {
            mllm_hip_row_fused a{};
            a.xa = add ? add->a : norm->a;
            a.xb = add ? add->b : nullptr;
            a.sum_out = add ? add->out : nullptr;
            a.norm_w = norm->w; a.norm_out = norm->out; a.eps = norm->eps; a.K = (int)norm->n;
            const size_t head = j - i;
            // the MLP's run
            if (j + 3 < n && L[j + 1].kind == LazyOp::SILU && L[j + 1].a == L[j].out && L[j + 1].n == L[j].n && fusable(L[j + 2]) && L[j + 2].a == norm->out &&
                L[j + 2].K == L[j].K && L[j + 2].n == L[j].n && L[j + 3].kind == LazyOp::MUL && L[j + 3].n == L[j].n &&
                ((L[j + 3].a == L[j + 1].out && L[j + 3].b == L[j + 2].out) || (L[j + 3].b == L[j + 1].out && L[j + 3].a == L[j + 2].out))) {
                a.nseg = 2; a.mode = 1;
                seg_of(a.seg[0], L[j]); seg_of(a.seg[1], L[j + 2]);
                a.silu_out = L[j + 1].out; a.mul_out = L[j + 3].out;
                if (mllm_hip_row_fused_supported(&a) && safe(a)) return launch(a, head + 4);
            }
            a.mode = 0; a.silu_out = a.mul_out = nullptr;
            int cnt = 0;
            while (cnt < 3 && j + cnt < n && fusable(L[j + cnt]) && L[j + cnt].a == norm->out && L[j + cnt].K == (int)norm->n) { seg_of(a.seg[cnt], L[j + cnt]); ++cnt; }
            for (; cnt >= 1; --cnt) {
                a.nseg = cnt;
                if (mllm_hip_row_fused_supported(&a) && safe(a)) return launch(a, head + cnt);
            }
        }
    }
    if (fusable(o)) {
        mllm_hip_row_fused a{};
        a.xa = o.a; a.K = o.K; a.mode = 0;
        // LINEAR -> F_TTADD of its output (o / down projection + the residual)
        if (i + 1 < n && L[i + 1].kind == LazyOp::ADD && L[i + 1].n == o.n && (L[i + 1].a == o.out || L[i + 1].b == o.out) && L[i + 1].a != L[i + 1].b) {
            a.nseg = 1;
            seg_of(a.seg[0], o);
            a.seg[0].post_add = L[i + 1].a == o.out ? L[i + 1].b : L[i + 1].a;
            a.seg[0].post_out = L[i + 1].out;
            if (mllm_hip_row_fused_supported(&a) && safe(a)) return launch(a, 2);
        }
        // LINEAR x 2..3 on the same row (q | k | v without a norm in front)
        int cnt = 0;
        while (cnt < 3 && i + cnt < n && fusable(L[i + cnt]) && L[i + cnt].a == o.a && L[i + cnt].K == o.K) { seg_of(a.seg[cnt], L[i + cnt]); ++cnt; }
        for (; cnt >= 2; --cnt) {
            a.nseg = cnt;
            if (mllm_hip_row_fused_supported(&a) && safe(a)) return launch(a, cnt);
        }
    }
    // prefill: [RMSNORM / LAYERNORM ->] LINEAR x 1..3 on the same M >= 16 rows: ONE packed Q8_K operand (written by the norm itself, else by one quantiser pass) for all of them
    {
        size_t j = i;
        const LazyOp *norm = nullptr;
        if (L[j].kind == LazyOp::NORM_M && j + 1 < n && L[j + 1].kind == LazyOp::LINEAR_M && L[j + 1].a == L[j].out && L[j + 1].K == (int)L[j].n && L[j + 1].M == L[j].M && L[j].n % 256 == 0) norm = &L[j++];
        if (L[j].kind == LazyOp::LINEAR_M) {
            const LazyOp &first = L[j];
            size_t cnt = 1;
            while (cnt < 3 && j + cnt < n && L[j + cnt].kind == LazyOp::LINEAR_M && L[j + cnt].a == first.a && L[j + cnt].K == first.K && L[j + cnt].M == first.M) ++cnt;
            // the MLP's run on prefill rows: gate, SILU, up, F_TTMUL -- the up projection reads the packed operand the norm wrote for the gate (nothing in between touches it)
            if (norm && cnt == 1 && j + 3 < n && L[j + 1].kind == LazyOp::SILU && L[j + 2].kind == LazyOp::LINEAR_M && L[j + 2].a == first.a && L[j + 2].K == first.K &&
                L[j + 2].M == first.M && L[j + 3].kind == LazyOp::MUL) {
                if (norm->layer_norm) defer("mllm_hip_layernorm_packed", mllm_hip_layernorm_packed, norm->a, norm->w, norm->b, norm->out, first.ws, norm->M, (int)norm->n, norm->eps, stream_);
                else defer("mllm_hip_rmsnorm_packed", mllm_hip_rmsnorm_packed, norm->a, norm->w, norm->out, first.ws, norm->M, (int)norm->n, norm->eps, norm->unit_offset, stream_);
                for (size_t k = 0; k < 4; ++k) {
                    const LazyOp &l = L[j + k];
                    if (l.kind == LazyOp::LINEAR_M)
                        defer("mllm_hip_linear_q4kp_packed", mllm_hip_linear_q4kp_packed, l.Wpacked, l.w, (const void *)first.ws, (void *)l.out, (int)MLLM_HIP_F32, (int64_t)l.n,
                              (const float *)nullptr, l.M, (int)l.n, l.K, stream_);
                    else emit_single(l);
                }
                ++fused_launches_;
                fused_ops_ += 5;
                return 5;
            }
            if (norm || cnt > 1) {
                // (no output of these launches may be the rows the later GEMMs still read: the packed operand is a scratch block of the backend's, the norm's input is read by the
                // norm's launch only, and a Linear's output cannot be the block of the rows it shares with its neighbours -- they are alive)
                if (norm) {
                    if (norm->layer_norm) defer("mllm_hip_layernorm_packed", mllm_hip_layernorm_packed, norm->a, norm->w, norm->b, norm->out, first.ws, norm->M, (int)norm->n, norm->eps, stream_);
                    else defer("mllm_hip_rmsnorm_packed", mllm_hip_rmsnorm_packed, norm->a, norm->w, norm->out, first.ws, norm->M, (int)norm->n, norm->eps, norm->unit_offset, stream_);
                } else defer("mllm_hip_quantize_q8k_packed", mllm_hip_quantize_q8k_packed, first.a, first.ws, first.M, first.K, stream_);
                for (size_t k = 0; k < cnt; ++k) {
                    const LazyOp &l = L[j + k];
                    defer("mllm_hip_linear_q4kp_packed", mllm_hip_linear_q4kp_packed, l.Wpacked, l.w, (const void *)first.ws, (void *)l.out, (int)MLLM_HIP_F32, (int64_t)l.n, (const float *)nullptr,
                          l.M, (int)l.n, l.K, stream_);
                }
                ++fused_launches_;
                fused_ops_ += (long)((norm ? 1 : 0) + cnt);
                return (norm ? 1 : 0) + cnt;
            }
        }
    }
    // ROPE(q), ROPE(k), KVCACHE(k) of the rotated k, KVCACHE(v), F_FA2 of one position over the slabs those appends extend: mllm_hip_fa2_decode_step
    if (o.kind == LazyOp::ROPE && i + 4 < n && L[i + 1].kind == LazyOp::ROPE && L[i + 2].kind == LazyOp::KVSTORE && L[i + 3].kind == LazyOp::KVSTORE && L[i + 4].kind == LazyOp::FA2) {
        const LazyOp &rq = o, &rk = L[i + 1], &sk = L[i + 2], &sv = L[i + 3], &fa = L[i + 4];
        const int T = fa.Sk - 1, gsize = fa.Hkv ? fa.H / fa.Hkv : 0;
        const size_t qb = (size_t)rq.H * rq.D * 4, kb = (size_t)rk.H * rk.D * 4;
        // a workgroup per query head reads its head's slice of q, its group's slices of k and v, and writes its head's slices of q_out and O (the group's first: k_out and the
        // slab rows).  A block the frontend has re-used for one of these outputs is harmless when the SAME workgroup owns the slice on both sides (the same base, and for k / v
        // one head per group); anything else must not overlap.
        auto apart = [](const void *p, size_t pn, const void *q, size_t qn) { return (const char *)p + pn <= (const char *)q || (const char *)q + qn <= (const char *)p; };
        auto ok = [&](const void *out, size_t on, const void *in, size_t in_n, bool same_owner) { return apart(out, on, in, in_n) || (same_owner && out == in && on == in_n); };
        const bool clear = ok(fa.out, qb, rq.a, qb, true) && ok(fa.out, qb, rk.a, kb, gsize == 1) && ok(fa.out, qb, sv.a, kb, gsize == 1) && ok(rq.out, qb, rq.a, qb, true) &&
                           ok(rq.out, qb, rk.a, kb, gsize == 1) && ok(rq.out, qb, sv.a, kb, gsize == 1) && ok(rk.out, kb, rq.a, qb, gsize == 1) && ok(rk.out, kb, rk.a, kb, gsize == 1) &&
                           ok(rk.out, kb, sv.a, kb, gsize == 1) && apart(fa.out, qb, rq.out, qb) && apart(fa.out, qb, rk.out, kb) && apart(rq.out, qb, rk.out, kb);
        if (clear && rq.S == 1 && rk.S == 1 && sk.S == 1 && sv.S == 1 && rq.D == rk.D && rq.D == fa.D && rq.H == fa.H && rk.H == fa.Hkv && fa.kvdt == MLLM_HIP_F16 && fa.a == rq.out &&
            sk.a == rk.out && sk.n == (int64_t)rk.H * rk.D && sv.n == sk.n && T >= 0 && sk.dst16 == (const uint16_t *)fa.kp + (size_t)T * sk.n &&
            sv.dst16 == (const uint16_t *)fa.vp + (size_t)T * sv.n && mllm_hip_fa2_decode_step_supported(T, fa.H, fa.Hkv, fa.D)) {
            defer("mllm_hip_fa2_decode_step", mllm_hip_fa2_decode_step, rq.a, rq.sin, rq.cos, rq.out, rk.a, rk.sin, rk.cos, rk.out, sv.a, (uint16_t *)fa.kp, (uint16_t *)fa.vp, T, fa.out,
                  fa.H, fa.Hkv, fa.D, stream_);
            ++fused_launches_;
            fused_ops_ += 5;
            return 5;
        }
    }
    // ROPE(q), ROPE(k), KVCACHE(k) of the rotated k, KVCACHE(v)
    if (o.kind == LazyOp::ROPE && i + 3 < n && L[i + 1].kind == LazyOp::ROPE && L[i + 2].kind == LazyOp::KVSTORE && L[i + 3].kind == LazyOp::KVSTORE) {
        const LazyOp &rq = o, &rk = L[i + 1], &sk = L[i + 2], &sv = L[i + 3];
        // the frontend may hand q's released block to the rotated k (equal head counts): then ROPE(q)'s reads and ROPE(k)'s writes would meet inside one launch
        auto apart = [](const void *p, size_t pn, const void *q, size_t qn) { return (const char *)p + pn <= (const char *)q || (const char *)q + qn <= (const char *)p; };
        const size_t qb = (size_t)rq.S * rq.H * rq.D * 4, kb = (size_t)rk.S * rk.H * rk.D * 4;
        const bool clear = apart(rq.out, qb, rq.a, qb) && apart(rq.out, qb, rk.a, kb) && apart(rq.out, qb, sv.a, kb) && apart(rk.out, kb, rq.a, qb) && apart(rk.out, kb, rk.a, kb) &&
                           apart(rk.out, kb, sv.a, kb) && apart(rq.out, qb, rk.out, kb);
        if (clear && rq.S == rk.S && rq.D == rk.D && sk.a == rk.out && sk.S == rk.S && sv.S == rk.S && sk.n == (int64_t)rk.H * rk.D && sv.n == sk.n) {
            defer("mllm_hip_rope2_store2", mllm_hip_rope2_store2, rq.a, rq.sin, rq.cos, rq.ld_tab, rq.out, rq.H, rk.a, rk.sin, rk.cos, rk.ld_tab, rk.out, sk.dst16, sv.a, sv.dst16, rk.H, rk.S,
                  rk.D, stream_);
            ++fused_launches_;
            fused_ops_ += 4;
            return 4;
        }
    }
    emit_single(o);
    return 1;
}

HIPBackend::~HIPBackend() {
    try { drain(); } catch (...) {}
    for (auto &kv : rope_hf_) { try { dev_release(kv.second); } catch (...) {} }
    rope_hf_.clear();
    stop_.store(true, std::memory_order_release);
    if (worker_.joinable()) worker_.join();
    drain_idle();
}
void HIPBackend::alloc_device(DeviceMemory &mem, DataType) {
    mem.type = MEM_TYPE_GENERIC;
    mem.handle = nullptr;
    if (mem.size_in_bytes == 0) return;      // the trace pass allocates shapeless placeholders (CPUBackend.cpp:318-349): nothing behind them
    auto idle = idle_.find(mem.size_in_bytes);
    if (idle != idle_.end() && !idle->second.empty()) {
        mem.handle = idle->second.back();
        idle->second.pop_back();
        idle_bytes_ -= mem.size_in_bytes;
    } else {
        check(mllm_hip_pool_alloc(&mem.handle, mem.size_in_bytes, stream_), "mllm_hip_pool_alloc");
    }
    blocks_[(uintptr_t)mem.handle] = Block{mem.size_in_bytes, 1, true};
}
void HIPBackend::free_device(DeviceMemory &mem) {
    if (mem.handle) release(mem.handle);
    mem.handle = nullptr;
}
void *HIPBackend::dev_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    check(mllm_hip_alloc(&p, bytes), "mllm_hip_alloc");
    blocks_[(uintptr_t)p] = Block{bytes, 1, false};
    return p;
}
void HIPBackend::dev_release(void *p) {
    if (p) release(p);
}
void HIPBackend::view_of(const std::shared_ptr<Tensor> &view, void *handle, size_t bytes) {
    DeviceMemory &m = view->device_memory();
    if (m.handle == handle) { m.size_in_bytes = bytes; return; }      // already this view (a second setUp on the same shell)
    if (m.handle) free_device(m);
    if (handle) retain(handle);
    m.handle = handle;
    m.type = MEM_TYPE_GENERIC;
    m.size_in_bytes = bytes;
}

void HIPBackend::upload(void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return;
    if (bytes <= (256u << 10)) {      // ids, positions, rotary rows, index tensors: the bytes travel with the deferred call, in order with the launches around them
        auto buf = std::make_shared<std::vector<char>>((const char *)src, (const char *)src + bytes);
        void *st = stream_;
        enqueue([dst, buf, bytes, st]() -> int { return mllm_hip_upload(dst, buf->data(), bytes, st); }, "mllm_hip_upload");
        return;
    }
    drain();      // weights at load time, images: straight from the caller's buffer
    check(mllm_hip_upload(dst, src, bytes, stream_), "mllm_hip_upload");
}
void HIPBackend::copy_from_host(const DeviceMemory &dest, const void *src) {
    if (!dest.handle || !src || dest.size_in_bytes == 0) return;
    upload(dest.handle, src, dest.size_in_bytes);
    if (dest.size_in_bytes <= (64u << 10) && dest.size_in_bytes % 4 == 0) remember_host(dest.handle, (const float *)src, dest.size_in_bytes / 4);      // fact 3
    else shadows_.erase(dest.handle);      // a larger upload into a handle that had a shadow: the shadow is stale now
}
void HIPBackend::copy_to_host(void *dest, const DeviceMemory &src) {
    if (!dest || !src.handle || src.size_in_bytes == 0) return;
    drain();
    check(mllm_hip_d2h(dest, src.handle, src.size_in_bytes, stream_), "mllm_hip_d2h");      // synchronises
}
void HIPBackend::convert_fp_data(Tensor *, Tensor *) {
    // the HIP path keeps activations fp32 (the reference CPU backend's arithmetic type): nothing to convert
}
void HIPBackend::sync() { drain(); check(mllm_hip_sync(stream_), "mllm_hip_sync"); }

void *HIPBackend::scratch(int slot, size_t bytes) {
    if (bytes > scratch_bytes_[slot]) {
        // in-order stream: the old block is freed behind the work that still reads it
        if (scratch_[slot]) defer("mllm_hip_pool_free", mllm_hip_pool_free, scratch_[slot], stream_);
        bytes += bytes / 4;
        check(mllm_hip_pool_alloc(&scratch_[slot], bytes, stream_), "mllm_hip_pool_alloc");
        scratch_bytes_[slot] = bytes;
    }
    return scratch_[slot];
}
static void upload_luts(HIPBackend *b, void **g, void **q) {
    std::vector<uint16_t> hg(65536), hq(65536);
    HIPBackend::check(mllm_hip_build_act_luts(hg.data(), hq.data()), "mllm_hip_build_act_luts");
    *g = b->dev_alloc(65536 * 2);
    *q = b->dev_alloc(65536 * 2);
    b->upload(*g, hg.data(), 65536 * 2);
    b->upload(*q, hq.data(), 65536 * 2);
}
const uint16_t *HIPBackend::gelu_lut() { if (!lut_gelu_) upload_luts(this, &lut_gelu_, &lut_qgelu_); return (const uint16_t *)lut_gelu_; }
const uint16_t *HIPBackend::quickgelu_lut() { if (!lut_qgelu_) upload_luts(this, &lut_gelu_, &lut_qgelu_); return (const uint16_t *)lut_qgelu_; }

// ---- host shadows ---------------------------------------------------------------------------------------------------------------------------------------------
void HIPBackend::remember_host(void *handle, const float *v, size_t n) { shadows_[handle].assign(v, v + n); }
const std::vector<float> &HIPBackend::host_floats(const std::shared_ptr<Tensor> &t) {
    const void *h = t->device_memory().handle;
    auto it = shadows_.find(h);
    if (it != shadows_.end() && it->second.size() >= (size_t)t->count()) return it->second;
    if (t->dtype() != MLLM_TYPE_F32) throw std::runtime_error("HIPBackend::host_floats: fp32 tensors only: " + t->name());
    std::vector<float> v((size_t)t->count());
    drain();
    if (!v.empty()) check(mllm_hip_d2h(v.data(), h, v.size() * 4, stream_), "mllm_hip_d2h");
    return shadows_[h] = std::move(v);
}

// ---- shared Q4_0 tables and rotary tables -----------------------------------------------------------------------------------------------------------------------
std::shared_ptr<HIPQ40Table> HIPBackend::q40_table(AbstructLoader &loader, const std::string &name, int rows, int cols) {
    auto it = q40_by_name_.find(name);
    if (it != q40_by_name_.end()) return it->second;
    auto t = std::make_shared<HIPQ40Table>();
    t->rows = rows; t->cols = cols;
    t->raw.setName(name);
    t->raw.setBackend(this);
    t->raw.reshape(1, 1, rows, cols);
    t->raw.setDtype(MLLM_TYPE_Q4_0);
    t->raw.alloc();
    loader.load(&t->raw);
    const int64_t nblk = (int64_t)rows * (cols / 32);
    t->qs = dev_alloc((size_t)nblk * 16);
    t->d = dev_alloc((size_t)nblk * 2);
    drain();
    check(mllm_hip_repack_q40(t->raw.device_memory().handle, (uint8_t *)t->qs, (uint16_t *)t->d, nblk, stream_), "mllm_hip_repack_q40");
    q40_by_name_[name] = t;
    q40_by_handle_[t->raw.device_memory().handle] = t;
    return t;
}
std::shared_ptr<HIPQ40Table> HIPBackend::q40_table_at(const void *raw_handle) const {
    auto it = q40_by_handle_.find(raw_handle);
    return it == q40_by_handle_.end() ? nullptr : it->second;
}

HIPBackend::RopeTables HIPBackend::mrope_tables(const std::shared_ptr<Tensor> &position_ids, float theta, int D, const std::vector<int> &section) {
    const void *h = position_ids->device_memory().handle;
    const int S = position_ids->dimension(), half = D / 2;
    if (mrope_key_.pos == h && mrope_key_.serial == forward_serial_ && mrope_key_.theta == theta && mrope_key_.D == D && mrope_key_.section == section && mrope_.S == S) return mrope_;
    const std::vector<float> &pos = host_floats(position_ids);      // [3][1][1][S] in BSHD memory = 3 rows of S
    if ((int)pos.size() < 3 * S) throw std::runtime_error("MULTIMODALROPE: position_ids must be [3,1,1,S]");
    std::vector<float> s((size_t)S * half), c((size_t)S * half);
    check(mllm_hip_mrope_table(theta, D, pos.data(), S, section.data(), (int)section.size(), s.data(), c.data()), "mllm_hip_mrope_table");
    const size_t bytes = (size_t)2 * S * half * 4;
    if (bytes > mrope_bytes_) {
        if (mrope_dev_) defer("mllm_hip_pool_free", mllm_hip_pool_free, mrope_dev_, stream_);
        mrope_bytes_ = bytes * 2;
        check(mllm_hip_pool_alloc(&mrope_dev_, mrope_bytes_, stream_), "mllm_hip_pool_alloc");
    }
    float *ds = (float *)mrope_dev_, *dc = ds + (size_t)S * half;
    upload(ds, s.data(), s.size() * 4);
    upload(dc, c.data(), c.size() * 4);
    mrope_key_.pos = h; mrope_key_.serial = forward_serial_; mrope_key_.theta = theta; mrope_key_.D = D; mrope_key_.section = section;
    mrope_ = RopeTables{ds, dc, S, half};
    return mrope_;
}
const float *HIPBackend::rope_hf_tables(float theta, int D, int max_pos) {
    const auto key = std::make_tuple(theta, D, max_pos);
    auto it = rope_hf_.find(key);
    if (it != rope_hf_.end()) return it->second;
    std::vector<float> s((size_t)max_pos * D), c((size_t)max_pos * D);
    check(mllm_hip_rope_table_hf(theta, D, max_pos, s.data(), c.data()), "mllm_hip_rope_table_hf");
    float *tab = (float *)dev_alloc((size_t)2 * max_pos * D * 4);
    upload(tab, s.data(), s.size() * 4);
    upload(tab + (size_t)max_pos * D, c.data(), c.size() * 4);
    rope_hf_[key] = tab;
    return tab;
}
void HIPBackend::set_vision_tables(void *angles_handle, const float *sin, const float *cos, int N, int half) { vision_[angles_handle] = RopeTables{sin, cos, N, half}; }
bool HIPBackend::vision_tables(const void *angles_handle, RopeTables *out) const {
    auto it = vision_.find(angles_handle);
    if (it == vision_.end()) return false;
    *out = it->second;
    return true;
}

// Weights go file -> the library's pinned double buffer -> HBM (mllm_hip_upload) straight from a read-only map of the .mllm; no host tensor stays alive
// (precedent OpenCLBackend.cpp:928-980: map the buffer, fread into it, unmap)
bool HIPBackend::load_from_file(Tensor *tensor, ParamLoader *loader) {
    if (tensor->device_memory().handle == nullptr) return false;      // load() before alloc(): let ParamLoader take its default path
    ParamMetadata md{0, 0};
    try { md = loader->getParamMetadata(tensor->name()); } catch (const std::exception &) { return false; }
    if (md.size == 0) return true;
    if (md.size != tensor->cntSize()) throw std::runtime_error("HIPBackend::load_from_file: size of '" + tensor->name() + "' in the file differs from what the Op expects");
    const std::string path = loader->getParamPath();
    auto it = maps_.find(path);
    if (it == maps_.end()) {
        const int fd = open(path.c_str(), O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); return false; }
        void *p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return false;
        it = maps_.emplace(path, std::make_pair(p, (size_t)st.st_size)).first;
    }
    if (md.offset + md.size > it->second.second) throw std::runtime_error("HIPBackend::load_from_file: '" + tensor->name() + "' lies outside the file");
    upload(tensor->device_memory().handle, (const char *)it->second.first + md.offset, (size_t)md.size);
    tensor->forceResetHostPointer(nullptr);
    return true;
}

Op *HIPBackend::opCreate(const OpParam &op_param, std::string name, int) {
    const OpType type = (OpType)(int)op_param.at("type");
    auto it = creators_.find(type);
    Op *op = it == creators_.end() ? nullptr : it->second(this, op_param, name);
    if (!op) {
        bool seen = false;
        for (auto &r : refused_) seen = seen || (r.first == (int)type && r.second == name);
        if (!seen) refused_.emplace_back((int)type, name);
    }
    return op;
}
TensorFunction *HIPBackend::funcCreate(TensorFuncType) { throw std::runtime_error("HIPBackend: the legacy TensorFunction path is not used (OpenCLBackend throws too)"); }
std::vector<Tensor> HIPBackend::runLayer(Layer *, std::vector<Tensor>, int) { throw std::runtime_error("runLayer is the QNN path"); }

#ifdef HIP_ADAPTER_TIMING
#include <x86intrin.h>
static unsigned long long g_t_runop = 0, g_t_ops = 0, g_t_wrap = 0, g_n = 0, g_t_reshape = 0, g_t_setup = 0, g_t_exec = 0, g_t_tail = 0;
struct TimingDump { ~TimingDump() { fprintf(stderr, "[adapter timing] runOp calls %llu: total %.1f Mcyc, reshape+setUp+execute %.1f Mcyc (reshape %.1f, setUp %.1f, execute %.1f), shell construction %.1f Mcyc, tail %.1f Mcyc\n", g_n, g_t_runop / 1e6, g_t_ops / 1e6, g_t_reshape / 1e6, g_t_setup / 1e6, g_t_exec / 1e6, g_t_wrap / 1e6, g_t_tail / 1e6); } } g_timing_dump;
#define TSC() __rdtsc()
#else
#define TSC() 0ull
#endif
std::vector<Tensor> HIPBackend::runOp(Op *op, std::vector<Tensor> inputs, std::vector<std::string> out_names, bool in_place) {
    const unsigned long long t_in = TSC();
    Module *module = inputs.empty() ? Module::llm_model_ptr : inputs[0].module();
    // fact 2: host-side scalar inputs go back to the CPU backend before anything reads them (the caller's Tensor shares the impl, so the model's own
    // `dataAt` on it works again); done in the trace passes too, where the model code runs on the dummy inputs
    if (auto *hop = dynamic_cast<HIPOp *>(op); hop && hop->host_inputs())
        for (auto &input : inputs)
            if (input.backend() && input.backend()->type() != MLLM_CPU) input.to(MLLM_CPU);
    static map<string, shared_ptr<Tensor>> empty_activation_tensors;
    map<string, shared_ptr<Tensor>> &activation_tensors = module ? module->activation_tensors : empty_activation_tensors;
    if (module && module->doTrace) {      // trace / load pass (CPUBackend.cpp:318-349): named placeholder tensors flow through the model code, nothing is computed.
        // Device Ops plan nothing ahead of time (outputs come from the stream-ordered pool per call), so setUp is not called on the placeholders.
        vector<Tensor> results = {};
        if (!module->tracedFlag) {
            for (auto &input : inputs) {
                if (input.shouldInGraphs() && activation_tensors.find(input.name()) == activation_tensors.end()) {
                    activation_tensors[input.name()] = std::make_shared<Tensor>(op->backend());
                    activation_tensors[input.name()]->setName(input.name());
                    activation_tensors[input.name()]->setModule(module);
                }
            }
        }
        for (const auto &out_name : out_names) {
            if (activation_tensors.find(out_name) == activation_tensors.end()) {
                activation_tensors[out_name] = std::make_shared<Tensor>(op->backend());
                activation_tensors[out_name]->setName(out_name);
                activation_tensors[out_name]->setModule(module);
            }
            results.push_back(*activation_tensors[out_name]);
        }
        return results;
    }
    // run pass: non-owning input handles, fresh output shells named as the caller asks, reshape -> setUp (allocates from the pool / makes views) -> execute
    vector<shared_ptr<Tensor>> input_tensors;
    static const std::shared_ptr<int> no_owner = std::make_shared<int>(0);      // aliasing constructor: a handle on the caller's Tensor without a control block per input
    input_tensors.reserve(inputs.size());
    for (auto &input : inputs) input_tensors.push_back(std::shared_ptr<Tensor>(no_owner, &input));
    vector<shared_ptr<Tensor>> out_tensors;
    if (!in_place) {
        for (const auto &out_name : out_names) {
            auto out_tensor = std::make_shared<Tensor>(op->backend());
            out_tensor->setName(out_name);
            out_tensor->setModule(module);
            out_tensors.push_back(out_tensor);
        }
    } else {
        for (size_t i = 0; i < input_tensors.size() && i < out_names.size(); ++i) {
            input_tensors[i]->setName(out_names[i]);
            out_tensors.push_back(input_tensors[i]);
        }
    }
    const unsigned long long t_a = TSC();
    HIPOp *const hop_fast = dynamic_cast<HIPOp *>(op);      // every Op this backend creates is one; the by-reference entry points spare six vector copies per Op
    if (hop_fast) hop_fast->reshape_(input_tensors, out_tensors); else op->reshape(input_tensors, out_tensors);
    const unsigned long long t_r = TSC();
    if (hop_fast) hop_fast->setUp_(input_tensors, out_tensors); else op->setUp(input_tensors, out_tensors);
    const unsigned long long t_s = TSC();
    if (hop_fast) hop_fast->execute_(input_tensors, out_tensors); else op->execute(input_tensors, out_tensors);
    const unsigned long long t_b = TSC();
    ++ops_run_;
    if (dump_dir_) dump_outputs(op, out_tensors);
    // a shadow describes what the host uploaded; an Op that wrote the block on the device (in place, or into a recycled pool block) has made it stale
    if (!shadows_.empty())
        if (!hop_fast || !hop_fast->keeps_shadow())
            for (const auto &out_tensor : out_tensors) shadows_.erase(out_tensor->device_memory().handle);
    vector<Tensor> results;
    for (const auto &out_tensor : out_tensors) results.push_back(*out_tensor);
#ifdef HIP_ADAPTER_TIMING
    g_t_ops += t_b - t_a; g_t_wrap += t_a - t_in; g_t_runop += TSC() - t_in; ++g_n;
    g_t_reshape += t_r - t_a; g_t_setup += t_s - t_r; g_t_exec += t_b - t_s; g_t_tail += TSC() - t_b;
#endif
    (void)t_in; (void)t_a; (void)t_b; (void)t_r; (void)t_s;
    return results;
}

// bring-up aid (MLLM_HIP_DUMP_DIR set when the backend is created): every Op's fp32 outputs land in <dir>/<serial>_<op name>_<k>.f32, in device memory order (contiguous
// BSHD), with the shape in <dir>/index.txt -- what a stage-by-stage comparison against the oracle's composition needs.  Synchronises after every Op; never on by default.
void HIPBackend::dump_outputs(Op *op, const std::vector<std::shared_ptr<Tensor>> &outs) {
    for (size_t k = 0; k < outs.size(); ++k) {
        auto &t = outs[k];
        if (!t->device_memory().handle || t->count() == 0 || (t->dtype() != MLLM_TYPE_F32 && t->dtype() != MLLM_TYPE_F16)) continue;
        const size_t bytes = (size_t)t->count() * (t->dtype() == MLLM_TYPE_F16 ? 2 : 4);
        std::vector<char> host(bytes);
        drain();
        check(mllm_hip_d2h(host.data(), t->device_memory().handle, bytes, stream_), "mllm_hip_d2h");
        std::string nm = t->name();
        for (auto &ch : nm) if (ch == '/' || ch == ' ') ch = '_';
        char pre[32];
        snprintf(pre, sizeof pre, "%05ld_", ops_run_);
        const std::string file = std::string(pre) + nm + "_" + std::to_string(k) + (t->dtype() == MLLM_TYPE_F16 ? ".f16" : ".f32");
        FILE *f = fopen((std::string(dump_dir_) + "/" + file).c_str(), "wb");
        if (f) { fwrite(host.data(), 1, bytes, f); fclose(f); }
        FILE *ix = fopen((std::string(dump_dir_) + "/index.txt").c_str(), "a");
        if (ix) { fprintf(ix, "%s %d %d %d %d optype=%d\n", file.c_str(), t->batch(), t->head(), t->sequence(), t->dimension(), (int)op->type()); fclose(ix); }
    }
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
        ++forward_serial_;
        time_start = mllm_time_us();
    }
    auto output = module->Forward(inputs, args);
    if (outermost) {
        sync();      // the launchers never synchronise: Module::profiling()'s numbers (mllm/Module.cpp:25-61) need the device to be done
        module->inference_times_.push_back((mllm_time_us() - time_start) / 1000.0F);
    }
    return output;
}

void registerHIPBackendCreator() { InsertBackendCreatorMap(MLLM_HIP_BACKEND_TYPE, std::make_shared<HIPBackendCreator>()); }
HIPBackend *installHIPBackend(int device) {
    auto &slot = Backend::global_backends[MLLM_HIP_BACKEND_TYPE];
    if (!slot) slot = std::make_unique<HIPBackend>(device);
    return static_cast<HIPBackend *>(slot.get());
}

}  // namespace mllm
