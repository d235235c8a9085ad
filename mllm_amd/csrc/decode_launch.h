// mllm_amd/csrc/decode_launch.h -- host-side description of one fused decode step (kernels_decode.hip), used by engine.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mllm_hip {

// per-step scalars in device memory, advanced by dec_next_kernel so one captured graph replays every step
struct DecodeState {
    int T;      // tokens already in the KV slab before this step
    int step;   // decode step since the prefill (row of the rotary tables)
    int token;  // token id to embed at this step
    int serial; // steps since the state was armed: the epoch of the in-launch hand-offs (merged attention + o-projection launch)
};

struct DecodeLayer {
    const float *in_norm, *post_norm;
    const uint8_t *Wqkv; const float *bqkv; int qkv_N;
    const uint8_t *Wo, *Wgu, *Wdown;          // all four in decode order (decode_order_q4k)
    const uint8_t *Wgu_raw, *Wdown_raw, *Wo_raw;       // the rows as stored on disk (the one-lane-per-super-block kernels dec_gateup_blk / dec_proj_blk)
};

// Weights the kernels behind the attention launch will stream, touched by workgroups the attention does not need (24 of 256 CUs hold a head): region r is read by the
// later launch's workgroup g as [base[r] + g * bytes[r], + bytes[r]); the warming workgroup with blockIdx % 8 == g % 8 runs on the XCD whose L2 that workgroup will read
// (consecutive launches place blockIdx b on XCD b % 8).  n == 0: off.  A speed matter only.
struct WeightWarm {
    static constexpr int MAXR = 4, GROUPS = 32;      // targets per XCD column: 8 x 32 = 256 workgroups of the later launches
    const uint8_t *base[MAXR];
    int bytes[MAXR], count[MAXR], n;
    uint32_t *sink;
};

struct DecodeCtx {
    DecodeState *state;
    int H, I, heads, kv_heads, D, vocab, cache_limit, nsplit, max_parts;
    float eps, final_eps;
    const uint8_t *emb_qs; const uint16_t *emb_d; const float *final_norm;
    const uint8_t *Whead;               // Linear lm_head rows (Q4_K, decode order) when the head is not tied to embed_tokens, else nullptr
    float *x0, *x1, *qkv, *act, *logits, *fa_ws, *part_val, *normed;
    int8_t *x80_qs; uint16_t *x80_d;
    int *part_idx, *tok_dev, *history;
    const float *rope_sin, *rope_cos;   // [cache_limit][D/2], row = DecodeState::step
    float *cur_sin, *cur_cos;           // [D/2]: the row of the step about to run (refreshed by dec_next)
    uint16_t *kslab, *vslab;            // K: [layers][cache_limit][Hkv*D]; V transposed: [layers][Hkv*D][vt_ld]
    int vt_ld;
    int n_layers;                       // entries of the DecodeLayer array handed to the launchers
    const WeightWarm *warm_tab;         // device, [n_layers] (decode_warm_table), or nullptr: no warming workgroups in the attention launch
    // merged attention + o-projection launch (option "merge_o"): the attention's output row travels as {value, epoch} pairs, one row per layer, epoch = DecodeState::serial;
    // the o-projection's workgroups ride in the attention's launch, fetch their weight rows at once and poll the pairs (profiles/r04_seam_overlap_microbench.md)
    unsigned long long *attn_pairs;     // [n_layers][heads * D], all-ones when the state is armed
    unsigned long long *x_pairs;        // [n_layers][H]: a layer's output row for the next layer's q|k|v role (and the o-projection's residual) when both ride in the down projection's launch (merge_o = 4)
    unsigned long long *qkv_pairs;      // [n_layers][(heads + 2 kv_heads) * D]: q | k | v for the attention role when the q|k|v projection rides in the same launch (merge_o = 3)
    int *poll_err;                      // set when a poll gave up (bounded spins): the step's results are then invalid and the host reports it
    int merge_o;
    int attn_flags;                     // decode_attn_flags() as it stood when the model was created: the warming table was built for these, and every launch of this
                                        // model uses them (option "attn_flags" must be set before mllm_hip_model_create; a later change does not reach a live model)
};

// raw Q4_K rows -> decode order: the nibble dwords of every super-block transposed so that a lane's 16 bytes are one column class (q4k_dot.h)
int decode_order_q4k(const void *src, void *dst, int64_t n_blocks, hipStream_t st);
int decode_attn_flags();
bool decode_merges_o(const DecodeCtx &c);      // the step folds the o-projection into the attention's launch (option merge_o and the shapes that form covers)
int decode_warm_table(const DecodeCtx &c, const DecodeLayer *layers, int n_layers, int flags, WeightWarm *host_out);
int decode_kernel_launch(const DecodeCtx &c, const DecodeLayer *layers, int li, int which, hipStream_t st);
int argmax_row_launch(const DecodeCtx &c, const float *logits, int n, int *out, hipStream_t st);      // first-maximum argmax over the chip, partials in c.part_val / c.part_idx
// Optional marks around every launch of a step (mllm_hip_model_time_step: the step run eagerly with a HIP event either side of each launch).  kind: 0 q|k|v, 1 attention
// (with the o-projection's workgroups when they ride in it), 2 o-projection, 3 gate|up, 4 down, 5 the chain launch (down + the next layer's q|k|v + attention + o-projection),
// 6 q|k|v + attention + o-projection as one launch, 7 model.norm + lm_head, 8 argmax / state advance.
constexpr int STEP_KINDS = 9;
struct StepMarks { int (*mark)(void *user, int kind, int after); void *user; };
int decode_step_launch(const DecodeCtx &c, const DecodeLayer *layers, int n_layers, hipStream_t st, const StepMarks *marks = nullptr);

// ---- batched decode (engine.hip: mllm_hip_model_batch_decode): the one Op of a step that is not row-wise, for all B sequences in one launch each ----
// one sequence's KV slabs (bases of layer 0) and the tokens its cache holds BEFORE this step
struct SeqKV { uint16_t *k; uint16_t *v; int t; int pad; };
// row b of qkv ([B][ldq]: q | k | v of sequence b's new token): q rotated in place, k rotated -> fp16 row t_b of sequence b's K slab, v -> column t_b of its transposed V slab
// (qkv_rope_append_kernel's arithmetic, S = 1 per sequence; rotary row b of sin_t / cos_t)
int seqs_rope_append_launch(float *qkv, int64_t ldq, const float *sin_t, const float *cos_t, int ld_tab, const SeqKV *seqs_dev, int64_t layer_k_off, int64_t layer_v_off,
                            int64_t ldk, int64_t ldvt, int B, int Hq, int Hkv, int D, hipStream_t st);
// __fa2_decode of row b's query over sequence b's t_b + 1 keys (fa2_decode_kernel's body, grid = heads x sequences); cap = the slabs' capacity in keys
int seqs_fa2_decode_launch(const float *q, int64_t ldq, const SeqKV *seqs_dev, int64_t layer_k_off, int64_t layer_v_off, int64_t ldk, int64_t ldvt, float *o, int64_t ldo, int B,
                           int Hq, int Hkv, int D, int cap, hipStream_t st);

}  // namespace mllm_hip
