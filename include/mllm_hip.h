/*
 * include/mllm_hip.h -- C ABI of libmllm_hip.so: the MI355X (gfx950) backend for mllm's Op/Layer hot path.
 *
 * This is the drop-in boundary of SURVEY.md §8(b): plain `extern "C"` launchers over raw device pointers, sizes and
 * a stream; no C++ or torch types in any signature.  Each entry point names the reference interface it replaces
 * (paths relative to the reference tree).  The reference-side adapter (`mllm/backends/hip/HIPBackend` + `HIP*Op`,
 * INTEGRATION.md) calls these from `Op::execute`; this repo's own engine (the mllm_hip_qwen2vl_* block at the end)
 * reproduces the reference's model graphs on top of the same launchers.
 *
 * Conventions
 *  - every function returns 0 (MLLM_HIP_OK) or a negative MLLM_HIP_ERR_*; nothing throws, nothing falls back to CPU.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Launchers never synchronise.
 *  - activations are row-major fp32 `[M][K]`, which is the reference's BSHD tensor `[B,1,S,K]` with M = B*S, or
 *    `[S][H][D]` for per-head tensors (mllm/Tensor.hpp:264-311 offset rule for BSHD).
 *  - weight bytes are the on-disk == in-memory block layouts of mllm/DataType.hpp (block_q4_K :93-98, 144 B / 256 w;
 *    block_q4_0 :75-78, 18 B / 32 w), `[N][K/blk]` per output row, except where a *_planes layout is named.
 *  - "q8k planes" is this backend's device layout of the reference's activation format block_q8_K
 *    (mllm/DataType.hpp:159-163): qs int8 `[M][K]`, d fp32 `[M][K/256]`, bsums int16 `[M][K/16]` -- the same
 *    values, stored as three planes so every access is 16-byte aligned.  "q80 planes": qs int8 `[M][K]`, d fp16
 *    `[M][K/32]` (block_q8_0 :137-140).
 */
#ifndef MLLM_HIP_H
#define MLLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes (the reference's ErrorCode, mllm/Types.hpp:54-61, is 0 = OK and positive kinds; we keep 0 = OK) */
#define MLLM_HIP_OK 0
#define MLLM_HIP_ERR_HIP (-1)    /* a HIP runtime call failed; see mllm_hip_last_error() */
#define MLLM_HIP_ERR_SHAPE (-2)  /* shape not supported by the kernel (the adapter must refuse at opCreate/reshape) */
#define MLLM_HIP_ERR_DTYPE (-3)
#define MLLM_HIP_ERR_IO (-4)
#define MLLM_HIP_ERR_ARG (-5)

/* DataType values of mllm/Types.hpp:63-97 that occur on this path */
#define MLLM_HIP_F32 0
#define MLLM_HIP_F16 1
#define MLLM_HIP_Q4_0 2
#define MLLM_HIP_Q8_0 8
#define MLLM_HIP_Q4_K 12
#define MLLM_HIP_Q8_K 15

/* ---- device context / memory: Backend::{alloc_device,free_device,copy_from_host,copy_to_host} (mllm/Backend.hpp:60-73),
 *      OpenCL precedent mllm/backends/opencl/OpenCLBackend.cpp:670-787 -------------------------------------------- */
int mllm_hip_init(int device);
const char *mllm_hip_last_error(void);
int mllm_hip_alloc(void **dptr, size_t nbytes);
int mllm_hip_free(void *dptr);
int mllm_hip_h2d(void *dst, const void *src, size_t nbytes, void *stream);
int mllm_hip_d2h(void *dst, const void *src, size_t nbytes, void *stream);
int mllm_hip_sync(void *stream);
/* Measurement / bring-up switches of the launch paths.  They live in the library, not in the caller's environment: a shipped process picks the same kernels whatever
 * its environment holds (the one environment variable left is MLLM_HIP_NO_GRAPH, read once per model by the profiling scripts to replay the decode step as plain launches).
 * value -1 = unset (built-in choice).  Names: vision_batch (images per tower pass), time_layers (layers mllm_hip_model_time_kernel cycles over), no_gub / no_pjb (the
 * 8-lanes-per-block GEMVs instead of one lane per super-block), pjb_min_ns, attn_flags, attn_ds (workgroups per head of the decode attention), head_wpc, gemm_order, no_lnf, merge_o (which kernels of the decode step share a launch, handing
 * their rows over as {value, epoch} pairs: 4 (default) a layer's down projection + the next layer's q|k|v + attention + o-projection; 3 q|k|v + attention + o-projection; 2 / 1 attention +
 * o-projection; 0 five launches per layer), chain_cont (0: the q|k|v role of the merge_o = 4 launch gets workgroups of its own; default: the first down-projection
 * workgroups carry on as that role), gu_persist / qkv_persist (workgroups per CU of the walking form of the long-row gate|up and norm + GEMV kernels: a workgroup makes its
 * Q8_K image once and walks several row groups; default 2 for rows of more than eight super-blocks and for untied lm_heads, 0 = one row group per workgroup). */
int mllm_hip_set_option(const char *name, int value);
int mllm_hip_get_option(const char *name, int *value);
/* one in-order stream per backend instance (hipStream_t as void*); the OpenCL backend's command queue, OpenCLBackend.cpp:476-477 */
int mllm_hip_stream_create(void **stream);
int mllm_hip_stream_destroy(void *stream);
/* stream-ordered pool for the per-call Op outputs that Backend::runOp allocates (alloc_device / free_device on the hot path; the reference's
 * host-side analogue is MemoryPoolManager, mllm/memory/MemoryPoolManager.hpp:15-214): hipMallocAsync / hipFreeAsync on the device's default pool
 * with the release threshold lifted.  A block freed on `stream` is only safe to reuse by work enqueued later on that same stream. */
int mllm_hip_pool_alloc(void **dptr, size_t nbytes, void *stream);
int mllm_hip_pool_free(void *dptr, void *stream);
/* Backend::load_from_file (mllm/Backend.hpp:118; precedent OpenCLBackend.cpp:928-980): pageable host bytes (the ParamLoader's mmap) -> HBM through
 * two pinned 32-MiB staging buffers, memcpy of chunk i+1 under the DMA of chunk i.  Returns when `src_host` has been consumed; the tail DMAs may
 * still be in flight on `stream`.  mllm_hip_upload_release frees the staging buffers (they are created on first use). */
int mllm_hip_upload(void *dst, const void *src_host, size_t nbytes, void *stream);
int mllm_hip_upload_release(void);
/* Page-lock a host buffer the caller keeps (hipHostRegister), so that the per-call copies into it -- the 608 KB logits row that Qwen2VLModel::Forward hands to the
 * host sampler every token (modeling_qwen2_vl.hpp:399-403, Generate.hpp:156-224) -- are one DMA instead of a staged pageable copy.  Unregister before freeing it. */
int mllm_hip_host_register(void *host, size_t nbytes);
int mllm_hip_host_unregister(void *host);

/* ---- A4: activation quantisation. quantize_row_q8_K_reference (ggml QuantizeQ8.cpp:216-251), quantize_row_q8_0_reference
 *      (:32-55), as called by mat_mul (compute/Matmul.cpp:77-120) ----------------------------------------------------- */
int mllm_hip_quantize_q8k(const float *x, int8_t *qs, float *d, int16_t *bsums, int M, int K, void *stream);
int mllm_hip_quantize_q80(const float *x, int8_t *qs, uint16_t *d, int M, int K, void *stream);

/* ---- A1/A2/A5/A6: Linear / mat_mul on pre-quantised activations.
 *      y[m][n] = vec_dot(W[n], xq[m]) (+ bias[n]); y fp32 or fp16 (Matmul.cpp:257-268), row stride ldy elements.
 *      q4k: vec_dot_q4_K_q8_K (ggml VecDotQ4.cpp:32-346); q40: vec_dot_q4_0_q8_0 (:514-545, weights as planes);
 *      f32: vec_dot_fp32 (ggml VecDotFP32.cpp:31-58). `residual` (optional, fp32 [M][N] with stride ldy) is added after
 *      the bias: the `x + inputs[0]` of the decoder blocks (models/qwen2_vl/modeling_qwen2_vl.hpp:320-323) fused in; with an fp16
 *      output a non-NULL residual is refused (MLLM_HIP_ERR_DTYPE). --- */
int mllm_hip_linear_q4k_q8k(const void *W, const float *bias, const int8_t *xqs, const float *xd, const int16_t *xbsums,
                            void *y, int y_dtype, int64_t ldy, const float *residual, int M, int N, int K, void *stream);
int mllm_hip_linear_q40_q80(const uint8_t *Wqs, const uint16_t *Wd, const float *bias, const int8_t *xqs, const uint16_t *xd,
                            float *y, int64_t ldy, int M, int N, int K, void *stream);
/* Q4_K weights that stay resident and meet M >= 16 rows (prefill, vision tower) are re-ordered ONCE into the operand order of
 * the matrix-core GEMM (per column class of vec_dot_q4_K_q8_K's AVX2 lanes, VecDotQ4.cpp:220-283): the nibbles stay nibbles
 * (0.72 bytes per weight in HBM with the fp16 sub-block scales, the mins operands and (d, dmin)) and become the fp16 operand
 * nibble * scale in registers inside the GEMM: mllm_hip_q4k_prepack -> `out` of mllm_hip_q4k_wpack_bytes(N, K) bytes.  The GEMM
 * packs the Q8_K activation planes into fp16 fragments in `xpack` (mllm_hip_q4k_prepack_bytes(M, K) bytes of scratch) and gives
 * the same bits as mllm_hip_linear_q4k_q8k, which for M >= 16 on raw blocks does both packs into stream-ordered scratch itself. */
size_t mllm_hip_q4k_prepack_bytes(int rows, int K);
size_t mllm_hip_q4k_wpack_bytes(int N, int K);
int mllm_hip_q4k_prepack(const void *W, int N, int K, void *out, void *stream);
int mllm_hip_linear_q4kp_q8k(const void *Wpacked, const float *bias, const int8_t *xqs, const float *xd, const int16_t *xbsums,
                             void *xpack, void *y, int y_dtype, int64_t ldy, const float *residual, int M, int N, int K,
                             void *stream);
/* producers that write the Q8_K activations straight in packed form (no separate pack pass in prefill), and the GEMM that
 * consumes them.  Same bytes as quantise-then-pack: quantize_row_q8_K_reference (QuantizeQ8.cpp:216-251) of the fp32 rows / of
 * the RMSNorm (CPURMSNorm.cpp:31-136) / LayerNorm (CPULayerNorm.cpp:49-88) output; `y` (fp32 normalised rows) optional. */
int mllm_hip_quantize_q8k_packed(const float *x, void *xpack, int M, int K, void *stream);
int mllm_hip_rmsnorm_packed(const float *x, const float *w, float *y, void *xpack, int M, int dim, float eps, int add_unit_offset,
                            void *stream);
int mllm_hip_layernorm_packed(const float *x, const float *w, const float *b, float *y, void *xpack, int M, int dim, float eps,
                              void *stream);
/* the quantiser with the preceding activation folded in (prefill): lut = the fp16 LUT of mllm_hip_act_lut (A18), or silu(gate)*up on a fused
 * [M][2 I] gate|up buffer (A14 + F_TTMUL); the values quantised are bit for bit those the separate launches would have stored */
int mllm_hip_quantize_q8k_packed_act(const float *x, const uint16_t *lut, void *xpack, int M, int K, void *stream);
int mllm_hip_quantize_q8k_packed_silu_mul(const float *gu, void *xpack, int M, int I, void *stream);
int mllm_hip_linear_q4kp_packed(const void *Wpacked, const float *bias, const void *xpack, void *y, int y_dtype, int64_t ldy,
                                const float *residual, int M, int N, int K, void *stream);
int mllm_hip_linear_f32(const float *W, const float *bias, const float *x, float *y, int64_t ldy, int M, int N, int K, void *stream);
/* F_MM on BHSD operands, the eager-attention form (backends/cpu/op/CPUMatmulFunc.hpp:155-172 -> compute/GemmFp.hpp:104-150 gemm_fp32, :233-283 gemm_fp32_fp16):
 * per head c[M][N] = a[M][K] b[K][N], all contiguous ([heads][M][K], [heads][K][N], [heads][M][N]); b_dtype MLLM_HIP_F32 or MLLM_HIP_F16.  Elements of full 8 x 8 tiles are one
 * fma chain over K from zero, elements of edge tiles per-256-block partial sums added up -- the reference's x86 micro-kernel / scalar split, bit for bit. */
int mllm_hip_gemm_f32_bhsd(const float *a, const void *b, int b_dtype, float *c, int heads, int M, int N, int K, void *stream);
/* one-call forms = CPULinear::execute (backends/cpu/op/CPULinear.cpp:98-234): fp32 x in, quantise, dot, bias.
 * `workspace` must hold mllm_hip_linear_workspace_bytes(wdtype, M, K) bytes of device memory. */
size_t mllm_hip_linear_workspace_bytes(int wdtype, int M, int K);
int mllm_hip_linear(const void *W, int wdtype, const float *bias, const float *x, void *y, int y_dtype, int64_t ldy,
                    int M, int N, int K, void *workspace, void *stream);
/* ---- one activation row (M = 1) through a RUN of consecutive Ops of the reference's decoder layer in ONE launch -- what integration/hip's lazy window issues when the
 *      frontend hands it, one Op at a time (mllm/Layer.hpp:128-218 -> Backend::runOp), a run such as
 *        F_TTADD -> RMSNORM -> LINEAR q, k, v      |   LINEAR o -> F_TTADD      |   RMSNORM -> LINEAR gate -> SILU -> LINEAR up -> F_TTMUL      |   LINEAR down -> F_TTADD
 *      (models/qwen2_vl/modeling_qwen2_vl.hpp:204-211,248-257,307-315; models/qwen/modeling_qwen.hpp:40-47,78-86).  Every Op's own output tensor is still written and every value is
 *      computed by the arithmetic of that Op's own entry point above (mllm_hip_add, mllm_hip_rmsnorm, mllm_hip_linear M = 1, mllm_hip_silu, mllm_hip_mul): bit-identical results.
 *        prologue  s = xa + xb (xb optional; stored to sum_out if given) -> n = RMSNorm(s; norm_w, eps) (norm_w optional; stored to norm_out if given) -> Q8_K(n)
 *        body      nseg <= 3 Linears of raw Q4_K rows [N][K/256] on that row: y = dot + bias
 *        epilogue  mode 0: post_out = y + post_add per segment (optional);  mode 1 (nseg == 2, equal N): silu_out = silu(y0) (optional store), mul_out = silu_out * y1
 *      K % 256 == 0, K <= 10240; each N >= the rows one workgroup takes (<= 32).  Fields ending in `_` are filled by the library. ---- */
typedef struct mllm_hip_row_seg { const void *W; const float *bias; float *y; const float *post_add; float *post_out; int N; int wg0_; } mllm_hip_row_seg;
typedef struct mllm_hip_row_fused {
    const float *xa; const float *xb; float *sum_out;
    const float *norm_w; float *norm_out; float eps;
    int K; int nseg; int mode; int rpw_;
    mllm_hip_row_seg seg[3];
    float *silu_out; float *mul_out;
} mllm_hip_row_fused;
int mllm_hip_row_fused_launch(const mllm_hip_row_fused *args, void *stream);
int mllm_hip_row_fused_supported(const mllm_hip_row_fused *args);      /* 1 when the launch covers these shapes (host-side check, no device work), else 0 */
/* RoPE(q), RoPE(k), the fp16 store of the rotated k rows and the fp16 store of the v rows (MULTIMODALROPE / ROPE x 2 + KVCACHE x 2 of one attention block) in one launch:
 * mllm_hip_rope_apply's and mllm_hip_store_f16's arithmetic element for element; q_out / k_out are the two RoPE Ops' fp32 outputs, k16 / v16 the cache slab rows to append to */
int mllm_hip_rope2_store2(const float *q, const float *sin_q, const float *cos_q, int ld_tab_q, float *q_out, int Hq, const float *k, const float *sin_k, const float *cos_k, int ld_tab_k,
                          float *k_out, uint16_t *k16, const float *v, uint16_t *v16, int Hkv, int S, int D, void *stream);
/* Q4_0 rows `[N][K/32]` of 18-B blocks -> nibble plane `[N][K/2]` + fp16 scale plane `[N][K/32]` (load-time repack;
 * Backend::load_from_file hook, mllm/Backend.hpp:118) */
int mllm_hip_repack_q40(const void *raw_blocks, uint8_t *qs, uint16_t *d, int64_t n_blocks, void *stream);

/* ---- A8: CPUEmbedding::execute (backends/cpu/op/CPUEmbedding.cpp:38-80); ids travel as fp32 (tokenizers/Tokenizer.hpp:78-88) */
int mllm_hip_embedding_q40(const float *ids, const uint8_t *Wqs, const uint16_t *Wd, float *out, int S, int hidden, int vocab, void *stream);

/* ---- A9: CPURMSNorm::execute (backends/cpu/op/CPURMSNorm.cpp:31-136). y may be NULL when only the q8k planes are wanted;
 *      qs/d/bsums may be NULL when only y is wanted (fusion of A9 with the A4 of the following Linear). --------------- */
int mllm_hip_rmsnorm(const float *x, const float *w, float *y, int8_t *qs, float *d, int16_t *bsums, int M, int dim,
                     float eps, int add_unit_offset, void *stream);
/* ---- A18: CPULayerNorm::execute (backends/cpu/op/CPULayerNorm.cpp:49-88), same optional fused outputs ------------- */
int mllm_hip_layernorm(const float *x, const float *w, const float *b, float *y, int8_t *qs, float *d, int16_t *bsums,
                       int M, int dim, float eps, void *stream);

/* ---- A14/A18/A20: activations and elementwise -------------------------------------------------------------------- */
/* CPUSiLU (op/CPUSiLU.cpp:24-52 -> mllm_v_expf polynomial, compute/ActivationFunction.hpp:96-134) */
int mllm_hip_silu(const float *x, float *y, int64_t n, void *stream);
/* the same per ROW of `dim` values, as CPUSiLU::execute applies it (op/CPUSiLU.cpp:35-47): the dim % 8 trailing values of each row go through libm's expf (mllm_silu_f32) */
int mllm_hip_silu_rows(const float *x, float *y, int64_t rows, int dim, void *stream);
/* silu(gate)*up of QWen2MLP (modeling_qwen2_vl.hpp:205-208): gu is `[M][2*I]` with gate in cols [0,I), up in [I,2I) */
int mllm_hip_silu_mul(const float *gu, float *y, int M, int I, void *stream);
/* CPUGELU / CPUQuickGELU through the fp16 LUTs (ggml Quantize.hpp:74-131). `lut` = 65536 fp16 entries on device;
 * mllm_hip_build_act_luts fills host tables with the reference's libm formulas. kind: 0 GELU(tanh), 1 QuickGELU */
int mllm_hip_build_act_luts(uint16_t *gelu_host, uint16_t *quickgelu_host);
int mllm_hip_act_lut(const float *x, float *y, int64_t n, const uint16_t *lut, void *stream);
/* CPUBinaryFunc F_TTADD / F_TTMUL (op/CPUBinaryFunc.hpp) */
int mllm_hip_add(const float *a, const float *b, float *y, int64_t n, void *stream);
int mllm_hip_mul(const float *a, const float *b, float *y, int64_t n, void *stream);
/* CPUSoftMax (op/CPUSoftMax.cpp:28-65): rows of n, `valid` leading columns un-masked (valid == NULL: all) */
int mllm_hip_softmax(const float *x, float *y, int rows, int n, const int *valid, void *stream);
/* CPUIndexPutFunc (op/CPUIndexPutFunc.hpp:25-92): rows of `value` replace rows idx[i] of `dst` */
int mllm_hip_index_put_rows(float *dst, const float *value, const int *idx, int n_rows, int dim, void *stream);
/* the same with the indices as the function receives them, a device tensor of floats (`(int)replace_idx->dataAt<float>`, CPUIndexPutFunc.hpp:85-92);
 * rows whose destination is outside [0, n_dst_rows) are skipped */
int mllm_hip_index_put_rows_fidx(float *dst, int n_dst_rows, const float *value, const float *idx, int n_rows, int dim, void *stream);
/* CPUSplitFunc (op/CPUSplitFunc.hpp:145-172 -> efficient_split, compute/Split.hpp) on DIMENSION, one output at a time: the `[rows][cols]` window of a pitched
 * fp32 buffer into another (cols, pitches and bases multiples of 4 floats) */
int mllm_hip_copy_2d_f32(const float *src, int64_t lds, float *dst, int64_t ldd, int rows, int cols, void *stream);
/* the data-moving case of CPUTransposeFunc (op/CPUTransposeFunc.hpp; most transposes of the graphs are metadata): y[c][r] = x[r][c].  Used by the
 * reference-side Conv2D adapter to hand its `[oh*ow][OC]` rows back in the reference's `[OC][oh][ow]` output order (Convolution.cpp:35-82). */
int mllm_hip_transpose_f32(const float *x, float *y, int rows, int cols, void *stream);
/* host argmax of the logits row (processing_qwen2_vl.hpp:284-289), moved to device */
int mllm_hip_argmax(const float *x, int n, int *out_index, void *stream);
/* SURVEY N2, top-k sampling (mllm/Generate.cpp:45-90, _LlmTextGenerateTopkSamplingMethod::generate): the k largest logits in descending
 * order with their indices (std::partial_sort of :60-61; equal logits by ascending index) selected on the device -- k values cross PCIe
 * instead of the vocabulary row -- and the method's temperature softmax + renormalisation (:69-87) over those k on the host with the
 * host's libm exp, as the reference computes it.  The draw itself (std::discrete_distribution seeded from std::random_device, Generate.hpp:38-44)
 * is not reproducible in the reference and stays with the caller.  k <= 64. */
int mllm_hip_topk(const float *x, int n, int k, float *out_val, int *out_idx, void *stream);
int mllm_hip_topk_probs_host(const float *top_val, int k, float temperature, float *probs);
/* SURVEY N2, top-p (nucleus) sampling (mllm/Generate.cpp:93-142, _LlmTextGenerateToppSamplingMethod::generate): the std::sort of the whole
 * (score, index) row (:99) as one device radix sort, keys descending, equal keys by ascending index; the caller then reads only the prefix whose
 * running float sum reaches p (:108-115).  workspace: mllm_hip_sort_desc_workspace_bytes(n) bytes of device memory. */
size_t mllm_hip_sort_desc_workspace_bytes(int n);
int mllm_hip_sort_desc(const float *x, int n, float *val_sorted, int *idx_sorted, void *workspace, size_t workspace_bytes, void *stream);
/* the draw of both sampling methods, _sample_element (mllm/Generate.hpp:38-44: std::discrete_distribution over the float probabilities), as an
 * inverse CDF on a caller-supplied uniform number u01 in [0,1): returns the index of the drawn candidate (host arithmetic). */
int mllm_hip_sample_index_host(const float *probs, int k, float u01);

/* ---- A10/A11/A19: rotary embeddings. Tables are built on the host with the reference's libm formulas
 *      (CPURoPE.cpp:22-31,100-128; CPUMultimodalRoPE.cpp:26-36,84-118,37-82; CPUVisionRoPE.cpp:19-55) and uploaded;
 *      the rotate is rope_hf (CPUMultimodalRoPE.cpp:153-221): out[d] = x[d]c - x[d+half]s, out[d+half] = x[d]s + x[d+half]c.
 *      x `[S][H][D]` with row stride ldx (elements, per s); out fp32 or fp16 with row stride ldo: writing K straight
 *      into the fp16 cache slab is A12's zero-copy append (CPUKVCache.cpp:253-275). sin/cos `[S][ld_tab]`, cols < D/2. - */
int mllm_hip_rope_table_hf(float base, int dim, int n_pos, float *sin_host, float *cos_host);
/* the HF table with llama3 frequency scaling (rope_scaling {rope_type: llama3}, Layer.hpp:493-531 -> _compute_llama3_theta, CPURoPE.cpp:33-71), for the
 * Llama-3.x family; the rotate itself is the same mllm_hip_rope_apply.  Tables `[n_pos][dim]`, both halves filled. */
int mllm_hip_rope_table_hf_llama3(float base, int dim, int n_pos, float factor, float low_freq_factor, float high_freq_factor, float original_max_pos,
                                  float *sin_host, float *cos_host);
/* NTKROPE (MiniCPM3 / Phi-3 LongRoPE; Layer.hpp:1171-1198 -> CPUNTKRoPE.cpp:27-80): tables `[n_pos][dim]` for n_pos = max_position_embeddings
 * positions; `long_factor` / `short_factor` hold dim / 2 divisors (the long ones apply iff n_pos > original_max_pos); the rotate is mllm_hip_rope_apply. */
int mllm_hip_rope_table_ntk(float theta, int dim, int n_pos, int original_max_pos, const float *long_factor, const float *short_factor,
                            float *sin_host, float *cos_host);
int mllm_hip_mrope_table(float base, int dim, const float *pos3xS_host, int S, const int *section, int n_section,
                         float *sin_host, float *cos_host);
int mllm_hip_vision_rope_table(int t, int h, int w, int merge, int rot_dim, float *sin_host, float *cos_host);
/* the VISIONROPE layer's own output (CPUVisionRoPE.cpp:19-147): the angle table `[t*h*w][rot_dim]` (h angles, then w angles) whose sin / cos
 * F_APPLY_VISIOROPE evaluates per use (CPUVisionRoPEFunc.hpp:21-60); mllm_hip_vision_rope_table = sinf / cosf of these */
int mllm_hip_vision_rope_angles(int t, int h, int w, int merge, int rot_dim, float *angles_host);
int mllm_hip_rope_apply(const float *x, int64_t ldx, const float *sin_t, const float *cos_t, int ld_tab, void *out,
                        int out_dtype, int64_t ldo, int S, int H, int D, void *stream);
/* fp32 -> fp16 strided copy: V rows into the cache slab (the fp16 store branch of mat_mul, Matmul.cpp:262-268) */
int mllm_hip_store_f16(const float *x, int64_t ldx, uint16_t *out, int64_t ldo, int S, int n, void *stream);
/* same values, transposed: out[c * ldo + s].  The resident engine keeps its V slab as [Hkv*D][cache rows] so that the sequential
 * P.V walk of __fa2_decode (FlashAttention2.hpp:1075-1110) reads one dim's values contiguously in key order. */
int mllm_hip_store_f16_t(const float *x, int64_t ldx, uint16_t *out, int64_t ldo, int S, int n, void *stream);
/* what sits between the fused q|k|v projection and the prefill attention, in one launch: rows [S][(Hq + 2 Hkv) D] fp32 -> q rotated in place, k rotated and stored as fp16 into
 * k_rows[s * ldk + h * D + d] (the KVCache append, CPUKVCache.cpp:253-275), v stored as fp16 transposed into v_t[(h * D + d) * ldv + s]; the same arithmetic as
 * mllm_hip_rope_apply (twice) + mllm_hip_store_f16_t, element for element */
int mllm_hip_qkv_rope_append(float *qkv, int64_t ldq, const float *sin_t, const float *cos_t, int ld_tab, uint16_t *k_rows, int64_t ldk, uint16_t *v_t, int64_t ldv,
                             int S, int Hq, int Hkv, int D, void *stream);

/* ---- A13: flash_attention_2_forward (compute/FlashAttention2.hpp:2236-2284; fp16-KV impl :1212, fp32-KV impl :87).
 *      O = softmax(Q K^T / sqrt(D) + causal) V, GQA kv_head = q_head / (Hq/Hkv), causal offset Sk - Sq, fp32 accumulate.
 *      Q `[Sq][Hq][D]` fp32 (row stride ldq), K/V `[Sk][Hkv][D]` fp16 or fp32 (row strides ldk/ldv), O `[Sq][Hq][D]` fp32.
 *      `sk_dev` (optional device int) overrides Sk at run time so one captured graph serves every decode step. ---------- */
int mllm_hip_fa2(const float *Q, int64_t ldq, const void *K, int64_t ldk, const void *V, int64_t ldv, int kv_dtype, float *O,
                 int64_t ldo, int Sq, int Sk, int Hq, int Hkv, int D, int causal, const int *sk_dev, void *workspace,
                 void *stream);
size_t mllm_hip_fa2_workspace_bytes(int Sq, int Hq, int D, int max_sk);
/* ONE decode position through the five Ops of the reference's attention block (models/qwen2_vl/modeling_qwen2_vl.hpp:254-262, models/transformer/modeling_transformer.hpp:169-184)
 * in one launch: q_out = RoPE(q_raw), k_out = RoPE(k_raw) (rows of D / 2 sines / cosines of this position), row T of the fp16 slabs `[T + 1][Hkv * D]` = fp16(k_out), fp16(v_raw)
 * (the two KVCache appends), O = F_FA2(q_out, keys 0 .. T).  mllm_hip_rope_apply's, mllm_hip_store_f16's and mllm_hip_fa2's (Sq = 1, fp16 K / V) arithmetic, element for
 * element; D = 64 or 128.  _supported: 1 when a launch with these extents exists (host-side check). */
int mllm_hip_fa2_decode_step(const float *q_raw, const float *sin_q, const float *cos_q, float *q_out, const float *k_raw, const float *sin_k, const float *cos_k, float *k_out,
                             const float *v_raw, uint16_t *kslab, uint16_t *vslab, int T, float *O, int Hq, int Hkv, int D, void *stream);
int mllm_hip_fa2_decode_step_supported(int T, int Hq, int Hkv, int D);
/* nbatch independent attentions of one geometry in one launch (the images of a vision pass): set b uses q / k / v / o at element offsets b*bq / b*bk / b*bv / b*bo.
 * Same arithmetic as mllm_hip_fa2 per set; Sq >= 4. */
int mllm_hip_fa2_batch(const float *Q, int64_t ldq, const void *K, int64_t ldk, const void *V, int64_t ldv, int kv_dtype, float *O,
                       int64_t ldo, int Sq, int Sk, int Hq, int Hkv, int D, int causal, int nbatch, int64_t bq, int64_t bk, int64_t bv,
                       int64_t bo, void *stream);
/* attention on the engine's KV layout: K fp16 rows `[Sk][Hkv*D]`, V fp16 transposed `[Hkv*D][ldvt]` (mllm_hip_store_f16_t) */
int mllm_hip_fa2_vt(const float *Q, int64_t ldq, const void *K, int64_t ldk, const void *Vt, int64_t ldvt, float *O, int64_t ldo,
                    int Sq, int Sk, int Hq, int Hkv, int D, int causal, void *stream);

/* ---- A16/A17: patch-embedding convolution with kernel == stride as a GEMM over flattened receptive fields
 *      (compute/Convolution.cpp:35-82,179-235): out[n][oc] = vec_dot_fp32(W[oc], patch[n]) + bias ------------------- */
int mllm_hip_patch_gemm_f32(const float *patches, const float *W, const float *bias, float *out, int N, int KK, int OC, void *stream);
/* gather of conv2d receptive fields from the (h, c, w)-ordered image into `[oh*ow][c][kh][kw]` rows (Convolution.cpp:8-33,45-60) */
int mllm_hip_im2patch_hcw(const float *img, float *patches, int H, int C, int W, int p, void *stream);
/* the same gather from an image in the MEMORY order of the reference's image Tensor: `[B, head = H, sequence = C, dimension = W]` is BSHD, i.e. `[C][H][W]` in memory
 * (mllm/Tensor.hpp:264-311 offset(); what Backend::copy_from_host uploads for CPUConvolution2D's input, op/CPUConvolution2D.cpp:112-149) -- the form the Backend / Op adapter binds */
int mllm_hip_im2patch_chw(const float *img, float *patches, int H, int C, int W, int p, void *stream);

/* ---- Qwen2-VL image preprocessing on the device.  Qwen2VLImageProcessor::preprocess_images (models/qwen2_vl/processing_qwen2_vl.hpp:190-235) after the
 *      decode: x/255 (processor/PreProcess.cpp:37-43), smart_resize (:84-109), the cubic B-spline resample of stb_image_resize2 with edge clamp (PreProcess.cpp:84-154), per-channel
 *      normalise (:233-262), the frame doubled, convertPatches (:119-177).  rgb_host: decoded image `[height][width][3]` uint8; patches_dev: device fp32
 *      `[grid_h*grid_w][1176]` (size it with _shape); grid_thw: 3 host ints out.  min/max_pixels: the processor's 4*28*28 / 16384*28*28 unless set_pixels changed them.
 *      The result can be handed to mllm_hip_model_prefill / _vision as the image (they accept device memory).  Held to the reference within 2e-5 absolute (the library's SIMD
 *      summation order is not reproduced), see kernels_image.hip. -------------------------------------------------------------------------------------------------------- */
int mllm_hip_qwen2vl_preprocess_shape(int height, int width, int min_pixels, int max_pixels, int32_t *grid_thw);
int mllm_hip_qwen2vl_preprocess(const uint8_t *rgb_host, int height, int width, int min_pixels, int max_pixels, float *patches_dev, int32_t *grid_thw, void *stream);

/* ================================================================================================================
 * Engine: the reference's model graphs for the hot-path configs, resident on the device, on top of the
 * launchers above.  One engine type serves the five BASELINE configs; `arch` picks the graph and the tensor names:
 *   QWEN2VL  Qwen2VLModel (models/qwen2_vl/modeling_qwen2_vl.hpp:21-404): M-RoPE decoder + its vision tower + PatchMerger
 *   QWEN     QWenForCausalLM (models/qwen/modeling_qwen.hpp:131-179): HF rotary, q/k/v bias, tied or Linear lm_head
 *   LLAMA    TinyLLaMAModel / LLaMAModel (models/tinyllama/modeling_tinyllama.hpp:15-84, models/llama/modeling_llama.hpp:38-117):
 *            HF rotary, no bias, Linear lm_head
 *   LLAVA    LLaVAModel (models/llava/modeling_llava.hpp:39-137): LLaMA body under "language_model." + CLIP tower + projector
 *   VIT      ViTModel (models/vit/modeling_vit.hpp:63-111): patch embedding, encoder, classifier (no language model)
 * Mirrors the demos' loop (examples/demo_qwen2_vl.cpp:32-66, demo_qwen.cpp, demo_llava.cpp:39-57: load -> model(input) ->
 * argmax -> next token) and Module::profiling()'s timing definition (mllm/Module.cpp:35-42).
 * Weights: Q4_K Linears, Q4_0 embed_tokens, F32 norms / biases / patch-embedding convolutions -- what `quantize ... Q4_K` writes
 * (tools/quantizer/QuantWriter.cpp:123-157).
 * ============================================================================================================== */
#define MLLM_HIP_ARCH_QWEN2VL 0
#define MLLM_HIP_ARCH_QWEN 1
#define MLLM_HIP_ARCH_LLAMA 2
#define MLLM_HIP_ARCH_LLAVA 3
#define MLLM_HIP_ARCH_VIT 4

typedef struct mllm_hip_model_config {
    int arch;
    int hidden, inter, layers, heads, kv_heads, vocab;
    float rms_eps;         /* decoder blocks' RMSNorm epsilon */
    float final_eps;       /* model.norm epsilon: 1e-6 hard-coded in Qwen2VLModel (:374), LLaMA/TinyLLaMA/LLaVA; config.rms_norm_eps in QWen (:105) */
    float rope_theta;
    int mrope_section[3];  /* QWEN2VL only */
    int cache_limit;       /* KV slab length, `-l` of the demos */
    int tie_embedding;     /* lm_head = embed_tokens^T (Tensor::mm, modeling_qwen.hpp:158-159) */
    int qkv_bias;
    /* vision tower (0 = none): QWEN2VL patch 14 / merge 2 / mlp 4*dim; LLAVA CLIP-L (v_ffn also the projector width); VIT */
    int v_dim, v_heads, v_blocks, v_patch, v_merge, v_ffn, v_img, v_classes;
    int image_token_id, vision_start_token_id, vision_end_token_id, video_token_id;
} mllm_hip_model_config;

typedef struct mllm_hip_model mllm_hip_model;

/* Module::load (mllm/Module.hpp:215-225) + ParamLoader (mllm/ParamLoader.cpp:88-141,157-286): mmap the .mllm and stream it to HBM
 * through two pinned staging buffers (hipMemcpyAsync on a copy stream) while the load-time repack kernels of the tensors already
 * resident run on the compute stream (Backend::load_from_file hook, mllm/Backend.hpp:118). */
int mllm_hip_model_create(const mllm_hip_model_config *cfg, const char *mllm_path, mllm_hip_model **out);
void mllm_hip_model_destroy(mllm_hip_model *m);
/* load-time report in the spirit of Module::profiling()'s load_time (mllm/Module.cpp:25-33): wall ms of create(), bytes read from
 * the file, ms the copy stream was busy, ms of repack kernels.  Any pointer may be NULL. */
int mllm_hip_model_load_stats(const mllm_hip_model *m, float *total_ms, int64_t *file_bytes, float *h2d_ms, float *repack_ms);
/* device bytes the model holds after the load (weights in the forms its kernels read, activations, KV slabs), and the bytes of weight forms that were NOT kept
 * because no caller reads them: the GEMM-order copy of a Linear lm_head (it only ever meets one row), the on-disk rows of the vision-tower blocks once packed
 * (a tower pass never has fewer than 16 rows). */
int mllm_hip_model_memory_stats(const mllm_hip_model *m, int64_t *resident_bytes, int64_t *released_bytes);
/* Module::clear_kvcache: KVCache sequence counters and RoPE position counters back to 0 (CPUKVCache.hpp:26-29, CPURoPE.hpp:63-65) */
int mllm_hip_model_clear_kvcache(mllm_hip_model *m);
/* tokens the KV cache holds (KVCache::getCacheSeqLen, CPUKVCache.hpp:22-25; mllm/Op.hpp:121-124): 0 on a fresh or cleared model; -1 for a NULL model */
int mllm_hip_model_cache_len(const mllm_hip_model *m);
/* One prefill forward.  ids: n_ids host ints.  image (optional, fp32, host or device memory): QWEN2VL pixel_values `[n_patch][3*2*14*14]` with
 * image_meta = grid_thw (3 ints); LLAVA one image `[H][C][W]` (CLIP img2Tensor layout, models/clip/processing_clip.hpp:28-44),
 * image_meta NULL.  visual_dev (optional, device fp32): the tower's output rows already computed (e.g. all-gathered from the ranks
 * of the image shard) -- then `image` is not run.  QWEN2VL puts row i at the i-th image_token_id position (where + index_put,
 * modeling_qwen2_vl.hpp:386-393); LLAVA replaces its single 32000 row by the N rows (modeling_llava.hpp:128-132), so the sequence
 * grows to n_ids - 1 + N.  Outputs (each optional): logits_host `[vocab]` of the last token, next_token (greedy argmax, first maximum like
 * std::max_element), elapsed_ms (device time, inputs resident in HBM when it starts). */
int mllm_hip_model_prefill(mllm_hip_model *m, const int32_t *ids, int n_ids, const float *image, const int32_t *image_meta,
                           const float *visual_dev, int n_visual_rows, float *logits_host, int32_t *next_token, float *elapsed_ms);
/* One decode forward for `token` at the next position */
int mllm_hip_model_decode(mllm_hip_model *m, int32_t token, float *logits_host, int32_t *next_token, float *elapsed_ms);
/* Batched decode (the reference's hook: KVCache_batch, mllm/Types.hpp:26-33): B <= 15 independent sequences share one pass over the weights per step.
 * batch_begin(B) gives the model B sequences (sequence 0 is its own cache; the others get their own K / V slabs); batch_select(seq) makes `seq` the one that
 * mllm_hip_model_prefill / _decode / _generate / _clear_kvcache / _cache_len act on (so every sequence is prefilled with the ordinary call, image prompts included);
 * batch_decode(B, tokens) steps sequences 0 .. B-1 together: tokens[b] is appended to sequence b, logits_host (optional) receives `[B][vocab]`, next_tokens (optional)
 * the B greedy ids.  Rows never mix (attention runs per sequence on its own cache; every other Op is row-wise), so row b equals, bit for bit, what sequence b
 * produces stepping alone.  The single-sequence fused decode step stays the headline path; this is the aggregate-throughput form. */
int mllm_hip_model_batch_begin(mllm_hip_model *m, int B);
int mllm_hip_model_batch_select(mllm_hip_model *m, int seq);
int mllm_hip_model_batch_decode(mllm_hip_model *m, int B, const int32_t *tokens, float *logits_host, int32_t *next_tokens, float *elapsed_ms);
/* `steps` greedy decode forwards back to back on the device (argmax on device, no per-token D2H): SURVEY N2, the loop of
 * Module::generate (mllm/Module.cpp:63-100) with the greedy method.  tokens_host receives the generated ids. */
int mllm_hip_model_generate(mllm_hip_model *m, int32_t first_token, int steps, int32_t *tokens_host, float *elapsed_ms);
/* Sampled generation (mllm/Generate.hpp:156-240; method 0 greedy, 1 top-k, 2 top-p = nucleus): the candidate set is selected on the
 * device each step (k values or the nucleus prefix cross PCIe, never the vocabulary row); the temperature softmax over it and the draw run
 * on the host as in the reference.  `u01` = steps uniform numbers in [0,1) that replace the reference's std::random_device-seeded
 * std::discrete_distribution draw (Generate.hpp:38-44), so a run is reproducible; stops after `steps` or at `eos` (< 0: never).
 * Returns the count generated in *n_out. */
int mllm_hip_model_generate_sampled(mllm_hip_model *m, int32_t first_token, int steps, int method, int top_k, float top_p, float temperature,
                                    const float *u01, int32_t eos, int32_t *tokens_host, int *n_out, float *elapsed_ms);
/* Vision tower only on `n_img` images (the unit of the multi-GPU image shard; uploads of image i+1 overlap the tower on image i).
 * QWEN2VL: images = pixel_values `[n_img][n_patch][1176]`, image_meta = grid_thw, out `[n_img][n_patch/4][hidden]`;
 * LLAVA:   images `[n_img][H][C][W]`, out `[n_img][(H/p)^2][v_ffn]`;   VIT: images `[n_img][H][C][W]`, out `[n_img][v_classes]`.
 * out_dev is device memory (so the shard can all-gather it without a host hop). */
int mllm_hip_model_vision(mllm_hip_model *m, const float *images_host, const int32_t *image_meta, int n_img, float *out_dev, float *elapsed_ms);
/* rows and columns of one image's tower output for this config (0 if the config has no tower) */
int mllm_hip_model_vision_shape(const mllm_hip_model *m, const int32_t *image_meta, int *rows, int *cols);
/* bytes of weights streamed per decode token (SURVEY §8d algorithmic bytes), for bench.py */
int64_t mllm_hip_model_decode_weight_bytes(const mllm_hip_model *m);
/* the stream the engine launches on (hipStream_t as void*), for event timing around it */
void *mllm_hip_model_stream(mllm_hip_model *m);
/* mean ms per launch (HIP events on the engine's stream) of one fused decode kernel on the live decode state, cycling over the layers so
 * that every launch streams cold HBM; which: 10 qkv, 11 attention, 12 o-proj, 13 gate|up, 14 down (0..3: the stand-alone GEMV launcher on
 * gate|up, down, qkv, o).  Returns its algorithmic bytes per launch too. */
int mllm_hip_model_time_kernel(mllm_hip_model *m, int which, int iters, float *ms_per_launch, int64_t *bytes_per_launch);
/* The decode step launch by launch: `steps` greedy steps (the tokens mllm_hip_model_generate would produce; the cache advances the same way) run eagerly with a
 * HIP event either side of every launch on the engine's stream.  us_by_kind[9] / launches_by_kind[9]: mean event-to-event microseconds of one launch of each kind and
 * launches of that kind per step -- 0 q|k|v, 1 attention (+ o-projection workgroups when merged), 2 o-projection, 3 gate|up, 4 down, 5 the chain launch (down + the next
 * layer's q|k|v + attention + o-projection; option merge_o = 4), 6 q|k|v + attention + o-projection, 7 model.norm + lm_head, 8 argmax + state advance.  The first step
 * is not counted when steps > 1 (its launches wait for the host).  For bench.py's per-launch roofline of the step's dominant launch. */
#define MLLM_HIP_STEP_KINDS 9
int mllm_hip_model_time_step(mllm_hip_model *m, int32_t first_token, int steps, float *us_by_kind, int32_t *launches_by_kind, int32_t *last_token);
/* The visual-token exchange of the sharded vision prefill behind the C ABI: every rank contributes `rows_per_rank` rows of `cols`
 * fp32 (its images' tower output, padded to the common count) and receives all ranks' rows in rank order -- one ncclAllGather (RCCL over
 * xGMI) on the engine's stream.  `comm` is an ncclComm_t the host created (mllm_hip_comm_* below); no torch types. */
int mllm_hip_comm_unique_id(void *id128);                                  /* ncclGetUniqueId: 128 bytes, rank 0 shares them */
int mllm_hip_comm_create(const void *id128, int world, int rank, void **comm);
int mllm_hip_comm_destroy(void *comm);
int mllm_hip_all_gather_rows(void *comm, const float *local_dev, float *all_dev, int64_t rows_per_rank, int cols, void *stream);

/* ---- the round-1 names of the Qwen2-VL engine, kept as thin forwards onto the generic engine ------------------------------------ */
typedef struct mllm_hip_qwen2vl_config {
    int hidden, inter, layers, heads, kv_heads, vocab;
    float rms_eps, rope_theta;
    int mrope_section[3];
    int cache_limit;       /* KV slab length, `-l` of the demo (default 800) */
    int tie_embedding;
    int v_dim, v_heads, v_blocks, v_patch, v_merge;
    int image_token_id, vision_start_token_id, vision_end_token_id, video_token_id;
} mllm_hip_qwen2vl_config;
typedef struct mllm_hip_model mllm_hip_qwen2vl;
int mllm_hip_qwen2vl_create(const mllm_hip_qwen2vl_config *cfg, const char *mllm_path, mllm_hip_qwen2vl **out);
void mllm_hip_qwen2vl_destroy(mllm_hip_qwen2vl *m);
int mllm_hip_qwen2vl_clear_kvcache(mllm_hip_qwen2vl *m);
int mllm_hip_qwen2vl_prefill(mllm_hip_qwen2vl *m, const int32_t *ids, int n_ids, const float *pixel_values,
                             const int32_t *grid_thw, float *logits_host, int32_t *next_token, float *elapsed_ms);
int mllm_hip_qwen2vl_decode(mllm_hip_qwen2vl *m, int32_t token, float *logits_host, int32_t *next_token, float *elapsed_ms);
int mllm_hip_qwen2vl_generate(mllm_hip_qwen2vl *m, int32_t first_token, int steps, int32_t *tokens_host, float *elapsed_ms);
int mllm_hip_qwen2vl_vision(mllm_hip_qwen2vl *m, const float *pixel_values_host, const int32_t *grid_thw, int n_img,
                            float *embeds_dev, float *elapsed_ms);
int64_t mllm_hip_qwen2vl_decode_weight_bytes(const mllm_hip_qwen2vl *m);
void *mllm_hip_qwen2vl_stream(mllm_hip_qwen2vl *m);
int mllm_hip_qwen2vl_time_gemv(mllm_hip_qwen2vl *m, int which, int iters, float *ms_per_launch, int64_t *bytes_per_launch);

/* ---- the extra ops of the other model families (kernels_n4.hip).  Index tensors are fp32 on the device, as the reference's functions
 * receive them (a Tensor of floats); every result is bit-identical to the reference's (tests/golden/n4_ops.npz). ------------------------------------------ */
/* SLIDINGWINDOWMASK (Layer.hpp:478-488 -> CPUSlidingWindowMask.cpp:30-58): scores `[S][H][keys]`; key d of row s survives iff
 * s - (window - 1) <= d <= s + (keys - S), everything else becomes std::numeric_limits<float>::lowest(); S == 1 passes through. */
int mllm_hip_sliding_window_mask(const float *x, float *y, int S, int H, int keys, int window, void *stream);
/* F_TOPK on DIMENSION (Tensor::topk, CPUTopkFunc.hpp:48-70; the MoE routers): per row the k largest (value, index) pairs in descending pair order -- among
 * equal values the larger index first; `indices` are written as floats like the reference's second output.  The HEAD-axis form of the function (input [1][H][S][1],
 * output [1][k][S][1]) is the same call: in BSHD memory those are the rows [S][H] -> [S][k] (ldx = H, n = H). */
int mllm_hip_topk_rows(const float *x, int64_t ldx, float *values, float *indices, int rows, int n, int k, void *stream);
/* F_BINCOUNT (CPUBinCountFunc.hpp:20-35): counts of the integer parts 0 .. nbins-1 (the reference's output has max + 1 entries; pass the bins wanted) */
int mllm_hip_bincount(const float *ids, int n, float *counts, int nbins, void *stream);
/* Tensor::clip(index, SEQUENCE) (CPUClipFunc.hpp:309-323): out[r] = src[idx[r]]; with skip_negative = 1 and out = the word-embedding rows it is
 * F_FUYU_GATHER_EMBD (CPUFuyuGatherEmbdFunc.hpp:45-62): rows whose index is negative keep their contents.  `src` has n_src_rows rows: an index outside
 * [0, n_src_rows) is never dereferenced and leaves its output row untouched (the reference would read out of bounds). */
int mllm_hip_gather_rows(const float *src, int64_t lds, int n_src_rows, const float *idx, float *out, int64_t ldo, int R, int D, int skip_negative, void *stream);
/* F_SCATTERADD on SEQUENCE (Tensor::scatter_add, CPUScatterAddFunc.hpp:38-52; the MoE combine): dst[idx[r]] += src[r] for r ascending -- a repeated
 * destination row accumulates in that order.  `dst` has n_dst_rows rows; a source row whose index lies outside [0, n_dst_rows) is skipped. */
int mllm_hip_scatter_add_rows(float *dst, int64_t ldd, int n_dst_rows, const float *src, int64_t lds, const float *idx, int R, int D, void *stream);

/* F_TTMUL with a `[R,1,1,1]` right operand (CPUBinaryFunc.hpp; `expert_out * expert_weights_clip`, models/minicpm_moe/modeling_minicpm_moe.hpp:86): y[r][:] *= w[r] */
int mllm_hip_scale_rows(float *y, int64_t ld, const float *w, int R, int D, void *stream);
/* One sparse-MoE feed-forward block = MiniCPMMoE::Forward (models/minicpm_moe/modeling_minicpm_moe.hpp:52-105; models/ling, models/smallthinker route the same way), composed
 * from the launchers above: router Linear (Q4_K rows `[n_experts][hidden]`) -> softmax -> top-k (per_tok) -> weights renormalised by their sequential sum -> per expert,
 * ascending: gather its tokens, gate (w1) / up (w3) / down (w2) MLP on Q4_K rows `[inter][hidden]`, `[inter][hidden]`, `[hidden][inter]`, rows scaled by their routing
 * weight, scatter-added into the zero-initialised `out [n_tok][hidden]`.  w1 / w3 / w2 are HOST arrays of n_experts device pointers.  The S * k routing pairs cross PCIe
 * once (the reference reads `tokens_per_expert` on the host per expert, :76-78); the call synchronises `stream`.  Bit-identical to the reference's block. */
int mllm_hip_moe_block(const float *x, float *out, int n_tok, int hidden, int inter, int n_experts, int per_tok, const void *router_q4k,
                       const void *const *w1_q4k, const void *const *w3_q4k, const void *const *w2_q4k, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MLLM_HIP_H */
