/*
 * mm_hip.h - C ABI of libmm_hip.so: the MI355X (gfx950) hot path of the NeoX/MAGMA
 * ParallelTransformerLayer (reference: megatron/model/transformer.py:644-801).
 *
 * Conventions (every entry point):
 *   - plain device pointers + sizes; no torch types; caller owns every buffer;
 *   - nothing is allocated and nothing synchronises inside (graph-capture safe) unless stated;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - `dtype`: MM_BF16 or MM_F16 (the reference's fused kernels accept exactly Half/BFloat16,
 *     megatron/fused_kernels/scaled_masked_softmax.cpp:34-37); fp32 accumulation everywhere;
 *   - return 0 on success, a negative MM_E* code otherwise; mm_last_error() gives the text
 *     (the Python shim raises RuntimeError, matching the reference's AT_ASSERTM behaviour).
 */
#ifndef MM_HIP_H
#define MM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MM_BF16 = 0, MM_F16 = 1 };
enum { MM_OK = 0, MM_EINVAL = -1, MM_EUNSUPPORTED = -2, MM_ELAUNCH = -3 };

const char* mm_last_error(void);
/* Library/arch identification: returns "gfx950". */
const char* mm_target_arch(void);
/* Scheduling hint for the calling thread's following launches; returns the previous value.  No counterpart in the reference
 * (its GEMMs are cuBLAS calls on one stream).  A GPT-J layer runs its attention branch and its MLP GEMMs on two HIP streams
 * (model/transformer.py); a persistent GEMM that takes every CU serialises the pair.  With max_workgroups = half the CUs
 * (128 on MI355X) the 256-wide ping-pong kernel (csrc/gemm_pp.hip) walks its tiles with at most that many workgroups when the
 * problem has more than 256 tiles, so that two such launches on two streams run side by side; 0 (the default) = no limit.
 * Every other kernel ignores it (measured: the 256x128 form of the N = h class loses).  Results do not depend on it. */
int mm_set_cu_share(int max_workgroups);

/* ------------------------------------------------------------------ fused softmax
 * Replaces scaled_upper_triang_masked_softmax_cuda.{forward,backward}
 *   (megatron/fused_kernels/scaled_upper_triang_masked_softmax.cpp:31-70,
 *    kernels scaled_upper_triang_masked_softmax.h:143-361).
 * in/out: [attn_batches, s, s] contiguous; row r gets softmax(scale * x[r, 0..r]), zeros for cols > r.
 * s <= 2048 (reference: _cuda.cu:36).  bwd writes IN PLACE into grad_inout (reference: _cuda.cu:77-87). */
int mm_softmax_causal_fwd(void* out, const void* in, float scale, int attn_batches, int s,
                          int dtype, void* stream);
int mm_softmax_causal_bwd(void* grad_inout, const void* softmax_out, float scale, int attn_batches,
                          int s, int dtype, void* stream);

/* Replaces scaled_masked_softmax_cuda.{forward,backward,get_batch_per_block}
 *   (megatron/fused_kernels/scaled_masked_softmax.cpp:33-83, kernels scaled_masked_softmax.h:117-352).
 * in/out: [b, np, sq, sk]; mask: uint8 [pad_batches, 1, sq, sk], pad_batches in {1, b}, value 1 = masked
 * (filled with -10000 before the max, .h:180-184).  sk <= 2048. */
int mm_softmax_masked_fwd(void* out, const void* in, const uint8_t* mask, float scale, int b, int np,
                          int sq, int sk, int pad_batches, int dtype, void* stream);
int mm_softmax_masked_bwd(void* grad_inout, const void* softmax_out, float scale, int b, int np, int sq,
                          int sk, int dtype, void* stream);
/* Same integer the reference computes with C10_WARP_SIZE = 64 on ROCm (scaled_masked_softmax.h:338-352);
 * fused_softmax.py:160-167 uses it for eligibility. Host-only. */
int mm_softmax_batch_per_block(int sq, int sk, int b, int np);

/* ------------------------------------------------------------------ LayerNorm
 * Replaces torch.nn.LayerNorm as used at megatron/model/transformer.py:669,696,737,776,791
 * (megatron/model/norms.py:16-31).  x,y: [rows, h]; w,b: [h] in `dtype`; mean/rstd: [rows] fp32 (saved for bwd).
 * `y2/w2/b2` optional (NULL): second affine output from the same statistics - the GPT-J residual
 * variant normalises the same x twice (transformer.py:737). */
int mm_layernorm_fwd(void* y, void* y2, float* mean, float* rstd, const void* x, const void* w,
                     const void* b, const void* w2, const void* b2, int rows, int h, float eps,
                     int dtype, void* stream);
/* dx = LN'(dy; w) [+ LN'(dy2; w2)] [+ dresidual].  dwdb_partial: fp32 [nblk, 2 or 4, h] scratch, or NULL to skip
 * parameter grads (frozen LM).  nblk = mm_layernorm_bwd_nblk(rows).  Finalize with mm_layernorm_bwd_finalize. */
int mm_layernorm_bwd_nblk(int rows);
int mm_layernorm_bwd(void* dx, float* dwdb_partial, const void* dy, const void* dy2,
                     const void* dresidual, const void* x, const void* w, const void* w2,
                     const float* mean, const float* rstd, int rows, int h, int dtype, void* stream);
/* out_k[h] (dtype) = sum over nblk of partial[:, k, :], k = 0..nvec-1; outs[k] may be NULL. */
int mm_layernorm_bwd_finalize(void* const* outs, const float* dwdb_partial, int nblk, int nvec, int h,
                              int dtype, void* stream);
/* The stack's final LayerNorm (NormPipe, megatron/model/transformer.py:828-839) fused with _post_transformer_block's
 * [s, b, h] -> [b, s, h] transpose(0, 1).contiguous() (megatron/model/gpt2_model.py:80-85): x is read in [s, b, h] order,
 * y / mean / rstd are written in [b, s, h] order - no transposed copy of the activation exists.  The backward takes dy in
 * [b, s, h] order, optionally multiplied by the DEVICE scalar dy_scale[0] (a fused loss node's upstream gradient), and
 * writes dx in [s, b, h] order; dwdb_partial as in mm_layernorm_bwd (nvec = 2). */
int mm_layernorm_fwd_sb(void* y, float* mean, float* rstd, const void* x, const void* w, const void* b, int s, int bsz,
                        int h, float eps, int dtype, void* stream);
int mm_layernorm_bwd_sb(void* dx, float* dwdb_partial, const void* dy, const float* dy_scale, const void* x, const void* w,
                        const float* mean, const float* rstd, int s, int bsz, int h, int dtype, void* stream);

/* ------------------------------------------------------------------ rotary
 * Replaces apply_rotary_pos_emb{,_torch} + the slicing/cat around it
 * (megatron/model/positional_embeddings.py:69-92, transformer.py:566-596), IN PLACE on the fused
 * QKV activation [s, b, np, 3*hn]: q = [..., 0:hn], k = [..., hn:2hn] (transformer.py:555-564).
 * cos/sin: [>= offset+s, d] tables in `dtype` (built the reference's way by the caller).
 * Rounds each product and the sum to `dtype` like the eager reference.  inverse != 0 applies the
 * transposed rotation (backward). */
int mm_rotary_qk_inplace(void* qkv, const void* cos, const void* sin, int s, int b, int np, int hn,
                         int d, int offset, int inverse, int dtype, void* stream);

/* ------------------------------------------------------------------ bias-GeLU / bias-dropout-add
 * Replaces bias_gelu / bias_gelu_back (megatron/model/activations.py:62-96) [mode 0, tanh] and
 * F.gelu(x + bias) (transformer.py:127-130) [mode 1, erf].  y: [rows, cols], bias: [cols]. */
int mm_bias_gelu_fwd(void* out, const void* y, const void* bias, int rows, int cols, int mode,
                     int dtype, void* stream);
/* dy = g * gelu'(y + bias), may alias g. */
int mm_bias_gelu_bwd(void* dy, const void* g, const void* y, const void* bias, int rows, int cols,
                     int mode, int dtype, void* stream);
/* Replaces bias_dropout_add (megatron/model/fused_bias_dropout.py:28-55):
 * out = residual + dropout(x + bias); bias [cols] or NULL, residual [rows, cols] or NULL.
 * p > 0 in training draws a Philox4x32-10 keep mask from (seed, offset) and (optionally) stores it as
 * uint8 in keep_mask (NULL to skip). */
int mm_bias_dropout_add(void* out, const void* x, const void* bias, const void* residual,
                        uint8_t* keep_mask, int rows, int cols, float p, int training,
                        uint64_t seed, uint64_t offset, int dtype, void* stream);
/* mm_bias_dropout_add with the Philox state in DEVICE memory: rng_state = uint64[2] {seed, base offset}; the mask is drawn
 * from (seed, base offset + offset).  A step captured in a HIP graph bakes `offset` (the call's position within the step) into
 * the launch but reads the base at run time, and mm_rng_advance (base offset += n, captured at the end of the step) moves it,
 * so every replay draws fresh masks - the training semantics of the reference's rng tracker (megatron/mpu/random.py:18-37,
 * DeepSpeed's CudaRNGStatesTracker advancing the generator state).  Same results as mm_bias_dropout_add(seed, base + offset). */
int mm_bias_dropout_add_devrng(void* out, const void* x, const void* bias, const void* residual, uint8_t* keep_mask,
                               int rows, int cols, float p, int training, const uint64_t* rng_state, uint64_t offset,
                               int dtype, void* stream);
int mm_rng_advance(uint64_t* rng_state, uint64_t n, void* stream);
/* Column sums: out[cols] (dtype) = sum_rows x[rows, cols] (bias gradients). partial: fp32 [mm_colsum_nblk(rows), cols]. */
int mm_colsum_nblk(int rows);
int mm_colsum(void* out, float* partial, const void* x, int rows, int cols, int dtype, void* stream);
/* out[c, r] = in[r, c] (2-byte elements). */
int mm_transpose2d(void* out, const void* in, int rows, int cols, void* stream);
/* elementwise out = a + b (+ c), n elements. */
int mm_add(void* out, const void* a, const void* b, const void* c, size_t n, int dtype, void* stream);

/* dx = act > 0 ? g : 0 - backward of the adapter's nn.ReLU (megatron/model/adapter.py:18-24); n elements. */
int mm_relu_bwd(void* dx, const void* g, const void* act, size_t n, int dtype, void* stream);
/* The GPT-J residual tail (megatron/model/transformer.py:747-766) in one pass:
 * out = [x +] ((attn + attn_bias) + (mlp + mlp_bias)), every intermediate rounded to `dtype` like the eager
 * bias_dropout_add ops (p = 0).  x == NULL leaves the residual add to the caller (TP > 1: the all-reduce sits between). */
int mm_gptj_combine(void* out, const void* x, const void* attn, const void* attn_bias, const void* mlp,
                    const void* mlp_bias, int rows, int cols, int dtype, void* stream);

/* ------------------------------------------------------------------ GEMM (MFMA)
 * Replaces F.linear in ColumnParallelLinear / RowParallelLinear (megatron/mpu/layers.py:557,745):
 *   C[M, N] = A[M, K] * B[N, K]^T  (+ bias[N]) (+ epilogue), fp32 accumulate, output in `dtype`.
 * A, B row-major with leading dimensions lda, ldb (elements); C row-major ldc.
 * epilogue: MM_EPI_NONE | MM_EPI_BIAS | MM_EPI_BIAS_GELU_TANH | MM_EPI_BIAS_GELU_ERF (aux [M, N], ld = ldc,
 *           receives the rounded pre-bias product that the reference's GeLU backward re-reads; C the activation)
 *           | MM_EPI_BIAS_RELU (adapter, megatron/model/adapter.py:18-24)
 *           | backward epilogues, the data-gradient GEMM fused with the activation's backward kernel; the saved forward
 *             tensor S [M, N] (ld = ldc) is passed in the `residual` slot:
 *               MM_EPI_RELU_BWD       C = S > 0 ? round(A B^T) : 0            (S = the ReLU output; torch relu backward)
 *               MM_EPI_GELU_TANH_BWD  C = round(A B^T) * gelu_tanh'(S + bias) (S = pre-bias product; bias_gelu_back,
 *                                                                              megatron/model/activations.py:75-86)
 *               MM_EPI_GELU_ERF_BWD   C = round(A B^T) * gelu_erf'(round(S + bias))   (F.gelu backward).
 * MFMA path: K % 64 == 0, N % 4 == 0, lda/ldb % 8 == 0, ldc % 4 == 0, 16-byte aligned A/B, 8-byte aligned C/aux/bias/residual;
 * anything else (tiny or ragged problems) runs a generic fp32-FMA kernel with the same epilogue semantics. */
enum {
    MM_EPI_NONE = 0,
    MM_EPI_BIAS = 1,
    MM_EPI_BIAS_GELU_TANH = 2,
    MM_EPI_BIAS_GELU_ERF = 3,
    MM_EPI_BIAS_RELU = 4,
    MM_EPI_RELU_BWD = 5,
    MM_EPI_GELU_TANH_BWD = 6,
    MM_EPI_GELU_ERF_BWD = 7,
    MM_EPI_BIAS_ROTARY = 8 /* internal to mm_gemm_nt_qkv_rotary; mm_gemm_nt rejects it */
};
/* residual (optional, [M, N] with leading dimension ldc, may alias C): C = round(epilogue(...)) + residual. */
int mm_gemm_nt(void* C, void* aux, const void* A, const void* B, const void* bias, const void* residual,
               int M, int N, int K, int lda, int ldb, int ldc, int epilogue, int dtype, void* stream);

/* mm_gemm_nt (epilogue MM_EPI_NONE, bias NULL, no residual) for shapes whose 256 x 256 output tiles would occupy at most
 * half of the 256 CUs: two workgroups per tile, each over half of K, exchange fp32 partials of half a tile through `workspace`
 * inside the launch and both finish half of it.  mm_gemm_nt_splitk_workspace_bytes returns the workspace size in bytes, or 0
 * when the shape does not qualify (M, N multiples of 256 with 64..128 tiles, K a multiple of 128 and >= 8192: the LM head's
 * data gradient; at the layer GEMMs' K <= 4096 the exchange costs more than it gains) - then use mm_gemm_nt.  Same F.linear contract (megatron/mpu/layers.py:557,745); sums in a different order than mm_gemm_nt (fp32
 * partials of the two K halves are added once).  The first bytes of the workspace hold the hand-off flags; the call zeroes
 * them on `stream`.  Calls that share a workspace must not overlap.  The two workgroups of a tile wait for each other with a
 * bounded spin; should a partner never run, its half tile is written as NaN (never as a wrong number). */
size_t mm_gemm_nt_splitk_workspace_bytes(int M, int N, int K);
int mm_gemm_nt_splitk(void* C, const void* A, const void* B, const void* bias, void* workspace, int M, int N, int K,
                      int lda, int ldb, int ldc, int epilogue, int dtype, void* stream);

/* Host-only: name of the kernel mm_gemm_nt dispatches for a contiguous (lda = ldb = K), 16-byte aligned problem of this shape
 * ("gemm_nt_pp_k<256>", "gemm_nt_pp_k<128>", "gemm_nt_128_k" or "gemm_nt_generic_k") - so that a profile row can be matched to a call. */
const char* mm_gemm_nt_kernel_name(int M, int N, int K, int epilogue);

/* The QKV projection with the rotary embedding as its epilogue: C = rotary(A B^T + bias) - ColumnParallelLinear
 * query_key_value followed by apply_rotary_pos_emb on the q and k slices (megatron/model/transformer.py:552-596,
 * positional_embeddings.py:76-92) without the separate in-place pass over the activation.  C is the fused activation
 * [M = s*b tokens, N = np*3*hn] (per head [q | k | v]); token row m sits at sequence position m / b + offset;
 * cos/sin: [>= offset + s, d] tables in `dtype`.  The product + bias is rounded to `dtype` first and the rotation uses
 * the eager reference's roundings, so C is bit-identical to mm_gemm_nt(MM_EPI_BIAS) + mm_rotary_qk_inplace.
 * Fused shapes: hn == 64, d in {16, 32, 64}, N % 192 == 0, the MFMA path's alignment rules, M * b < 2^32;
 * anything else returns MM_EUNSUPPORTED (run the two calls instead). */
int mm_gemm_nt_qkv_rotary(void* C, const void* A, const void* B, const void* bias, const void* cos, const void* sin, int M,
                          int N, int K, int lda, int ldb, int ldc, int b, int hn, int d, int offset, int dtype, void* stream);

/* Weight gradient: C[N, Kp] = A[M, N]^T * B[M, Kp] (contraction over the token dimension M) - the dW = dY^T X half of
 * F.linear's backward (megatron/mpu/layers.py:557,745; adapter nn.Linear, megatron/model/adapter.py:18-24).
 * A, B, C row-major (lda, ldb, ldc in elements).  Requirements: N % 128 == 0, Kp % 128 == 0, M % (64*splits) == 0.
 * splits > 1 splits M over workgroups: workspace is fp32 [splits, N*Kp + N]; the slabs are summed in a fixed order.
 * a_colsum (optional, [N]): also receives the column sums of A - the bias gradient db = sum_m dY[m, :] that the same
 * backward needs - from the tiles already staged for the GEMM (no extra pass over dY).
 * mm_gemm_tn_splits() returns the split count the library would pick (0 = shape not supported: transpose + mm_gemm_nt). */
int mm_gemm_tn_splits(int M, int N, int Kp);
int mm_gemm_tn(void* C, void* a_colsum, float* workspace, const void* A, const void* B, int M, int N, int Kp, int lda,
               int ldb, int ldc, int splits, int dtype, void* stream);
/* mm_gemm_tn without a workspace argument and without any allocation.  NO COUNTERPART IN THE REFERENCE: it exists for callers
 * that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 * The split-K slabs live in a 24 MiB file-scope device array inside the library (one copy per GPU, held until the process
 * ends): splits * (N*Kp + (a_colsum ? N : 0)) must not exceed MM_GEMM_TN_STATIC_MAX_FLOATS (6 Mi fp32), larger requests
 * return MM_EINVAL.  Same kernels, same fixed summation order, so the result is bit-identical to mm_gemm_tn with the same
 * `splits`.  Every call on one GPU shares that GPU's array: calls that overlap in time on one GPU (two streams) race;
 * calls on one stream are ordered.  The array's address is resolved on the first call on each device (hipGetSymbolAddress,
 * no allocation); make that first call outside a graph capture. */
#define MM_GEMM_TN_STATIC_MAX_FLOATS ((size_t)6 << 20)
int mm_gemm_tn_static(void* C, void* a_colsum, const void* A, const void* B, int M, int N, int Kp, int lda, int ldb, int ldc,
                      int splits, int dtype, void* stream);
/* Two weight gradients over the same M tokens in one launch pair (a linear pair such as the adapter's has two small
 * ones: together they fill the chip).  Contiguous operands (lda = N, ldb = ldc = Kp).  workspace: fp32
 * [splits, N0*Kp0 + N0 + N1*Kp1 + N1] with splits = mm_gemm_tn_pair_splits(...) (0 = not a supported shape). */
int mm_gemm_tn_pair_splits(int M, int N0, int Kp0, int N1, int Kp1);
int mm_gemm_tn_pair(void* C0, void* a_colsum0, const void* A0, const void* B0, int N0, int Kp0, void* C1, void* a_colsum1,
                    const void* A1, const void* B1, int N1, int Kp1, int M, float* workspace, int splits, int dtype,
                    void* stream);

/* ------------------------------------------------------------------ MAGMA adapter, one kernel per direction
 * Replaces Adapter.forward and its backward (megatron/model/adapter.py:18-25,109-117):
 *   out = o + relu(o W1^T + b1) W2^T + b2,   o/out [rows, h], W1 [r, h], W2 [h, r], h1 = relu(.) [rows, r] (saved).
 * mm_adapter_bwd_dx: dpre = (dy W2) * (h1 > 0) [rows, r],  dx = dy + dpre W1 [rows, h]; W2 and W1 as stored (the kernel
 * reads them transposed on chip; the weight gradients are mm_gemm_tn_pair(dpre, o, dy, h1)).
 * Results are bit-identical to the same chain built from mm_gemm_nt.  Fused shapes: mm_adapter_supported(h, r) != 0
 * (currently (1024, 128) and (512, 64)); others return MM_EUNSUPPORTED - compose mm_gemm_nt instead. */
int mm_adapter_supported(int h, int r);
int mm_adapter_fwd(void* out, void* h1, const void* o, const void* w1, const void* b1, const void* w2, const void* b2,
                   int rows, int h, int r, int dtype, void* stream);
/* mm_adapter_fwd with the GPT-J residual tail of the layer on its output store (the MLP's adapter is the last kernel of a
 * MAGMA layer): out = [residual +] ((attn + attn_bias) + (adapter(o) + mlp_bias)), every intermediate rounded to `dtype`
 * as mm_gptj_combine does (megatron/model/transformer.py:747-766) - bit-identical to mm_adapter_fwd + mm_gptj_combine.
 * residual may be NULL (TP > 1: the all-reduce sits between); h1 as in mm_adapter_fwd. */
int mm_adapter_fwd_combine(void* out, void* h1, const void* o, const void* w1, const void* b1, const void* w2, const void* b2,
                           const void* residual, const void* attn, const void* attn_bias, const void* mlp_bias, int rows,
                           int h, int r, int dtype, void* stream);
int mm_adapter_bwd_dx(void* dx, void* dpre, const void* dy, const void* w2, const void* w1, const void* h1, int rows,
                      int h, int r, int dtype, void* stream);

/* ------------------------------------------------------------------ fused attention
 * Replaces ParallelSelfAttention.attention + the permute after it
 * (megatron/model/transformer.py:315-416, 623-630) without materialising [b*np, s, s].
 * qkv: [s, b, np, 3*hn] (the QKV linear's output, after rotary); ctx: [s, b, np*hn];
 * lse: [b, np, s] fp32 (log-sum-exp of scaled scores, saved for bwd).
 * mask_mode 0: implicit causal (-inf above the diagonal: scaled_upper_triang semantics);
 * mask_mode 1: explicit uint8 mask [mask_batches(1|b), 1, s, s], 1 = masked, filled with -10000
 *              (scaled_masked_softmax / eager gpt2_attention_mask_func semantics).
 * softmax_scale multiplies q.k (reference: 1/norm_factor, transformer.py:351).  hn in {64, 128}. */
int mm_attn_fwd(void* ctx, float* lse, const void* qkv, const uint8_t* mask, int mask_mode,
                int mask_batches, int s, int b, int np, int hn, float softmax_scale, int dtype,
                void* stream);
/* dqkv: [s, b, np, 3*hn] gradient wrt the (post-rotary) qkv; delta: [b, np, s] fp32 scratch. */
int mm_attn_bwd(void* dqkv, float* delta, const void* dctx, const void* ctx, const float* lse,
                const void* qkv, const uint8_t* mask, int mask_mode, int mask_batches, int s, int b,
                int np, int hn, float softmax_scale, int dtype, void* stream);

/* mm_attn_bwd with the backward of the rotary embedding fused into the dq / dk stores: dqkv is then the gradient wrt the
 * PRE-rotary QKV activation, i.e. mm_attn_bwd followed by mm_rotary_qk_inplace(inverse = 1) (same roundings, bit-identical)
 * without the extra pass (rotary backward of megatron/model/positional_embeddings.py:76-92 under autograd).
 * cos/sin: [>= rot_offset + s, rot_d] tables in `dtype`; rot_d in {0 (off), 16, 32, 64, 128}, <= hn. */
int mm_attn_bwd_rotary(void* dqkv, float* delta, const void* dctx, const void* ctx, const float* lse, const void* qkv,
                       const uint8_t* mask, int mask_mode, int mask_batches, int s, int b, int np, int hn,
                       float softmax_scale, const void* cos, const void* sin, int rot_d, int rot_offset, int dtype,
                       void* stream);

/* Attention dropout (megatron/model/transformer.py:301-302, 383-384: torch.nn.Dropout(attention_dropout) on the softmax output
 * inside mpu.get_cuda_rng_tracker().fork()): mm_attn_fwd / mm_attn_bwd_rotary with the probabilities dropped with probability p
 * and the kept ones scaled by 1 / (1 - p) between the softmax and the P.V product.  The keep mask is never stored: the forward
 * and both backward kernels regenerate it from rng_state (device uint64[2] {seed, base offset}, as mm_bias_dropout_add_devrng:
 * a replayed HIP graph draws fresh masks), the launch's position `pos` inside the step and the element's coordinates
 * (batch * np + head, q, key) - the definition is in csrc/attention.hip and mm_attn_dropout_mask writes that mask as bytes
 * keep[b, np, s, s] (1 = kept).  p is quantised to 1/65536.  The backward call must repeat the forward's p, rng_state contents
 * and pos.  The reference draws its mask from the device generator: the two random streams cannot match element for element. */
int mm_attn_fwd_dropout(void* ctx, float* lse, const void* qkv, const uint8_t* mask, int mask_mode, int mask_batches, int s, int b,
                        int np, int hn, float softmax_scale, float p, const void* rng_state, unsigned long long pos, int dtype,
                        void* stream);
int mm_attn_bwd_dropout(void* dqkv, float* delta, const void* dctx, const void* ctx, const float* lse, const void* qkv,
                        const uint8_t* mask, int mask_mode, int mask_batches, int s, int b, int np, int hn, float softmax_scale,
                        const void* cos, const void* sin, int rot_d, int rot_offset, float p, const void* rng_state,
                        unsigned long long pos, int dtype, void* stream);
int mm_attn_dropout_mask(uint8_t* keep, int s, int b, int np, float p, const void* rng_state, unsigned long long pos, void* stream);

/* ------------------------------------------------------------------ cross entropy (TP = 1)
 * Replaces vocab_parallel_cross_entropy(output.float(), labels) (megatron/mpu/cross_entropy.py:27-112 as called from
 * megatron/model/gpt2_model.py:52-70) for one tensor-parallel rank: loss[t] = logsumexp(logits[t, :]) - logits[t, label[t]],
 * fp32 math on 16-bit logits [T, V]; lse[T] is saved for the backward. */
int mm_cross_entropy_fwd(float* loss, float* lse, const void* logits, const int64_t* labels, int T, int V,
                         int dtype, void* stream);
/* dlogits[t, v] = (softmax(logits[t])[v] - [v == label[t]]) * row_scale[t]; may alias logits. */
int mm_cross_entropy_bwd(void* dlogits, const void* logits, const int64_t* labels, const float* lse,
                         const float* row_scale, int T, int V, int dtype, void* stream);
/* Loss and gradient in one pass, for the masked-mean loss of megatron/model/gpt2_model.py:52-70 whose per-row gradient
 * weight row_weight[t] = loss_mask[t] / sum(loss_mask) is known at forward time: loss[t] and lse[t] as mm_cross_entropy_fwd,
 * and logits[t, :] is OVERWRITTEN with (softmax(logits[t]) - onehot(label[t])) * row_weight[t] (the caller scales what it
 * derives from that by the loss's upstream gradient).  A block keeps its row in registers: V <= mm_cross_entropy_fwd_grad_max_v()
 * (65536), larger V returns MM_EUNSUPPORTED. */
int mm_cross_entropy_fwd_grad(float* loss, float* lse, void* logits, const int64_t* labels, const float* row_weight,
                              int T, int V, int dtype, void* stream);
int mm_cross_entropy_fwd_grad_max_v(void);
/* The masked mean around the per-token losses (megatron/model/gpt2_model.py:64-70: `loss_mask.view(-1)`,
 * `torch.sum(losses.view(-1) * loss_mask) / loss_mask.sum()`), fp32 [T] vectors, one block each, fixed summation order:
 * mm_loss_mask_weights: den[0] = sum(loss_mask), row_weight[t] = loss_mask[t] / den (mm_cross_entropy_fwd_grad's weights);
 * mm_masked_mean: out[0] = sum(losses * loss_mask) / den[0]. */
int mm_loss_mask_weights(float* row_weight, float* den, const float* loss_mask, int T, void* stream);
int mm_masked_mean(float* out, const float* losses, const float* loss_mask, const float* den, int T, void* stream);
/* out = x * scalar[0] on n 16-bit elements (scalar: DEVICE fp32): applies the upstream gradient of a fused loss node
 * (autograd's grad_output of the scalar loss, megatron/training.py:706-712 `model.backward(loss)`) without a host sync. */
int mm_scale_dev(void* out, const void* x, const float* scalar, size_t n, int dtype, void* stream);

/* ------------------------------------------------------------------ embedding + image prefix
 * The step in front of the first layer in one pass: word-embedding gather, `cat((image_prefix, text[:, :s-P]), dim=1)` and
 * the [b, s, h] -> [s, b, h] transpose (megatron/model/word_embeddings.py:135-205, gpt2_model.py:73-77):
 *   out[si, bi, :] = si < P ? prefix[bi, si, :] : table[tokens[bi, si - P], :]
 * out [s, b, h], table [V, h], prefix [b, P, h] (NULL when P == 0) - 2-byte elements of any 16-bit dtype (pure copy);
 * tokens [b, s] int64 (ids clamped to [0, V)). */
int mm_embed_prefix(void* out, const void* table, const int64_t* tokens, const void* prefix, int s, int b, int h, int P,
                    int V, void* stream);
/* ImagePrefix.forward behind its encoder (megatron/model/image_prefix.py:232-252): proj = nn.Linear(enc_dim, out_dim),
 * nn.Dropout(p), nn.LayerNorm(out_dim) on x [rows = b * P, enc_dim] in ONE launch for the few rows of a step (rows <= 16; the
 * pooled openclip-H encoder gives P = 1, image_prefix.py:106).  out [rows, out_dim]; ln_in (optional) receives the LayerNorm's
 * input (post-dropout), keep_mask (optional, bytes, 1 = kept) the dropout mask and mean / rstd the statistics - what the
 * backward (mm_layernorm_bwd, mm_dropout_bwd, the GEMM entry points) needs.  Dropout draws from the device-resident Philox
 * state exactly as mm_bias_dropout_add_devrng(rng_state, offset) does on the same [rows, out_dim] tensor.  workspace:
 * mm_prefix_tail_workspace_bytes(rows, out_dim) bytes of device memory (its first 256 bytes are zeroed on `stream` by the
 * call: the ticket of the workgroup that arrives last and normalises the rows).  Shapes outside mm_prefix_tail_supported
 * return MM_EINVAL: run mm_gemm_nt + mm_bias_dropout_add + mm_layernorm_fwd instead. */
int mm_prefix_tail_supported(int rows, int enc_dim, int out_dim);
size_t mm_prefix_tail_workspace_bytes(int rows, int out_dim);
int mm_prefix_tail_fwd(void* out, void* ln_in, uint8_t* keep_mask, float* mean, float* rstd, const void* x, const void* w,
                       const void* bias, const void* ln_w, const void* ln_b, void* workspace, int rows, int enc_dim, int out_dim,
                       float eps, float p, int training, const uint64_t* rng_state, uint64_t offset, int dtype, void* stream);
/* dx = keep_mask ? g / (1 - p) : 0 (n elements, n % 8 == 0): backward of nn.Dropout for a mask written by
 * mm_prefix_tail_fwd / mm_bias_dropout_add. */
int mm_dropout_bwd(void* dx, const void* g, const uint8_t* keep_mask, size_t n, float p, int dtype, void* stream);

/* ------------------------------------------------------------------ optimizer step (part of the timed step)
 * Mixed-precision Adam with global-norm gradient clipping over a list of 16-bit tensors, two launches per step.  States
 * the update rule of the optimizer the recipe constructs (apex / DeepSpeed FusedAdam in adam_w_mode under DeepSpeed's
 * 16-bit wrapper with fp32 master weights, megatron/training.py:571-580,659-668; gradient_clipping,
 * configs/magma_pythia_410M.yml:57) - those libraries are not in the reference tree: parity unpinned.
 *   params / grads : DEVICE arrays [n_tensors] of device pointers (16-bit weights / their gradients; a NULL gradient
 *                    leaves that tensor untouched and contributes 0 to the norm);
 *   master, exp_avg, exp_avg_sq : flat fp32 [sum of tensor sizes, each padded to a multiple of 8];
 *   chunks : DEVICE array [nchunks] of {int32 tensor, int32 count, int64 tensor_offset, int64 flat_offset}, one per block,
 *            count <= mm_adam_chunk_elems(), tensor_offset and flat_offset multiples of 8 elements;
 *   partial : fp32 [nchunks] workspace; step : device int32, incremented by the call (bias correction uses the new value);
 *   grad_norm_out : optional device float, receives the pre-clip global L2 norm; max_grad_norm <= 0 disables clipping;
 *   lr_dev : optional device float; when non-NULL the learning rate is read from it at run time and `lr` is ignored (a step
 *            captured in a HIP graph then follows a scheduler that rewrites the device value between replays).
 * Deterministic (no float atomics); no allocation, no sync. */
int mm_adam_chunk_elems(void);
int mm_adam_step(void* const* params, const void* const* grads, float* master, float* exp_avg, float* exp_avg_sq,
                 const void* chunks, int nchunks, float* partial, int* step, float* grad_norm_out, float lr,
                 const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm, int bias_correction, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MM_HIP_H */
