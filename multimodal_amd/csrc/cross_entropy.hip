// cross_entropy.hip - token-level cross entropy over [T, V] 16-bit logits, fp32 math (HBM-bound).
//
// Replaces `vocab_parallel_cross_entropy(output.float(), labels)` at TP = 1 as used by the reference's loss
// (megatron/model/gpt2_model.py:52-70, megatron/mpu/cross_entropy.py:27-112) without materialising the fp32 copy
// of the logits (1.6 GB at the 410M config) or the fp32 softmax: forward is one read of the logits, backward one
// read + one write.  One 256-thread block per token row, 16-byte loads, online max/sum, wave butterflies + LDS.
#include "common.h"

namespace mm {

__device__ __forceinline__ void online_merge(float& m, float& l, float m2, float l2) {
    const float mn = fmaxf(m, m2);
    if (mn == -INFINITY) return;  // both sides empty (threads beyond V/8 chunks): exp(-inf + inf) would be NaN
    l = l * __expf(m - mn) + l2 * __expf(m2 - mn);
    m = mn;
}

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_k(float* __restrict__ loss, float* __restrict__ lse_out,
                                                const u16* __restrict__ logits, const long long* __restrict__ labels, int V) {
    __shared__ float sm[4], sl[4];
    const size_t row = blockIdx.x;
    const u16* x = logits + row * (size_t)V;
    float m = -INFINITY, l = 0.f;
    const int nch = V >> 3;
    auto fold = [&](const u32x4& raw) {
        float v[8];
        unpack8<T>(raw, v);
        float cm = v[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) cm = fmaxf(cm, v[j]);
        const float mn = fmaxf(m, cm);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += __expf(v[j] - mn);
        l = l * __expf(m - mn) + acc;
        m = mn;
    };
    // four 16-byte loads in flight per lane (a 100 KB row per block: the loop is bound by bytes in flight, not by the exps)
    int c = threadIdx.x;
    for (; c + 3 * 256 < nch; c += 4 * 256) {
        const u32x4 r0 = *reinterpret_cast<const u32x4*>(x + (size_t)c * 8);
        const u32x4 r1 = *reinterpret_cast<const u32x4*>(x + (size_t)(c + 256) * 8);
        const u32x4 r2 = *reinterpret_cast<const u32x4*>(x + (size_t)(c + 512) * 8);
        const u32x4 r3 = *reinterpret_cast<const u32x4*>(x + (size_t)(c + 768) * 8);
        fold(r0);
        fold(r1);
        fold(r2);
        fold(r3);
    }
    for (; c < nch; c += 256) fold(*reinterpret_cast<const u32x4*>(x + (size_t)c * 8));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
        online_merge(m, l, m2, l2);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sm[wave] = m; sl[wave] = l; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = sm[0], L = sl[0];
        for (int w = 1; w < 4; ++w) online_merge(M, L, sm[w], sl[w]);
        const float lse = M + __logf(L);
        const long long lab = labels[row];
        const float tgt = (lab >= 0 && lab < V) ? to_f32<T>(x[lab]) : 0.f;
        lse_out[row] = lse;
        loss[row] = lse - tgt;
    }
}

// dlogits[row, v] = (exp(x - lse) - [v == label]) * row_scale[row]
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_k(u16* __restrict__ dlogits, const u16* __restrict__ logits,
                                                const long long* __restrict__ labels, const float* __restrict__ lse,
                                                const float* __restrict__ row_scale, int V) {
    const size_t row = blockIdx.x;
    const u16* x = logits + row * (size_t)V;
    u16* dx = dlogits + row * (size_t)V;
    const float l = lse[row], sc = row_scale[row];
    const int lab = (int)labels[row];
    for (int c = threadIdx.x; c < (V >> 3); c += 256) {
        float v[8], o[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(x + c * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = __expf(v[j] - l);
            o[j] = (p - ((c * 8 + j) == lab ? 1.f : 0.f)) * sc;
        }
        *reinterpret_cast<u32x4*>(dx + c * 8) = pack8<T>(o);
    }
}

// Forward and gradient in ONE pass over the logits, for callers that know the per-row gradient weight at forward time (the
// masked-mean loss of gpt2_model.py:52-70: w[row] = mask[row] / sum(mask)): the block keeps its whole row in registers
// (NT threads x MAXCH 16-byte chunks), so the logits are read once and the gradient (softmax - onehot) * w[row] is written
// over them.  Against ce_fwd_k + ce_bwd_k this drops one of the two reads of the [T, V] tensor (786 MiB at the 410M shapes).
template <typename T, int NT, int MAXCH>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void ce_fwd_grad_k(float* __restrict__ loss, float* __restrict__ lse_out, u16* __restrict__ logits,
                                                    const long long* __restrict__ labels, const float* __restrict__ row_w, int V) {
    constexpr int NW = NT / 64;
    __shared__ float sred[NW];
    __shared__ float sbc[2];
    const size_t row = blockIdx.x;
    u16* x = logits + row * (size_t)V;
    const int nch = V >> 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 raw[MAXCH];
    // the row stays PACKED between the three passes (64 registers): without this the compiler keeps the unpacked floats of
    // one pass alive for the next and the kernel drops to one block per CU
    auto repack = [&]() {
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) asm volatile("" : "+v"(raw[i][0]), "+v"(raw[i][1]), "+v"(raw[i][2]), "+v"(raw[i][3]));
    };
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {  // every load of the row is in flight before the first use
        const int c = threadIdx.x + i * NT;
        if (c < nch) raw[i] = *reinterpret_cast<const u32x4*>(x + (size_t)c * 8);
    }
    const long long lab = labels[row];
    const float w = row_w[row];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        if (threadIdx.x + i * NT < nch) {
            float v[8];
            unpack8<T>(raw[i], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) m = fmaxf(m, v[j]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) sred[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = sred[0];
        for (int k = 1; k < NW; ++k) M = fmaxf(M, sred[k]);
        sbc[0] = M;
    }
    __syncthreads();
    m = sbc[0];
    repack();
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        if (threadIdx.x + i * NT < nch) {
            float v[8];
            unpack8<T>(raw[i], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) l += __expf(v[j] - m);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
    if (lane == 0) sred[wave] = l;  // every wave has read sbc[0]/sred of the max round before the barrier above
    __syncthreads();
    if (threadIdx.x == 0) {
        float L = sred[0];
        for (int k = 1; k < NW; ++k) L += sred[k];
        const float lse = m + __logf(L);
        const float tgt = (lab >= 0 && lab < V) ? to_f32<T>(x[lab]) : 0.f;  // x[lab] is overwritten only after the next barrier
        lse_out[row] = lse;
        loss[row] = lse - tgt;
        sbc[1] = lse;
    }
    __syncthreads();
    const float lse = sbc[1];
    repack();
    const int labi = (int)lab;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = threadIdx.x + i * NT;
        if (c < nch) {
            float v[8], o[8];
            unpack8<T>(raw[i], v);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__expf(v[j] - lse) - ((c * 8 + j) == labi ? 1.f : 0.f)) * w;
            *reinterpret_cast<u32x4*>(x + (size_t)c * 8) = pack8<T>(o);
        }
    }
}

// ---- the masked mean around the per-token losses (megatron/model/gpt2_model.py:52-70: sum(losses * mask) / mask.sum()) as
// two one-block launches instead of seven framework kernels: den = sum(mask), w[t] = mask[t] / den before the loss kernel
// (its per-row gradient weight), and loss = sum(losses * mask) / den after it.  Fixed summation order: deterministic.
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();  // red may still be read from a previous call
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += red[k];
    return t;
}
__global__ __launch_bounds__(1024) void loss_mask_weights_k(float* __restrict__ w, float* __restrict__ den_out,
                                                            const float* __restrict__ mask, int T) {
    __shared__ float red[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < T; i += 1024) a += mask[i];
    const float den = block_sum_1024(a, red);
    const float inv = 1.0f / den;
    for (int i = threadIdx.x; i < T; i += 1024) w[i] = inv * mask[i];
    if (threadIdx.x == 0) den_out[0] = den;
}
__global__ __launch_bounds__(1024) void masked_mean_k(float* __restrict__ out, const float* __restrict__ losses,
                                                      const float* __restrict__ mask, const float* __restrict__ den, int T) {
    __shared__ float red[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < T; i += 1024) a += losses[i] * mask[i];
    const float tot = block_sum_1024(a, red);
    if (threadIdx.x == 0) out[0] = tot / den[0];
}
// out = x * (*scalar): applies a fused loss node's upstream gradient (a device scalar) to a parameter gradient
template <typename T>
__global__ __launch_bounds__(256) void scale_dev_k(u16* __restrict__ out, const u16* __restrict__ x, const float* __restrict__ scalar, size_t n8) {
    const float sc = *scalar;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        float v[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(x + i * 8), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= sc;
        *reinterpret_cast<u32x4*>(out + i * 8) = pack8<T>(v);
    }
}

}  // namespace mm

using namespace mm;

extern "C" int mm_cross_entropy_fwd(float* loss, float* lse, const void* logits, const int64_t* labels, int T, int V,
                                    int dtype, void* stream) {
    MM_CHECK_ARG(loss && lse && logits && labels, "mm_cross_entropy_fwd: null pointer");
    MM_CHECK_ARG(T > 0 && V > 0 && V % 8 == 0, "mm_cross_entropy_fwd: T=%d V=%d (V must be a positive multiple of 8)", T, V);
    MM_CHECK_ARG(aligned16(logits), "mm_cross_entropy_fwd: logits must be 16-byte aligned");
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using TT = decltype(t);
        ce_fwd_k<TT><<<T, 256, 0, (hipStream_t)stream>>>(loss, lse, (const u16*)logits, (const long long*)labels, V);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

extern "C" int mm_cross_entropy_bwd(void* dlogits, const void* logits, const int64_t* labels, const float* lse,
                                    const float* row_scale, int T, int V, int dtype, void* stream) {
    MM_CHECK_ARG(dlogits && logits && labels && lse && row_scale, "mm_cross_entropy_bwd: null pointer");
    MM_CHECK_ARG(T > 0 && V > 0 && V % 8 == 0, "mm_cross_entropy_bwd: T=%d V=%d (V must be a positive multiple of 8)", T, V);
    MM_CHECK_ARG(aligned16(logits) && aligned16(dlogits), "mm_cross_entropy_bwd: logits/dlogits must be 16-byte aligned");
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using TT = decltype(t);
        ce_bwd_k<TT><<<T, 256, 0, (hipStream_t)stream>>>((u16*)dlogits, (const u16*)logits, (const long long*)labels, lse, row_scale, V);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

constexpr int CE_NT = 512, CE_MAXCH = 16;  // a row of up to 512 * 16 * 8 = 65536 logits lives in one block's registers

extern "C" int mm_cross_entropy_fwd_grad_max_v() { return CE_NT * CE_MAXCH * 8; }

extern "C" int mm_cross_entropy_fwd_grad(float* loss, float* lse, void* logits, const int64_t* labels, const float* row_weight,
                                         int T, int V, int dtype, void* stream) {
    MM_CHECK_ARG(loss && lse && logits && labels && row_weight, "mm_cross_entropy_fwd_grad: null pointer");
    MM_CHECK_ARG(T > 0 && V > 0 && V % 8 == 0, "mm_cross_entropy_fwd_grad: T=%d V=%d (V must be a positive multiple of 8)", T, V);
    MM_CHECK_ARG(aligned16(logits), "mm_cross_entropy_fwd_grad: logits must be 16-byte aligned");
    if (V > CE_NT * CE_MAXCH * 8) {
        set_error("mm_cross_entropy_fwd_grad: V=%d exceeds the %d logits a block keeps in registers (use mm_cross_entropy_fwd + _bwd)",
                  V, CE_NT * CE_MAXCH * 8);
        return MM_EUNSUPPORTED;
    }
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using TT = decltype(t);
        ce_fwd_grad_k<TT, CE_NT, CE_MAXCH><<<T, CE_NT, 0, (hipStream_t)stream>>>(loss, lse, (u16*)logits, (const long long*)labels,
                                                                                 row_weight, V);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

extern "C" int mm_loss_mask_weights(float* row_weight, float* den, const float* loss_mask, int T, void* stream) {
    MM_CHECK_ARG(row_weight && den && loss_mask && T > 0, "mm_loss_mask_weights: bad arguments");
    loss_mask_weights_k<<<1, 1024, 0, (hipStream_t)stream>>>(row_weight, den, loss_mask, T);
    MM_CHECK_LAUNCH();
    return MM_OK;
}

extern "C" int mm_masked_mean(float* out, const float* losses, const float* loss_mask, const float* den, int T, void* stream) {
    MM_CHECK_ARG(out && losses && loss_mask && den && T > 0, "mm_masked_mean: bad arguments");
    masked_mean_k<<<1, 1024, 0, (hipStream_t)stream>>>(out, losses, loss_mask, den, T);
    MM_CHECK_LAUNCH();
    return MM_OK;
}

extern "C" int mm_scale_dev(void* out, const void* x, const float* scalar, size_t n, int dtype, void* stream) {
    MM_CHECK_ARG(out && x && scalar, "mm_scale_dev: null pointer");
    MM_CHECK_ARG(n % 8 == 0 && aligned16(out) && aligned16(x), "mm_scale_dev: n must be a multiple of 8 and the tensors 16-byte aligned");
    const size_t n8 = n / 8;
    if (n8 == 0) return MM_OK;
    size_t g = (n8 + 255) / 256;
    if (g > 2048) g = 2048;
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using TT = decltype(t);
        scale_dev_k<TT><<<dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream>>>((u16*)out, (const u16*)x, scalar, n8);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

