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
