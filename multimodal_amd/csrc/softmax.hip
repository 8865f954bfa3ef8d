// softmax.hip - scaled (upper-triangular | explicitly masked) softmax, forward and backward.
//
// MI355X design: one wave64 per row, the whole row (<= 2048 elements) held in registers as
// ITERS chunks of 8 elements per lane (16-byte global accesses), max/sum by wave-level
// __shfl_xor butterflies, fp32 accumulation, 4 rows per 256-thread block.  The causal variant
// only reads the r+1 live elements of row r (half the read traffic of the square matrix) and
// writes zeros above the diagonal.
//
// Semantics follow the reference kernels (behaviour, not code):
//   megatron/fused_kernels/scaled_upper_triang_masked_softmax.h:143-361
//   megatron/fused_kernels/scaled_masked_softmax.h:117-352
#include "common.h"

namespace mm {

constexpr int kRowsPerBlock = 4;

// ---------------------------------------------------------------- vector (s % 8 == 0) kernels
// MODE 0: causal (row r keeps cols <= r, rest -inf);  MODE 1: byte mask (1 -> -10000)
template <typename T, int ITERS, int MODE>
__global__ __launch_bounds__(256) void softmax_fwd_vec(u16* __restrict__ out, const u16* __restrict__ in,
                                                       const uint8_t* __restrict__ mask, float scale,
                                                       long long nrows, int sq, int sk, int np, int pad_batches) {
    const int lane = threadIdx.x & 63;
    const long long g = (long long)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (g >= nrows) return;
    const int r = (int)(g % sq);
    const int len = (MODE == 0) ? (r + 1) : sk;
    const u16* src = in + g * sk;
    u16* dst = out + g * sk;
    const uint8_t* mrow = nullptr;
    if (MODE == 1) {
        long long mr = (pad_batches == 1) ? (long long)r : ((g / ((long long)np * sq)) * sq + r);
        mrow = mask + mr * sk;
    }
    float v[ITERS][8];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int base = (it * 64 + lane) * 8;
        if (base < len) {
            u32x4 raw = *reinterpret_cast<const u32x4*>(src + base);
            unpack8<T>(raw, v[it]);
            if (MODE == 1) {
                u32x2 mraw = *reinterpret_cast<const u32x2*>(mrow + base);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    uint32_t mb = (mraw[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    v[it][j] = (mb != 1u) ? v[it][j] * scale : -10000.0f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[it][j] = (base + j < len) ? v[it][j] * scale : -INFINITY;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, v[it][j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[it][j] = -INFINITY;
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float e = __expf(v[it][j] - mx);  // exp(-inf) = 0 for the dead elements
            v[it][j] = e;
            sum += e;
        }
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int base = (it * 64 + lane) * 8;
        if (base < sk) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (base + j < len) ? v[it][j] / sum : 0.f;
            *reinterpret_cast<u32x4*>(dst + base) = pack8<T>(o);
        }
    }
}

// in place: grad <- scale * (grad*y - y * sum(grad*y)) over the live part of the row
template <typename T, int ITERS, int MODE>
__global__ __launch_bounds__(256) void softmax_bwd_vec(u16* __restrict__ grad, const u16* __restrict__ y,
                                                       float scale, long long nrows, int sq, int sk) {
    const int lane = threadIdx.x & 63;
    const long long g = (long long)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (g >= nrows) return;
    const int r = (int)(g % sq);
    const int len = (MODE == 0) ? (r + 1) : sk;
    u16* gp = grad + g * sk;
    const u16* yp = y + g * sk;
    float gy[ITERS][8], yy[ITERS][8];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int base = (it * 64 + lane) * 8;
        if (base < len) {
            u32x4 graw = *reinterpret_cast<const u32x4*>(gp + base);
            u32x4 yraw = *reinterpret_cast<const u32x4*>(yp + base);
            unpack8<T>(graw, gy[it]);
            unpack8<T>(yraw, yy[it]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool live = base + j < len;
                yy[it][j] = live ? yy[it][j] : 0.f;
                gy[it][j] = live ? gy[it][j] * yy[it][j] : 0.f;
                sum += gy[it][j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) gy[it][j] = yy[it][j] = 0.f;
        }
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int base = (it * 64 + lane) * 8;
        if (base < sk) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = scale * (gy[it][j] - yy[it][j] * sum);
            *reinterpret_cast<u32x4*>(gp + base) = pack8<T>(o);
        }
    }
}

// ---------------------------------------------------------------- scalar (any sk <= 2048) kernels
template <typename T, int MODE>
__global__ __launch_bounds__(256) void softmax_fwd_scalar(u16* __restrict__ out, const u16* __restrict__ in,
                                                          const uint8_t* __restrict__ mask, float scale,
                                                          long long nrows, int sq, int sk, int np, int pad_batches) {
    const int lane = threadIdx.x & 63;
    const long long g = (long long)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (g >= nrows) return;
    const int r = (int)(g % sq);
    const int len = (MODE == 0) ? (r + 1) : sk;
    const u16* src = in + g * sk;
    u16* dst = out + g * sk;
    const uint8_t* mrow = nullptr;
    if (MODE == 1) {
        long long mr = (pad_batches == 1) ? (long long)r : ((g / ((long long)np * sq)) * sq + r);
        mrow = mask + mr * sk;
    }
    float v[32];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < 32; ++it) {
        const int e = it * 64 + lane;
        float x = -INFINITY;
        if (e < len) {
            x = to_f32<T>(src[e]) * scale;
            if (MODE == 1 && mrow[e] == 1) x = -10000.0f;
        }
        v[it] = x;
        mx = fmaxf(mx, x);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < 32; ++it) {
        v[it] = __expf(v[it] - mx);
        sum += v[it];
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int it = 0; it < 32; ++it) {
        const int e = it * 64 + lane;
        if (e < sk) dst[e] = from_f32<T>(e < len ? v[it] / sum : 0.f);
    }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void softmax_bwd_scalar(u16* __restrict__ grad, const u16* __restrict__ y,
                                                          float scale, long long nrows, int sq, int sk) {
    const int lane = threadIdx.x & 63;
    const long long g = (long long)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (g >= nrows) return;
    const int r = (int)(g % sq);
    const int len = (MODE == 0) ? (r + 1) : sk;
    u16* gp = grad + g * sk;
    const u16* yp = y + g * sk;
    float gy[32], yy[32];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < 32; ++it) {
        const int e = it * 64 + lane;
        float a = 0.f, b = 0.f;
        if (e < len) {
            b = to_f32<T>(yp[e]);
            a = to_f32<T>(gp[e]) * b;
        }
        gy[it] = a;
        yy[it] = b;
        sum += a;
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int it = 0; it < 32; ++it) {
        const int e = it * 64 + lane;
        if (e < sk) gp[e] = from_f32<T>(scale * (gy[it] - yy[it] * sum));
    }
}

template <typename T, int MODE>
static int launch_fwd(void* out, const void* in, const uint8_t* mask, float scale, long long nrows, int sq, int sk,
                      int np, int pad_batches, hipStream_t st) {
    dim3 grid((unsigned)((nrows + kRowsPerBlock - 1) / kRowsPerBlock)), block(256);
    u16* o = (u16*)out;
    const u16* i = (const u16*)in;
    const bool vec = (sk % 8 == 0) && aligned16(out) && aligned16(in) && (MODE == 0 || (((uintptr_t)mask) & 7) == 0);
    if (!vec) {
        softmax_fwd_scalar<T, MODE><<<grid, block, 0, st>>>(o, i, mask, scale, nrows, sq, sk, np, pad_batches);
    } else if (sk <= 512) {
        softmax_fwd_vec<T, 1, MODE><<<grid, block, 0, st>>>(o, i, mask, scale, nrows, sq, sk, np, pad_batches);
    } else if (sk <= 1024) {
        softmax_fwd_vec<T, 2, MODE><<<grid, block, 0, st>>>(o, i, mask, scale, nrows, sq, sk, np, pad_batches);
    } else {
        softmax_fwd_vec<T, 4, MODE><<<grid, block, 0, st>>>(o, i, mask, scale, nrows, sq, sk, np, pad_batches);
    }
    MM_CHECK_LAUNCH();
    return MM_OK;
}

template <typename T, int MODE>
static int launch_bwd(void* grad, const void* y, float scale, long long nrows, int sq, int sk, hipStream_t st) {
    dim3 grid((unsigned)((nrows + kRowsPerBlock - 1) / kRowsPerBlock)), block(256);
    u16* g = (u16*)grad;
    const u16* yy = (const u16*)y;
    const bool vec = (sk % 8 == 0) && aligned16(grad) && aligned16(y);
    if (!vec) {
        softmax_bwd_scalar<T, MODE><<<grid, block, 0, st>>>(g, yy, scale, nrows, sq, sk);
    } else if (sk <= 512) {
        softmax_bwd_vec<T, 1, MODE><<<grid, block, 0, st>>>(g, yy, scale, nrows, sq, sk);
    } else if (sk <= 1024) {
        softmax_bwd_vec<T, 2, MODE><<<grid, block, 0, st>>>(g, yy, scale, nrows, sq, sk);
    } else {
        softmax_bwd_vec<T, 4, MODE><<<grid, block, 0, st>>>(g, yy, scale, nrows, sq, sk);
    }
    MM_CHECK_LAUNCH();
    return MM_OK;
}

}  // namespace mm

using namespace mm;

extern "C" int mm_softmax_causal_fwd(void* out, const void* in, float scale, int attn_batches, int s, int dtype,
                                     void* stream) {
    MM_CHECK_ARG(out && in, "mm_softmax_causal_fwd: null pointer");
    MM_CHECK_ARG(attn_batches > 0 && s > 0, "mm_softmax_causal_fwd: attn_batches=%d s=%d must be positive", attn_batches, s);
    MM_CHECK_ARG(s <= 2048, "mm_softmax_causal_fwd: seq_len %d > 2048 is not supported", s);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        return launch_fwd<T, 0>(out, in, nullptr, scale, (long long)attn_batches * s, s, s, 1, 1, (hipStream_t)stream);
    });
}

extern "C" int mm_softmax_causal_bwd(void* grad_inout, const void* y, float scale, int attn_batches, int s, int dtype,
                                     void* stream) {
    MM_CHECK_ARG(grad_inout && y, "mm_softmax_causal_bwd: null pointer");
    MM_CHECK_ARG(attn_batches > 0 && s > 0, "mm_softmax_causal_bwd: attn_batches=%d s=%d must be positive", attn_batches, s);
    MM_CHECK_ARG(s <= 2048, "mm_softmax_causal_bwd: seq_len %d > 2048 is not supported", s);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        return launch_bwd<T, 0>(grad_inout, y, scale, (long long)attn_batches * s, s, s, (hipStream_t)stream);
    });
}

extern "C" int mm_softmax_masked_fwd(void* out, const void* in, const uint8_t* mask, float scale, int b, int np, int sq,
                                     int sk, int pad_batches, int dtype, void* stream) {
    MM_CHECK_ARG(out && in && mask, "mm_softmax_masked_fwd: null pointer");
    MM_CHECK_ARG(b > 0 && np > 0 && sq > 0 && sk > 0, "mm_softmax_masked_fwd: sizes must be positive");
    MM_CHECK_ARG(sk <= 2048, "mm_softmax_masked_fwd: key seq_len %d > 2048 is not supported", sk);
    MM_CHECK_ARG(sq > 1, "mm_softmax_masked_fwd: query seq_len must be > 1");
    MM_CHECK_ARG(pad_batches == 1 || pad_batches == b, "mm_softmax_masked_fwd: mask batch %d must be 1 or %d", pad_batches, b);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        return launch_fwd<T, 1>(out, in, mask, scale, (long long)b * np * sq, sq, sk, np, pad_batches, (hipStream_t)stream);
    });
}

extern "C" int mm_softmax_masked_bwd(void* grad_inout, const void* y, float scale, int b, int np, int sq, int sk,
                                     int dtype, void* stream) {
    MM_CHECK_ARG(grad_inout && y, "mm_softmax_masked_bwd: null pointer");
    MM_CHECK_ARG(b > 0 && np > 0 && sq > 0 && sk > 0, "mm_softmax_masked_bwd: sizes must be positive");
    MM_CHECK_ARG(sk <= 2048, "mm_softmax_masked_bwd: key seq_len %d > 2048 is not supported", sk);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        return launch_bwd<T, 1>(grad_inout, y, scale, (long long)b * np * sq, sq, sk, (hipStream_t)stream);
    });
}

// Host-only integer: the reference formula with the wave size the ROCm build sees (64).
extern "C" int mm_softmax_batch_per_block(int sq, int sk, int b, int np) {
    (void)sq; (void)b; (void)np;
    int log2e = 0;
    while ((1 << log2e) < sk) ++log2e;
    const int pow2 = 1 << log2e;
    const int warp = pow2 < kWave ? pow2 : kWave;
    const int per_warp = pow2 <= 128 ? 2 : 1;
    return (128 / warp) * per_warp;
}
