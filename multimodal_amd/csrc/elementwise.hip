// elementwise.hip - the HBM-bound glue of the layer: rotary (in place on the fused QKV activation),
// bias-GeLU fwd/bwd, bias-dropout-add, column sums (bias gradients), 2-D transpose, add.
// All kernels move 16 bytes per lane per access and grid-stride over <= 2048 blocks.
#include "common.h"
#include "philox.h"

namespace mm {

static inline int grid_for(size_t work_items, int block = 256) {
    size_t g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------ rotary
// qkv: [T = s*b, np, 3*hn]; rotate the first d dims of the q and k slices of every head, in place.
// One work item = 8 elements of the first half + the 8 partner elements of the second half.
// Rounding follows the eager reference: (x*cos) and (rotate_half(x)*sin) are each rounded to T, then summed.
template <typename T>
__global__ __launch_bounds__(256) void rotary_qk_k(u16* __restrict__ qkv, const u16* __restrict__ cos_t,
                                                   const u16* __restrict__ sin_t, int T_tokens, int b, int np, int hn,
                                                   int d, int offset, int inverse) {
    // No fma contraction here: for fp16 the compiler narrows these fp32 expressions to native half arithmetic and would then
    // fuse rnd(a*b) +- rnd(c*d) into one half fma, skipping a rounding the eager reference performs.
#pragma clang fp contract(off)
    const int half = d >> 1;
    const int cpr = half >> 3;  // 8-element chunks per half
    const size_t total = (size_t)T_tokens * np * 2 * cpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr);
        size_t r = i / cpr;
        const int qk = (int)(r % 2);
        r /= 2;
        const int head = (int)(r % np);
        const size_t tok = r / np;
        const int pos = (int)(tok / b) + offset;
        u16* base = qkv + (tok * np + head) * (size_t)(3 * hn) + (size_t)qk * hn + c * 8;
        const u16* cp = cos_t + (size_t)pos * d + c * 8;
        const u16* sp = sin_t + (size_t)pos * d + c * 8;
        float x1[8], x2[8], c1[8], c2[8], s1[8], s2[8], o1[8], o2[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(base), x1);
        unpack8<T>(*reinterpret_cast<const u32x4*>(base + half), x2);
        unpack8<T>(*reinterpret_cast<const u32x4*>(cp), c1);
        unpack8<T>(*reinterpret_cast<const u32x4*>(cp + half), c2);
        unpack8<T>(*reinterpret_cast<const u32x4*>(sp), s1);
        unpack8<T>(*reinterpret_cast<const u32x4*>(sp + half), s2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (!inverse) {
                // y1 = x1*c1 + (-x2)*s1 ; y2 = x2*c2 + x1*s2
                o1[j] = rnd<T>(x1[j] * c1[j]) + rnd<T>(-x2[j] * s1[j]);
                o2[j] = rnd<T>(x2[j] * c2[j]) + rnd<T>(x1[j] * s2[j]);
            } else {
                // transposed rotation: dx1 = g1*c1 + g2*s2 ; dx2 = g2*c2 - g1*s1
                o1[j] = rnd<T>(x1[j] * c1[j]) + rnd<T>(x2[j] * s2[j]);
                o2[j] = rnd<T>(x2[j] * c2[j]) - rnd<T>(x1[j] * s1[j]);
            }
        }
        *reinterpret_cast<u32x4*>(base) = pack8<T>(o1);
        *reinterpret_cast<u32x4*>(base + half) = pack8<T>(o2);
    }
}

// ------------------------------------------------------------------------------------ GeLU
__device__ __forceinline__ float gelu_tanh_f(float x) { return gelu_tanh_fast(x); }
__device__ __forceinline__ float gelu_tanh_grad_f(float x) { return gelu_tanh_grad_fast(x); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// MODE 0: tanh approximation on fp32 (bias + y) (the fused reference op); MODE 1: exact erf on the
// 16-bit-rounded sum (eager `F.gelu(y + bias)`).  BWD: out = g * gelu'(.)
template <typename T, int MODE, bool BWD>
__global__ __launch_bounds__(256) void bias_gelu_k(u16* __restrict__ out, const u16* __restrict__ g, const u16* __restrict__ y,
                                                   const u16* __restrict__ bias, size_t rows, int cols) {
    const int cpr = cols >> 3;
    const size_t total = rows * cpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr);
        float yv[8], bv[8], gv[8], o[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(y + i * 8), yv);
        unpack8<T>(*reinterpret_cast<const u32x4*>(bias + c * 8), bv);
        if (BWD) unpack8<T>(*reinterpret_cast<const u32x4*>(g + i * 8), gv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = yv[j] + bv[j];
            if (MODE == 1) x = rnd<T>(x);
            if (!BWD) o[j] = (MODE == 0) ? gelu_tanh_f(x) : gelu_erf_f(x);
            else o[j] = gv[j] * ((MODE == 0) ? gelu_tanh_grad_f(x) : gelu_erf_grad_f(x));
        }
        *reinterpret_cast<u32x4*>(out + i * 8) = pack8<T>(o);
    }
}

// ------------------------------------------------------------------------------------ bias-dropout-add
// out = residual + dropout(x + bias); eager rounding: (x+bias) rounded, scaled/masked, rounded, + residual rounded.
template <typename T, bool DROP>
__global__ __launch_bounds__(256) void bias_dropout_add_k(u16* __restrict__ out, const u16* __restrict__ x,
                                                          const u16* __restrict__ bias, const u16* __restrict__ res,
                                                          uint8_t* __restrict__ keep, size_t rows, int cols, float p,
                                                          uint64_t seed, uint64_t offset, const uint64_t* __restrict__ rng_state) {
    if (DROP && rng_state) {  // device-resident Philox state {seed, base offset}: a replayed HIP graph draws a fresh mask
        seed = rng_state[0];
        offset += rng_state[1];
    }
    const int cpr = cols >> 3;
    const size_t total = rows * cpr;
    const float inv_keep = DROP ? 1.0f / (1.0f - p) : 1.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr);
        float xv[8], bv[8], rv[8], o[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(x + i * 8), xv);
        if (bias) unpack8<T>(*reinterpret_cast<const u32x4*>(bias + c * 8), bv);
        if (res) unpack8<T>(*reinterpret_cast<const u32x4*>(res + i * 8), rv);
        uint32_t rnd_bits[8];
        if (DROP) {
            uint32_t r0[4], r1[4];
            Philox::gen(seed, i, offset, r0);
            Philox::gen(seed, i, offset + 0x8000000000000000ull, r1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { rnd_bits[j] = r0[j]; rnd_bits[4 + j] = r1[j]; }
        }
        uint32_t keep_lo = 0, keep_hi = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = bias ? rnd<T>(xv[j] + bv[j]) : xv[j];
            if (DROP) {
                const float u = (float)(rnd_bits[j] >> 8) * (1.0f / 16777216.0f);  // [0,1)
                const bool k = u >= p;
                v = k ? rnd<T>(v * inv_keep) : 0.f;
                if (j < 4) keep_lo |= (k ? 1u : 0u) << (8 * j); else keep_hi |= (k ? 1u : 0u) << (8 * (j - 4));
            }
            o[j] = res ? v + rv[j] : v;
        }
        *reinterpret_cast<u32x4*>(out + i * 8) = pack8<T>(o);
        if (DROP && keep) *reinterpret_cast<u32x2*>(keep + i * 8) = u32x2{keep_lo, keep_hi};
    }
}

// ------------------------------------------------------------------------------------ column sums
// partial[blk][col] = sum over the block's row slice; one wave per block, 8 columns per lane.
template <typename T>
__global__ __launch_bounds__(64) void colsum_partial_k(float* __restrict__ partial, const u16* __restrict__ x, int rows,
                                                       int cols, int nblk) {
    const int col = (blockIdx.x * 64 + threadIdx.x) * 8;
    if (col >= cols) return;
    const int blk = blockIdx.y;
    const int per = (rows + nblk - 1) / nblk;
    const int r0 = blk * per, r1 = min(rows, r0 + per);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = r0; r < r1; ++r) {
        float v[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(x + (size_t)r * cols + col), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
    float* p = partial + (size_t)blk * cols + col;
    *reinterpret_cast<f32x4*>(p) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
}
// out[col] = sum_blk partial[blk][col]: 64 columns x 4 row-groups per block, LDS combine
template <typename T>
__global__ __launch_bounds__(256) void colsum_final_k(u16* __restrict__ out, const float* __restrict__ partial, int nblk, int cols) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx;
    float s0 = 0.f, s1 = 0.f;
    if (col < cols) {
        int bI = g;
        for (; bI + 4 < nblk; bI += 8) {
            s0 += partial[(size_t)bI * cols + col];
            s1 += partial[(size_t)(bI + 4) * cols + col];
        }
        if (bI < nblk) s0 += partial[(size_t)bI * cols + col];
    }
    red[g][cx] = s0 + s1;
    __syncthreads();
    if (g == 0 && col < cols) out[col] = from_f32<T>(red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx]);
}

// ------------------------------------------------------------------------------------ transpose (2-byte elements)
// 64x64 tiles through LDS; out[c][r] = in[r][c].
__global__ __launch_bounds__(256) void transpose2d_k(u16* __restrict__ out, const u16* __restrict__ in, int rows, int cols) {
    __shared__ u16 tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        const int r = r0 + rr, c = c0 + tx;
        tile[rr][tx] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : (u16)0;
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc, r = r0 + tx;
        if (c < cols && r < rows) out[(size_t)c * rows + r] = tile[tx][cc];
    }
}

// The same with 16-byte global accesses (rows % 8 == 0 and cols % 8 == 0, 16-byte aligned bases): the 2-byte kernel above moves
// 128 B per wave instruction and reaches ~0.4 TB/s on a [9216, 2304] weight; activations of a large model's weight gradients
// ([8192, 9216]: 151 MiB) go through here twice per layer.  Rows of the tile are written to LDS as 16-byte pieces (row pitch
// 144 B: 16-byte aligned, and 8 consecutive rows of one column fall into different banks), columns are gathered back as 8
// two-byte reads per 16-byte output piece.
__global__ __launch_bounds__(256) void transpose2d_vec_k(u16* __restrict__ out, const u16* __restrict__ in, int rows, int cols) {
    __shared__ __attribute__((aligned(16))) u16 tile[64][72];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int rr = (tid >> 3) + 32 * pass, ch = tid & 7;
        const int r = r0 + rr, c = c0 + ch * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r < rows && c < cols) v = *reinterpret_cast<const u32x4*>(in + (size_t)r * cols + c);
        *reinterpret_cast<u32x4*>(&tile[rr][ch * 8]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int cc = (tid >> 3) + 32 * pass, rch = tid & 7;  // output row cc (an input column), 8 input rows rch*8..
        const int c = c0 + cc, r = r0 + rch * 8;
        if (c < cols && r < rows) {
            u32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = (uint32_t)tile[rch * 8 + 2 * j][cc] | ((uint32_t)tile[rch * 8 + 2 * j + 1][cc] << 16);
            *reinterpret_cast<u32x4*>(out + (size_t)c * rows + r) = v;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_k(u16* __restrict__ out, const u16* __restrict__ a, const u16* __restrict__ b,
                                             const u16* __restrict__ c, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        float av[8], bv[8], cv[8], o[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(a + i * 8), av);
        unpack8<T>(*reinterpret_cast<const u32x4*>(b + i * 8), bv);
        if (c) unpack8<T>(*reinterpret_cast<const u32x4*>(c + i * 8), cv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = c ? av[j] + bv[j] + cv[j] : av[j] + bv[j];
        *reinterpret_cast<u32x4*>(out + i * 8) = pack8<T>(o);
    }
}

}  // namespace mm

using namespace mm;

extern "C" int mm_rotary_qk_inplace(void* qkv, const void* cos, const void* sin, int s, int b, int np, int hn, int d,
                                    int offset, int inverse, int dtype, void* stream) {
    MM_CHECK_ARG(qkv && cos && sin, "mm_rotary_qk_inplace: null pointer");
    MM_CHECK_ARG(s > 0 && b > 0 && np > 0 && hn > 0, "mm_rotary_qk_inplace: sizes must be positive");
    MM_CHECK_ARG(d > 0 && d <= hn && d % 16 == 0, "mm_rotary_qk_inplace: rotary dims %d must be a multiple of 16 and <= hn=%d", d, hn);
    MM_CHECK_ARG(hn % 8 == 0, "mm_rotary_qk_inplace: hn=%d must be a multiple of 8", hn);
    MM_CHECK_ARG(offset >= 0, "mm_rotary_qk_inplace: negative offset");
    MM_CHECK_ARG(aligned16(qkv) && aligned16(cos) && aligned16(sin), "mm_rotary_qk_inplace: pointers must be 16-byte aligned");
    const size_t items = (size_t)s * b * np * 2 * (d / 16);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        rotary_qk_k<T><<<grid_for(items), 256, 0, (hipStream_t)stream>>>((u16*)qkv, (const u16*)cos, (const u16*)sin, s * b, b, np,
                                                                       hn, d, offset, inverse);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

extern "C" int mm_bias_gelu_fwd(void* out, const void* y, const void* bias, int rows, int cols, int mode, int dtype, void* stream) {
    MM_CHECK_ARG(out && y && bias, "mm_bias_gelu_fwd: null pointer");
    MM_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0, "mm_bias_gelu_fwd: rows=%d cols=%d (cols must be a multiple of 8)", rows, cols);
    MM_CHECK_ARG(mode == 0 || mode == 1, "mm_bias_gelu_fwd: mode must be 0 (tanh) or 1 (erf)");
    MM_CHECK_ARG(aligned16(out) && aligned16(y) && aligned16(bias), "mm_bias_gelu_fwd: pointers must be 16-byte aligned");
    const size_t items = (size_t)rows * (cols / 8);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        if (mode == 0)
            bias_gelu_k<T, 0, false><<<grid_for(items), 256, 0, (hipStream_t)stream>>>((u16*)out, nullptr, (const u16*)y, (const u16*)bias, rows, cols);
        else
            bias_gelu_k<T, 1, false><<<grid_for(items), 256, 0, (hipStream_t)stream>>>((u16*)out, nullptr, (const u16*)y, (const u16*)bias, rows, cols);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

extern "C" int mm_bias_gelu_bwd(void* dy, const void* g, const void* y, const void* bias, int rows, int cols, int mode,
                                int dtype, void* stream) {
    MM_CHECK_ARG(dy && g && y && bias, "mm_bias_gelu_bwd: null pointer");
    MM_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0, "mm_bias_gelu_bwd: rows=%d cols=%d (cols must be a multiple of 8)", rows, cols);
    MM_CHECK_ARG(mode == 0 || mode == 1, "mm_bias_gelu_bwd: mode must be 0 (tanh) or 1 (erf)");
    MM_CHECK_ARG(aligned16(dy) && aligned16(g) && aligned16(y) && aligned16(bias), "mm_bias_gelu_bwd: pointers must be 16-byte aligned");
    const size_t items = (size_t)rows * (cols / 8);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        if (mode == 0)
            bias_gelu_k<T, 0, true><<<grid_for(items), 256, 0, (hipStream_t)stream>>>((u16*)dy, (const u16*)g, (const u16*)y, (const u16*)bias, rows, cols);
        else
            bias_gelu_k<T, 1, true><<<grid_for(items), 256, 0, (hipStream_t)stream>>>((u16*)dy, (const u16*)g, (const u16*)y, (const u16*)bias, rows, cols);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

// state[1] += n, one lane (vector store)
__global__ void rng_advance_k(uint64_t* __restrict__ state, uint64_t n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[1] += n;
}

static int bias_dropout_add_impl(void* out, const void* x, const void* bias, const void* residual, uint8_t* keep_mask,
                                 int rows, int cols, float p, int training, uint64_t seed, uint64_t offset,
                                 const uint64_t* rng_state, int dtype, void* stream);

extern "C" int mm_bias_dropout_add(void* out, const void* x, const void* bias, const void* residual, uint8_t* keep_mask,
                                   int rows, int cols, float p, int training, uint64_t seed, uint64_t offset, int dtype,
                                   void* stream) {
    return bias_dropout_add_impl(out, x, bias, residual, keep_mask, rows, cols, p, training, seed, offset, nullptr, dtype, stream);
}

extern "C" int mm_bias_dropout_add_devrng(void* out, const void* x, const void* bias, const void* residual, uint8_t* keep_mask,
                                          int rows, int cols, float p, int training, const uint64_t* rng_state,
                                          uint64_t offset, int dtype, void* stream) {
    MM_CHECK_ARG(rng_state && (((uintptr_t)rng_state) & 7) == 0, "mm_bias_dropout_add_devrng: rng_state must be an 8-byte aligned device uint64[2]");
    return bias_dropout_add_impl(out, x, bias, residual, keep_mask, rows, cols, p, training, 0, offset, rng_state, dtype, stream);
}

extern "C" int mm_rng_advance(uint64_t* rng_state, uint64_t n, void* stream) {
    MM_CHECK_ARG(rng_state && (((uintptr_t)rng_state) & 7) == 0, "mm_rng_advance: rng_state must be an 8-byte aligned device uint64[2]");
    rng_advance_k<<<1, 64, 0, (hipStream_t)stream>>>(rng_state, n);
    MM_CHECK_LAUNCH();
    return MM_OK;
}

static int bias_dropout_add_impl(void* out, const void* x, const void* bias, const void* residual, uint8_t* keep_mask,
                                 int rows, int cols, float p, int training, uint64_t seed, uint64_t offset,
                                 const uint64_t* rng_state, int dtype, void* stream) {
    MM_CHECK_ARG(out && x, "mm_bias_dropout_add: null pointer");
    MM_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0, "mm_bias_dropout_add: rows=%d cols=%d (cols must be a multiple of 8)", rows, cols);
    MM_CHECK_ARG(p >= 0.f && p < 1.f, "mm_bias_dropout_add: dropout probability %f out of [0,1)", (double)p);
    MM_CHECK_ARG(aligned16(out) && aligned16(x) && (!bias || aligned16(bias)) && (!residual || aligned16(residual)),
                 "mm_bias_dropout_add: pointers must be 16-byte aligned");
    const size_t items = (size_t)rows * (cols / 8);
    const bool drop = training && p > 0.f;
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        if (drop)
            bias_dropout_add_k<T, true><<<grid_for(items), 256, 0, (hipStream_t)stream>>>((u16*)out, (const u16*)x, (const u16*)bias,
                                                                                        (const u16*)residual, keep_mask, rows, cols, p, seed, offset, rng_state);
        else
            bias_dropout_add_k<T, false><<<grid_for(items), 256, 0, (hipStream_t)stream>>>((u16*)out, (const u16*)x, (const u16*)bias,
                                                                                         (const u16*)residual, nullptr, rows, cols, p, seed, offset, nullptr);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

extern "C" int mm_colsum_nblk(int rows) {
    int n = (rows + 31) / 32;
    return n < 1 ? 1 : (n > 256 ? 256 : n);
}

extern "C" int mm_colsum(void* out, float* partial, const void* x, int rows, int cols, int dtype, void* stream) {
    MM_CHECK_ARG(out && partial && x, "mm_colsum: null pointer");
    MM_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0, "mm_colsum: rows=%d cols=%d (cols must be a multiple of 8)", rows, cols);
    MM_CHECK_ARG(aligned16(x) && aligned16(partial), "mm_colsum: pointers must be 16-byte aligned");
    const int nblk = mm_colsum_nblk(rows);
    hipStream_t st = (hipStream_t)stream;
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        colsum_partial_k<T><<<dim3((cols + 511) / 512, nblk), 64, 0, st>>>(partial, (const u16*)x, rows, cols, nblk);
        MM_CHECK_LAUNCH();
        colsum_final_k<T><<<(cols + 63) / 64, 256, 0, st>>>((u16*)out, partial, nblk, cols);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

extern "C" int mm_transpose2d(void* out, const void* in, int rows, int cols, void* stream) {
    MM_CHECK_ARG(out && in && out != in, "mm_transpose2d: null or aliased pointer");
    MM_CHECK_ARG(rows > 0 && cols > 0, "mm_transpose2d: sizes must be positive");
    MM_CHECK_ARG((cols + 63) / 64 <= 2147483647 && (rows + 63) / 64 <= 65535, "mm_transpose2d: %d x %d exceeds the grid limit", rows, cols);
    const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    if (rows % 8 == 0 && cols % 8 == 0 && aligned16(out) && aligned16(in))
        transpose2d_vec_k<<<grid, 256, 0, (hipStream_t)stream>>>((u16*)out, (const u16*)in, rows, cols);
    else
        transpose2d_k<<<grid, 256, 0, (hipStream_t)stream>>>((u16*)out, (const u16*)in, rows, cols);
    MM_CHECK_LAUNCH();
    return MM_OK;
}

extern "C" int mm_add(void* out, const void* a, const void* b, const void* c, size_t n, int dtype, void* stream) {
    MM_CHECK_ARG(out && a && b, "mm_add: null pointer");
    MM_CHECK_ARG(n > 0 && n % 8 == 0, "mm_add: n must be a positive multiple of 8");
    MM_CHECK_ARG(aligned16(out) && aligned16(a) && aligned16(b) && (!c || aligned16(c)), "mm_add: pointers must be 16-byte aligned");
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        add_k<T><<<grid_for(n / 8), 256, 0, (hipStream_t)stream>>>((u16*)out, (const u16*)a, (const u16*)b, (const u16*)c, n / 8);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

// ------------------------------------------------------------------------------------ relu backward / gpt-j combine
namespace mm {
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_k(u16* __restrict__ dx, const u16* __restrict__ g, const u16* __restrict__ act, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 gr = *reinterpret_cast<const u32x4*>(g + i * 8);
        const u32x4 ar = *reinterpret_cast<const u32x4*>(act + i * 8);
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // act is a ReLU output (>= 0): "positive" <=> non-zero magnitude bits and sign clear
            const uint32_t lo = ((ar[j] & 0x7fffu) != 0 && (ar[j] & 0x8000u) == 0) ? (gr[j] & 0xffffu) : 0u;
            const uint32_t hi = ((ar[j] & 0x7fff0000u) != 0 && (ar[j] & 0x80000000u) == 0) ? (gr[j] & 0xffff0000u) : 0u;
            o[j] = lo | hi;
        }
        *reinterpret_cast<u32x4*>(dx + i * 8) = o;
    }
}

// GPT-J residual tail at TP = 1 (reference transformer.py:747-766), one pass:
//   t1 = attn + b_a ; t2 = mlp + b_m ; t3 = t1 + t2 ; out = x + t3, each rounded to T like the eager ops.
template <typename T>
__global__ __launch_bounds__(256) void gptj_combine_k(u16* __restrict__ out, const u16* __restrict__ x, const u16* __restrict__ attn,
                                                      const u16* __restrict__ ba, const u16* __restrict__ mlp,
                                                      const u16* __restrict__ bm, size_t rows, int cols, int add_x) {
    const int cpr = cols >> 3;
    const size_t total = rows * cpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr);
        float xv[8], av[8], mv[8], b1[8], b2[8], o[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(attn + i * 8), av);
        unpack8<T>(*reinterpret_cast<const u32x4*>(mlp + i * 8), mv);
        unpack8<T>(*reinterpret_cast<const u32x4*>(ba + c * 8), b1);
        unpack8<T>(*reinterpret_cast<const u32x4*>(bm + c * 8), b2);
        if (add_x) unpack8<T>(*reinterpret_cast<const u32x4*>(x + i * 8), xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t3 = rnd<T>(rnd<T>(av[j] + b1[j]) + rnd<T>(mv[j] + b2[j]));
            o[j] = add_x ? xv[j] + t3 : t3;
        }
        *reinterpret_cast<u32x4*>(out + i * 8) = pack8<T>(o);
    }
}
}  // namespace mm

extern "C" int mm_relu_bwd(void* dx, const void* g, const void* act, size_t n, int dtype, void* stream) {
    MM_CHECK_ARG(dx && g && act, "mm_relu_bwd: null pointer");
    MM_CHECK_ARG(n > 0 && n % 8 == 0, "mm_relu_bwd: n must be a positive multiple of 8");
    MM_CHECK_ARG(aligned16(dx) && aligned16(g) && aligned16(act), "mm_relu_bwd: pointers must be 16-byte aligned");
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        relu_bwd_k<T><<<grid_for(n / 8), 256, 0, (hipStream_t)stream>>>((u16*)dx, (const u16*)g, (const u16*)act, n / 8);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

extern "C" int mm_gptj_combine(void* out, const void* x, const void* attn, const void* attn_bias, const void* mlp,
                               const void* mlp_bias, int rows, int cols, int dtype, void* stream) {
    MM_CHECK_ARG(out && attn && attn_bias && mlp && mlp_bias, "mm_gptj_combine: null pointer");
    MM_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0, "mm_gptj_combine: rows=%d cols=%d (cols must be a multiple of 8)", rows, cols);
    MM_CHECK_ARG(aligned16(out) && aligned16(attn) && aligned16(mlp) && aligned16(attn_bias) && aligned16(mlp_bias) && (!x || aligned16(x)),
                 "mm_gptj_combine: pointers must be 16-byte aligned");
    const size_t items = (size_t)rows * (cols / 8);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        gptj_combine_k<T><<<grid_for(items), 256, 0, (hipStream_t)stream>>>((u16*)out, (const u16*)x, (const u16*)attn, (const u16*)attn_bias,
                                                                           (const u16*)mlp, (const u16*)mlp_bias, rows, cols, x ? 1 : 0);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

// ------------------------------------------------------------------------------------ embedding + image prefix
// out[si, bi, :] = si < P ? prefix[bi, si, :] : table[tokens[bi, si - P], :]
// = word embedding gather + `cat((image_prefix, text[:, :s-P]), dim=1)` + the [b, s, h] -> [s, b, h] transpose in front of
// the first layer (megatron/model/word_embeddings.py:135-205, gpt2_model.py:73-77) in one pass: 16 bytes per lane, each
// output row written once.  Token ids are clamped to [0, V) (the reference would fault on an id outside the table).
namespace mm {
template <int DUMMY = 0>
__global__ __launch_bounds__(256) void embed_prefix_k(u16* __restrict__ out, const u16* __restrict__ table,
                                                      const long long* __restrict__ tokens, const u16* __restrict__ prefix, int s,
                                                      int b, int h, int P, int V) {
    const int cpr = h >> 3;  // 16-byte chunks per row
    const size_t total = (size_t)s * b * cpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr);
        const size_t row = i / cpr;  // = si * b + bi
        const int si = (int)(row / b), bi = (int)(row % b);
        const u16* src;
        if (si < P) {
            src = prefix + ((size_t)bi * P + si) * h;
        } else {
            long long t = tokens[(size_t)bi * s + (si - P)];
            t = t < 0 ? 0 : (t >= V ? V - 1 : t);
            src = table + (size_t)t * h;
        }
        *reinterpret_cast<u32x4*>(out + row * h + c * 8) = *reinterpret_cast<const u32x4*>(src + c * 8);
    }
}
}  // namespace mm

extern "C" int mm_embed_prefix(void* out, const void* table, const int64_t* tokens, const void* prefix, int s, int b, int h, int P,
                               int V, void* stream) {
    MM_CHECK_ARG(out && table && tokens, "mm_embed_prefix: null pointer");
    MM_CHECK_ARG(s > 0 && b > 0 && h > 0 && V > 0 && h % 8 == 0, "mm_embed_prefix: s=%d b=%d h=%d V=%d (h must be a multiple of 8)", s, b, h, V);
    MM_CHECK_ARG(P >= 0 && P <= s && (P == 0 || prefix != nullptr), "mm_embed_prefix: prefix length %d out of range (s=%d) or prefix missing", P, s);
    MM_CHECK_ARG(aligned16(out) && aligned16(table) && (!prefix || aligned16(prefix)), "mm_embed_prefix: pointers must be 16-byte aligned");
    const size_t items = (size_t)s * b * (h / 8);
    mm::embed_prefix_k<><<<mm::grid_for(items), 256, 0, (hipStream_t)stream>>>((u16*)out, (const u16*)table, (const long long*)tokens,
                                                                              (const u16*)prefix, s, b, h, P, V);
    MM_CHECK_LAUNCH();
    return MM_OK;
}
