// norm.hip - LayerNorm forward / backward for [rows, h] activations (HBM-bound).
//
// MI355X design: one wave64 per row, the row lives in registers (16-byte loads, <= ITERS*8 values
// per lane), two-pass mean / variance in fp32 with wave butterflies, 4 rows per 256-thread block.
// The GPT-J residual variant (reference transformer.py:737) normalises the SAME x with two affine
// pairs: the kernel shares the statistics and emits both outputs from one read of x; the backward
// likewise folds both branches and the residual gradient into one pass.
// Parameter gradients (not needed when the LM is frozen) are a separate column-reduction kernel.
#include "common.h"

namespace mm {

// Transposed-input variants (final LayerNorm of the stack: the layers' [s, b, h] activation is read by stride and the output,
// statistics and incoming gradient are in [b, s, h] order - _post_transformer_block's transpose(0, 1).contiguous(),
// megatron/model/gpt2_model.py:80-85, without the copy): output row r = bi * s + si reads input row si * b + bi.
__device__ __forceinline__ int in_row_of(int row, int tr_s, int tr_b) { return tr_b ? (row % tr_s) * tr_b + row / tr_s : row; }

template <typename T, int ITERS, bool DUAL>
__global__ __launch_bounds__(256) void layernorm_fwd_k(u16* __restrict__ y, u16* __restrict__ y2, float* __restrict__ mean_o,
                                                       float* __restrict__ rstd_o, const u16* __restrict__ x,
                                                       const u16* __restrict__ w, const u16* __restrict__ b,
                                                       const u16* __restrict__ w2, const u16* __restrict__ b2, int rows,
                                                       int h, float eps, int tr_s = 0, int tr_b = 0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = h >> 3;
    const u16* xr = x + (size_t)in_row_of(row, tr_s, tr_b) * h;
    float v[ITERS][8];
    float sum = 0.f;
    // the affine parameters are requested together with the row (their latency hides behind the two reductions)
    u32x4 wr[ITERS], br[ITERS], wr2[ITERS], br2[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = it * 64 + lane;
        if (c < nchunk) {
            wr[it] = *reinterpret_cast<const u32x4*>(w + c * 8);
            br[it] = *reinterpret_cast<const u32x4*>(b + c * 8);
            if (DUAL) {
                wr2[it] = *reinterpret_cast<const u32x4*>(w2 + c * 8);
                br2[it] = *reinterpret_cast<const u32x4*>(b2 + c * 8);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = it * 64 + lane;
        if (c < nchunk) {
            u32x4 raw = *reinterpret_cast<const u32x4*>(xr + c * 8);
            unpack8<T>(raw, v[it]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[it][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[it][j] = 0.f;
        }
    }
    const float inv_h = 1.0f / (float)h;
    const float mean = wave_sum(sum) * inv_h;
    float sq = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = it * 64 + lane;
        if (c < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[it][j] - mean;
                sq += d * d;
            }
        }
    }
    const float var = wave_sum(sq) * inv_h;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
        if (mean_o) mean_o[row] = mean;
        if (rstd_o) rstd_o[row] = rstd;
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = it * 64 + lane;
        if (c < nchunk) {
            float wv[8], bv[8], o[8];
            unpack8<T>(wr[it], wv);
            unpack8<T>(br[it], bv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[it][j] - mean) * rstd * wv[j] + bv[j];
            *reinterpret_cast<u32x4*>(y + (size_t)row * h + c * 8) = pack8<T>(o);
            if (DUAL) {
                unpack8<T>(wr2[it], wv);
                unpack8<T>(br2[it], bv);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (v[it][j] - mean) * rstd * wv[j] + bv[j];
                *reinterpret_cast<u32x4*>(y2 + (size_t)row * h + c * 8) = pack8<T>(o);
            }
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ dres],  g = dy*w [+ dy2*w2]
template <typename T, int ITERS, bool DUAL, bool RES>
__global__ __launch_bounds__(256) void layernorm_bwd_dx_k(u16* __restrict__ dx, const u16* __restrict__ dy,
                                                          const u16* __restrict__ dy2, const u16* __restrict__ dres,
                                                          const u16* __restrict__ x, const u16* __restrict__ w,
                                                          const u16* __restrict__ w2, const float* __restrict__ mean_i,
                                                          const float* __restrict__ rstd_i, int rows, int h, int tr_s = 0,
                                                          int tr_b = 0, const float* __restrict__ dy_scale = nullptr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = h >> 3;
    const size_t ro = (size_t)row * h;                            // dy / dy2 / statistics: output order
    const size_t xo = (size_t)in_row_of(row, tr_s, tr_b) * h;     // x / dx / dres: input order
    const float mean = mean_i[row], rstd = rstd_i[row];
    const float gsc = dy_scale ? *dy_scale : 1.0f;  // upstream scalar gradient of a fused loss node (device scalar, no host sync)
    float g[ITERS][8], xh[ITERS][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = it * 64 + lane;
        if (c < nchunk) {
            float a[8], wv[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(x + xo + c * 8), xh[it]);
            unpack8<T>(*reinterpret_cast<const u32x4*>(dy + ro + c * 8), a);
            unpack8<T>(*reinterpret_cast<const u32x4*>(w + c * 8), wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[it][j] = (dy_scale ? a[j] * gsc : a[j]) * wv[j];
            if (DUAL) {
                unpack8<T>(*reinterpret_cast<const u32x4*>(dy2 + ro + c * 8), a);
                unpack8<T>(*reinterpret_cast<const u32x4*>(w2 + c * 8), wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) g[it][j] += a[j] * wv[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[it][j] = (xh[it][j] - mean) * rstd;
                s1 += g[it][j];
                s2 += g[it][j] * xh[it][j];
            }
        }
    }
    const float inv_h = 1.0f / (float)h;
    const float c1 = wave_sum(s1) * inv_h, c2 = wave_sum(s2) * inv_h;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = it * 64 + lane;
        if (c < nchunk) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rstd * (g[it][j] - c1 - xh[it][j] * c2);
            if (RES) {
                float r[8];
                unpack8<T>(*reinterpret_cast<const u32x4*>(dres + xo + c * 8), r);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += r[j];
            }
            *reinterpret_cast<u32x4*>(dx + xo + c * 8) = pack8<T>(o);
        }
    }
}

// partial[blk][k][col]: k=0 dw = sum dy*xhat, k=1 db = sum dy, (k=2 dw2, k=3 db2 from dy2).
// grid: (ceil(h/512), nblk); one wave per block, 8 columns per lane.
template <typename T, bool DUAL>
__global__ __launch_bounds__(64) void layernorm_bwd_param_k(float* __restrict__ partial, const u16* __restrict__ dy,
                                                            const u16* __restrict__ dy2, const u16* __restrict__ x,
                                                            const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                            int rows, int h, int nblk, int tr_s = 0, int tr_b = 0,
                                                            const float* __restrict__ dy_scale = nullptr) {
    const int col = (blockIdx.x * 64 + threadIdx.x) * 8;
    const int blk = blockIdx.y;
    constexpr int NV = DUAL ? 4 : 2;
    float acc[NV][8];
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    if (col < h) {
        const int per = (rows + nblk - 1) / nblk;
        const int r0 = blk * per, r1 = min(rows, r0 + per);
        const float gsc = dy_scale ? *dy_scale : 1.0f;
        for (int r = r0; r < r1; ++r) {
            const float mean = mean_i[r], rstd = rstd_i[r];
            float xv[8], a[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(x + (size_t)in_row_of(r, tr_s, tr_b) * h + col), xv);
            unpack8<T>(*reinterpret_cast<const u32x4*>(dy + (size_t)r * h + col), a);
            if (dy_scale) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] *= gsc;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xv[j] = (xv[j] - mean) * rstd;
                acc[0][j] += a[j] * xv[j];
                acc[1][j] += a[j];
            }
            if (DUAL) {
                unpack8<T>(*reinterpret_cast<const u32x4*>(dy2 + (size_t)r * h + col), a);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[2][j] += a[j] * xv[j];
                    acc[3][j] += a[j];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            float* p = partial + ((size_t)blk * NV + k) * h + col;
            *reinterpret_cast<f32x4*>(p) = f32x4{acc[k][0], acc[k][1], acc[k][2], acc[k][3]};
            *reinterpret_cast<f32x4*>(p + 4) = f32x4{acc[k][4], acc[k][5], acc[k][6], acc[k][7]};
        }
    }
}

// out_k[col] = sum_blk partial[blk][k][col]
template <typename T>
__global__ __launch_bounds__(256) void reduce_partials_k(u16* o0, u16* o1, u16* o2, u16* o3, const float* __restrict__ partial,
                                                         int nblk, int nvec, int h) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (col >= h) return;
    u16* o = k == 0 ? o0 : (k == 1 ? o1 : (k == 2 ? o2 : o3));
    if (!o) return;
    float s = 0.f;
    for (int bI = 0; bI < nblk; ++bI) s += partial[((size_t)bI * nvec + k) * h + col];
    o[col] = from_f32<T>(s);
}

static int pick_iters(int h) {
    const int need = ((h >> 3) + 63) / 64;
    const int opts[] = {1, 2, 3, 4, 5, 8, 16};
    for (int o : opts)
        if (o >= need) return o;
    return -1;
}

}  // namespace mm

using namespace mm;

extern "C" int mm_layernorm_bwd_nblk(int rows) {
    int n = (rows + 31) / 32;
    return n < 1 ? 1 : (n > 256 ? 256 : n);
}

#define MM_LN_ITERS_SWITCH(iters, ...) \
    switch (iters) {                    \
        case 1: { constexpr int IT = 1; __VA_ARGS__; } break;   \
        case 2: { constexpr int IT = 2; __VA_ARGS__; } break;   \
        case 3: { constexpr int IT = 3; __VA_ARGS__; } break;   \
        case 4: { constexpr int IT = 4; __VA_ARGS__; } break;   \
        case 5: { constexpr int IT = 5; __VA_ARGS__; } break;   \
        case 8: { constexpr int IT = 8; __VA_ARGS__; } break;   \
        default: { constexpr int IT = 16; __VA_ARGS__; } break; \
    }

extern "C" int mm_layernorm_fwd(void* y, void* y2, float* mean, float* rstd, const void* x, const void* w, const void* b,
                                const void* w2, const void* b2, int rows, int h, float eps, int dtype, void* stream) {
    MM_CHECK_ARG(y && x && w && b, "mm_layernorm_fwd: null pointer");
    MM_CHECK_ARG(rows > 0 && h > 0, "mm_layernorm_fwd: rows=%d h=%d must be positive", rows, h);
    MM_CHECK_ARG(h % 8 == 0 && h <= 8192, "mm_layernorm_fwd: h=%d must be a multiple of 8 and <= 8192", h);
    MM_CHECK_ARG(aligned16(y) && aligned16(x) && aligned16(w) && aligned16(b), "mm_layernorm_fwd: pointers must be 16-byte aligned");
    const bool dual = y2 != nullptr;
    if (dual) MM_CHECK_ARG(w2 && b2 && aligned16(y2) && aligned16(w2) && aligned16(b2), "mm_layernorm_fwd: dual output needs aligned w2/b2");
    const int iters = pick_iters(h);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((rows + 3) / 4), block(256);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        MM_LN_ITERS_SWITCH(iters, {
            if (dual)
                layernorm_fwd_k<T, IT, true><<<grid, block, 0, st>>>((u16*)y, (u16*)y2, mean, rstd, (const u16*)x, (const u16*)w,
                                                                     (const u16*)b, (const u16*)w2, (const u16*)b2, rows, h, eps);
            else
                layernorm_fwd_k<T, IT, false><<<grid, block, 0, st>>>((u16*)y, nullptr, mean, rstd, (const u16*)x, (const u16*)w,
                                                                      (const u16*)b, nullptr, nullptr, rows, h, eps);
        });
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

extern "C" int mm_layernorm_bwd(void* dx, float* dwdb_partial, const void* dy, const void* dy2, const void* dres,
                                const void* x, const void* w, const void* w2, const float* mean, const float* rstd, int rows,
                                int h, int dtype, void* stream) {
    MM_CHECK_ARG(dx && dy && x && w && mean && rstd, "mm_layernorm_bwd: null pointer");
    MM_CHECK_ARG(rows > 0 && h > 0, "mm_layernorm_bwd: rows=%d h=%d must be positive", rows, h);
    MM_CHECK_ARG(h % 8 == 0 && h <= 8192, "mm_layernorm_bwd: h=%d must be a multiple of 8 and <= 8192", h);
    MM_CHECK_ARG(aligned16(dx) && aligned16(dy) && aligned16(x) && aligned16(w), "mm_layernorm_bwd: pointers must be 16-byte aligned");
    const bool dual = dy2 != nullptr;
    if (dual) MM_CHECK_ARG(w2 && aligned16(dy2) && aligned16(w2), "mm_layernorm_bwd: dual branch needs aligned dy2/w2");
    if (dres) MM_CHECK_ARG(aligned16(dres), "mm_layernorm_bwd: dresidual must be 16-byte aligned");
    const int iters = pick_iters(h);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((rows + 3) / 4), block(256);
    const int nblk = mm_layernorm_bwd_nblk(rows);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
#define MM_LN_BWD(DUALV, RESV)                                                                                             \
    layernorm_bwd_dx_k<T, IT, DUALV, RESV><<<grid, block, 0, st>>>((u16*)dx, (const u16*)dy, (const u16*)dy2, (const u16*)dres, \
                                                                  (const u16*)x, (const u16*)w, (const u16*)w2, mean, rstd, rows, h)
        MM_LN_ITERS_SWITCH(iters, {
            if (dual && dres) MM_LN_BWD(true, true);
            else if (dual) MM_LN_BWD(true, false);
            else if (dres) MM_LN_BWD(false, true);
            else MM_LN_BWD(false, false);
        });
#undef MM_LN_BWD
        MM_CHECK_LAUNCH();
        if (dwdb_partial) {
            dim3 pg((h + 511) / 512, nblk), pb(64);
            if (dual)
                layernorm_bwd_param_k<T, true><<<pg, pb, 0, st>>>(dwdb_partial, (const u16*)dy, (const u16*)dy2, (const u16*)x, mean, rstd, rows, h, nblk);
            else
                layernorm_bwd_param_k<T, false><<<pg, pb, 0, st>>>(dwdb_partial, (const u16*)dy, nullptr, (const u16*)x, mean, rstd, rows, h, nblk);
            MM_CHECK_LAUNCH();
        }
        return (int)MM_OK;
    });
}

extern "C" int mm_layernorm_bwd_finalize(void* const* outs, const float* partial, int nblk, int nvec, int h, int dtype,
                                         void* stream) {
    MM_CHECK_ARG(outs && partial, "mm_layernorm_bwd_finalize: null pointer");
    MM_CHECK_ARG(nvec >= 1 && nvec <= 4 && nblk >= 1 && h > 0, "mm_layernorm_bwd_finalize: bad sizes nblk=%d nvec=%d h=%d", nblk, nvec, h);
    hipStream_t st = (hipStream_t)stream;
    void* o[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < nvec; ++k) o[k] = outs[k];
    dim3 grid((h + 255) / 256, nvec), block(256);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        reduce_partials_k<T><<<grid, block, 0, st>>>((u16*)o[0], (u16*)o[1], (u16*)o[2], (u16*)o[3], partial, nblk, nvec, h);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

// ---- the stack's final LayerNorm without _post_transformer_block's copy (megatron/model/gpt2_model.py:80-85,
// megatron/model/transformer.py:828-839): x is the layers' [s, b, h] activation, y / mean / rstd / dy are in [b, s, h] order.
extern "C" int mm_layernorm_fwd_sb(void* y, float* mean, float* rstd, const void* x, const void* w, const void* b, int s, int bsz,
                                   int h, float eps, int dtype, void* stream) {
    MM_CHECK_ARG(y && x && w && b, "mm_layernorm_fwd_sb: null pointer");
    MM_CHECK_ARG(s > 0 && bsz > 0 && h > 0 && (long)s * bsz < (1l << 31), "mm_layernorm_fwd_sb: s=%d b=%d h=%d", s, bsz, h);
    MM_CHECK_ARG(h % 8 == 0 && h <= 8192, "mm_layernorm_fwd_sb: h=%d must be a multiple of 8 and <= 8192", h);
    MM_CHECK_ARG(aligned16(y) && aligned16(x) && aligned16(w) && aligned16(b), "mm_layernorm_fwd_sb: pointers must be 16-byte aligned");
    const int rows = s * bsz, iters = pick_iters(h);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((rows + 3) / 4), block(256);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        MM_LN_ITERS_SWITCH(iters, {
            layernorm_fwd_k<T, IT, false><<<grid, block, 0, st>>>((u16*)y, nullptr, mean, rstd, (const u16*)x, (const u16*)w,
                                                                  (const u16*)b, nullptr, nullptr, rows, h, eps, s, bsz);
        });
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

// dx [s, b, h] from dy [b, s, h] (times the device scalar *dy_scale when given); dwdb_partial as mm_layernorm_bwd.
extern "C" int mm_layernorm_bwd_sb(void* dx, float* dwdb_partial, const void* dy, const float* dy_scale, const void* x,
                                   const void* w, const float* mean, const float* rstd, int s, int bsz, int h, int dtype,
                                   void* stream) {
    MM_CHECK_ARG(dx && dy && x && w && mean && rstd, "mm_layernorm_bwd_sb: null pointer");
    MM_CHECK_ARG(s > 0 && bsz > 0 && h > 0 && (long)s * bsz < (1l << 31), "mm_layernorm_bwd_sb: s=%d b=%d h=%d", s, bsz, h);
    MM_CHECK_ARG(h % 8 == 0 && h <= 8192, "mm_layernorm_bwd_sb: h=%d must be a multiple of 8 and <= 8192", h);
    MM_CHECK_ARG(aligned16(dx) && aligned16(dy) && aligned16(x) && aligned16(w), "mm_layernorm_bwd_sb: pointers must be 16-byte aligned");
    const int rows = s * bsz, iters = pick_iters(h);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((rows + 3) / 4), block(256);
    const int nblk = mm_layernorm_bwd_nblk(rows);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        MM_LN_ITERS_SWITCH(iters, {
            layernorm_bwd_dx_k<T, IT, false, false><<<grid, block, 0, st>>>((u16*)dx, (const u16*)dy, nullptr, nullptr, (const u16*)x,
                                                                            (const u16*)w, nullptr, mean, rstd, rows, h, s, bsz, dy_scale);
        });
        MM_CHECK_LAUNCH();
        if (dwdb_partial) {
            dim3 pg((h + 511) / 512, nblk), pb(64);
            layernorm_bwd_param_k<T, false><<<pg, pb, 0, st>>>(dwdb_partial, (const u16*)dy, nullptr, (const u16*)x, mean, rstd, rows, h,
                                                               nblk, s, bsz, dy_scale);
            MM_CHECK_LAUNCH();
        }
        return (int)MM_OK;
    });
}

