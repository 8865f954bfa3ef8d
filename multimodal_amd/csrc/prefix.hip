// prefix.hip - the ImagePrefix tail behind the (out-of-scope) image encoder, reference
// megatron/model/image_prefix.py:232-252: proj (nn.Linear enc -> h) -> dropout -> LayerNorm -> [b, P, h], as ONE kernel
// for the few rows a step has (P = 1 prefix token per sample with the pooled openclip-H encoder, image_prefix.py:106: b rows).
//
// MI355X design: with 4 rows this is a GEMV over the 2 MiB projection weight followed by a per-row normalisation over all
// N outputs, i.e. a grid-wide dependency after a bandwidth-bound pass.  One launch does both: N / 32 workgroups each read 32
// weight rows once (16-byte loads, 8 lanes per output column, the <= 16 input rows staged in LDS), leave their [rows, 32]
// slice of the rounded projection in an fp32 workspace, and the workgroup whose ticket is last (agent-scope release on every
// producer, agent-scope acquire on the last arriver: the L2s of the eight XCDs are not coherent) applies the dropout and the
// LayerNorm to the whole rows.  Roundings follow the three-kernel chain it replaces (GEMM epilogue rounds x W^T + b once;
// bias_dropout_add_k's Philox stream and roundings; layernorm_fwd_k's two-pass statistics), so the result equals that chain's
// up to the summation order of the dot products.
#include "common.h"
#include "philox.h"

namespace mm {

constexpr int kPrefixMaxRows = 16;
constexpr int kPrefixColsPerWg = 32;

template <typename T>
__global__ __launch_bounds__(256) void prefix_proj_ln_k(u16* __restrict__ out, u16* __restrict__ y_ln_in, uint8_t* __restrict__ keep,
                                                        float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                        const u16* __restrict__ x, const u16* __restrict__ w,
                                                        const u16* __restrict__ bias, const u16* __restrict__ ln_w,
                                                        const u16* __restrict__ ln_b, float* ws_y, unsigned* counter, int R, int K,
                                                        int N, float eps, float p, int drop, uint64_t offset,
                                                        const uint64_t* __restrict__ rng_state) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [R][K] input rows (16-bit), then one int: "I am last"
    const int tid = threadIdx.x;
    {
        const int n16 = R * K / 8;
        for (int i = tid; i < n16; i += 256) reinterpret_cast<u32x4*>(smem)[i] = reinterpret_cast<const u32x4*>(x)[i];
    }
    int* last_flag = reinterpret_cast<int*>(smem + (size_t)R * K * 2);
    __syncthreads();
    const int cl = tid >> 3, sub = tid & 7;
    const int c = blockIdx.x * kPrefixColsPerWg + cl;
    float acc[kPrefixMaxRows];
#pragma unroll
    for (int r = 0; r < kPrefixMaxRows; ++r) acc[r] = 0.f;
    const u16* wr = w + (size_t)c * K;
    const int nchunkK = K >> 3;
    for (int j = sub; j < nchunkK; j += 8) {
        float wv[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(wr + j * 8), wv);
#pragma unroll
        for (int r = 0; r < kPrefixMaxRows; ++r) {
            if (r < R) {  // uniform over the grid
                float xv[8];
                unpack8<T>(*reinterpret_cast<const u32x4*>(smem + ((size_t)r * K + j * 8) * 2), xv);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[r] = __builtin_fmaf(xv[e], wv[e], acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kPrefixMaxRows; ++r) {
        acc[r] += __shfl_xor(acc[r], 1, 64);
        acc[r] += __shfl_xor(acc[r], 2, 64);
        acc[r] += __shfl_xor(acc[r], 4, 64);
    }
    // lane `sub` of a column's 8 lanes writes rows sub and sub + 8 (static register indices: no scratch)
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (sub == r) {
            v0 = acc[r];
            v1 = acc[r + 8];
        }
    const float bv = bias ? to_f32<T>(bias[c]) : 0.f;
    if (sub < R) ws_y[(size_t)sub * N + c] = rnd<T>(v0 + bv);  // the GEMM epilogue's single rounding of x W^T + b
    if (sub + 8 < R) ws_y[(size_t)(sub + 8) * N + c] = rnd<T>(v1 + bv);

    // ---- hand-off: every storing wave drains, the workgroup meets, ONE lane releases at agent scope and draws a ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == gridDim.x - 1) ? 1 : 0;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *last_flag = last;
    }
    __syncthreads();
    if (!*last_flag) return;

    // ---- last arriver: dropout (bias_dropout_add_k's stream: element chunk i = (row * N + col) / 8) + LayerNorm per row
    uint64_t seed = 0;
    if (drop) {
        seed = rng_state[0];
        offset += rng_state[1];
    }
    const float inv_keep = drop ? 1.0f / (1.0f - p) : 1.0f;
    const int wave = tid >> 6, lane = tid & 63;
    const int nchunk = N >> 3;
    constexpr int ITERS = 8;  // N <= 4096
    for (int row = wave; row < R; row += 4) {
        float v[ITERS][8];
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int ch = it * 64 + lane;
            if (ch < nchunk) {
                const float* src = ws_y + (size_t)row * N + ch * 8;
                const f32x4 a = *reinterpret_cast<const f32x4*>(src), b4 = *reinterpret_cast<const f32x4*>(src + 4);
                float y[8] = {a[0], a[1], a[2], a[3], b4[0], b4[1], b4[2], b4[3]};
                if (drop) {
                    const uint64_t i = (uint64_t)row * nchunk + ch;
                    uint32_t r0[4], r1[4];
                    Philox::gen(seed, i, offset, r0);
                    Philox::gen(seed, i, offset + 0x8000000000000000ull, r1);
                    uint32_t klo = 0, khi = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t bits = j < 4 ? r0[j] : r1[j - 4];
                        const bool k = (float)(bits >> 8) * (1.0f / 16777216.0f) >= p;
                        y[j] = k ? rnd<T>(y[j] * inv_keep) : 0.f;
                        if (j < 4) klo |= (k ? 1u : 0u) << (8 * j);
                        else khi |= (k ? 1u : 0u) << (8 * (j - 4));
                    }
                    if (keep) *reinterpret_cast<u32x2*>(keep + i * 8) = u32x2{klo, khi};
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[it][j] = y[j];
                    sum += y[j];
                }
                if (y_ln_in) *reinterpret_cast<u32x4*>(y_ln_in + (size_t)row * N + ch * 8) = pack8<T>(y);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[it][j] = 0.f;
            }
        }
        const float inv_n = 1.0f / (float)N;
        const float mean = wave_sum(sum) * inv_n;
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int ch = it * 64 + lane;
            if (ch < nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = v[it][j] - mean;
                    sq += d * d;
                }
            }
        }
        const float var = wave_sum(sq) * inv_n;
        const float rstd = 1.0f / sqrtf(var + eps);
        if (lane == 0) {
            if (mean_o) mean_o[row] = mean;
            if (rstd_o) rstd_o[row] = rstd;
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int ch = it * 64 + lane;
            if (ch < nchunk) {
                float wv[8], bb[8], o[8];
                unpack8<T>(*reinterpret_cast<const u32x4*>(ln_w + ch * 8), wv);
                unpack8<T>(*reinterpret_cast<const u32x4*>(ln_b + ch * 8), bb);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (v[it][j] - mean) * rstd * wv[j] + bb[j];
                *reinterpret_cast<u32x4*>(out + (size_t)row * N + ch * 8) = pack8<T>(o);
            }
        }
    }
}

// Zeroes the ticket words in front of a launch.  A kernel, not hipMemsetAsync: under the framework's stream capture on this
// ROCm the memset ran at capture time instead of becoming a graph node, so a replayed step found the ticket already spent and
// no workgroup normalised the rows (the replay returned the previous batch's prefix: tools/dev/dbg_prefix_graph.py).
__global__ void zero_words_k(unsigned* __restrict__ p) { p[threadIdx.x] = 0u; }

// dx = keep ? rnd(g / (1 - p)) : 0 - backward of the dropout above (and of bias_dropout_add's), 8 elements per lane
template <typename T>
__global__ __launch_bounds__(256) void dropout_bwd_k(u16* __restrict__ dx, const u16* __restrict__ g, const uint8_t* __restrict__ keep,
                                                     size_t n8, float inv_keep) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        float gv[8], o[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(g + i * 8), gv);
        const u32x2 k = *reinterpret_cast<const u32x2*>(keep + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t kb = ((j < 4 ? k[0] : k[1]) >> (8 * (j & 3))) & 0xffu;
            o[j] = kb ? gv[j] * inv_keep : 0.f;
        }
        *reinterpret_cast<u32x4*>(dx + i * 8) = pack8<T>(o);
    }
}

}  // namespace mm

using namespace mm;

extern "C" int mm_prefix_tail_supported(int rows, int enc_dim, int out_dim) {
    return rows >= 1 && rows <= kPrefixMaxRows && enc_dim % 64 == 0 && enc_dim > 0 && out_dim % kPrefixColsPerWg == 0 &&
           out_dim % 8 == 0 && out_dim > 0 && out_dim <= 4096 && (long)rows * enc_dim <= 32768;
}

extern "C" size_t mm_prefix_tail_workspace_bytes(int rows, int out_dim) { return 256 + (size_t)rows * out_dim * sizeof(float); }

extern "C" int mm_prefix_tail_fwd(void* out, void* ln_in, uint8_t* keep_mask, float* mean, float* rstd, const void* x, const void* w,
                                  const void* bias, const void* ln_w, const void* ln_b, void* workspace, int rows, int enc_dim,
                                  int out_dim, float eps, float p, int training, const uint64_t* rng_state, uint64_t offset,
                                  int dtype, void* stream) {
    MM_CHECK_ARG(out && x && w && ln_w && ln_b && workspace, "mm_prefix_tail_fwd: null pointer");
    MM_CHECK_ARG(mm_prefix_tail_supported(rows, enc_dim, out_dim),
                 "mm_prefix_tail_fwd: rows=%d enc_dim=%d out_dim=%d outside the fused kernel's range (rows <= %d, enc_dim %% 64 == 0, "
                 "out_dim %% 32 == 0 and <= 4096, rows * enc_dim <= 32768): use mm_gemm_nt + mm_bias_dropout_add + mm_layernorm_fwd",
                 rows, enc_dim, out_dim, kPrefixMaxRows);
    MM_CHECK_ARG(aligned16(out) && aligned16(x) && aligned16(w) && aligned16(ln_w) && aligned16(ln_b) && aligned16(workspace) &&
                     (!ln_in || aligned16(ln_in)) && (!keep_mask || (((uintptr_t)keep_mask) & 7) == 0),
                 "mm_prefix_tail_fwd: pointers must be 16-byte aligned");
    MM_CHECK_ARG(p >= 0.f && p < 1.f, "mm_prefix_tail_fwd: dropout probability %g outside [0, 1)", (double)p);
    const int drop = (training && p > 0.f) ? 1 : 0;
    if (drop) MM_CHECK_ARG(rng_state != nullptr, "mm_prefix_tail_fwd: training with p > 0 needs rng_state (device uint64[2] {seed, offset})");
    hipStream_t st = (hipStream_t)stream;
    // the ticket counter lives in the first 256 bytes of the workspace and is zeroed on the stream before every launch
    unsigned* counter = (unsigned*)workspace;
    zero_words_k<<<1, 64, 0, st>>>(counter);
    MM_CHECK_LAUNCH();
    float* ws_y = (float*)((char*)workspace + 256);
    const size_t lds = (size_t)rows * enc_dim * 2 + 16;
    dim3 grid(out_dim / kPrefixColsPerWg), block(256);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        prefix_proj_ln_k<T><<<grid, block, lds, st>>>((u16*)out, (u16*)ln_in, keep_mask, mean, rstd, (const u16*)x, (const u16*)w,
                                                     (const u16*)bias, (const u16*)ln_w, (const u16*)ln_b, ws_y, counter, rows,
                                                     enc_dim, out_dim, eps, p, drop, offset, rng_state);
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}

extern "C" int mm_dropout_bwd(void* dx, const void* g, const uint8_t* keep_mask, size_t n, float p, int dtype, void* stream) {
    MM_CHECK_ARG(dx && g && keep_mask, "mm_dropout_bwd: null pointer");
    MM_CHECK_ARG(n % 8 == 0 && aligned16(dx) && aligned16(g) && (((uintptr_t)keep_mask) & 7) == 0,
                 "mm_dropout_bwd: n must be a multiple of 8 and the pointers 16-byte (mask: 8-byte) aligned");
    MM_CHECK_ARG(p >= 0.f && p < 1.f, "mm_dropout_bwd: dropout probability %g outside [0, 1)", (double)p);
    const size_t n8 = n / 8;
    if (n8 == 0) return MM_OK;
    size_t g_ = (n8 + 255) / 256;
    if (g_ > 2048) g_ = 2048;
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        dropout_bwd_k<T><<<dim3((unsigned)g_), dim3(256), 0, (hipStream_t)stream>>>((u16*)dx, (const u16*)g, keep_mask, n8, 1.0f / (1.0f - p));
        MM_CHECK_LAUNCH();
        return (int)MM_OK;
    });
}
