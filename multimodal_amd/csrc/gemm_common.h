// gemm_common.h - pieces shared by the MFMA GEMM kernels (gemm.hip: 128x128 tile; gemm256.hip: 256-row pipelined tile).
#pragma once
#include "common.h"

namespace mm {

typedef __attribute__((address_space(3))) void lds_void;

template <typename T> struct Mfma;
template <> struct Mfma<BF16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

__device__ __forceinline__ float gelu_tanh_dev(float x) { return gelu_tanh_fast(x); }
__device__ __forceinline__ float gelu_erf_dev(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// derivatives: the formulas of bias_gelu_back (reference megatron/model/activations.py:75-86) / of erf-GeLU
__device__ __forceinline__ float gelu_tanh_grad_dev(float x) { return gelu_tanh_grad_fast(x); }
__device__ __forceinline__ float gelu_erf_grad_dev(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
// MM_EPI_BIAS_ROTARY (mm_gemm_nt_qkv_rotary only): bias, then the rotary embedding on the q and k slices of the fused
// QKV activation.  The kernels' `aux` / `residual` pointer slots carry the cos / sin tables; EpiExtra the integers.
struct EpiExtra {
    uint32_t magic;  // floor(2^32 / b) + 1: token row m -> sequence position m / b by one v_mul_hi_u32 (b > 1, M * b < 2^32)
    int b, d, offset;
};
// which epilogues read `bias` / use the `residual` slot as a second INPUT (not an addend)
constexpr bool epi_is_gelu_fwd(int e) { return e == MM_EPI_BIAS_GELU_TANH || e == MM_EPI_BIAS_GELU_ERF; }
constexpr bool epi_is_gelu_bwd(int e) { return e == MM_EPI_GELU_TANH_BWD || e == MM_EPI_GELU_ERF_BWD; }
constexpr bool epi_is_bwd(int e) { return e == MM_EPI_RELU_BWD || epi_is_gelu_bwd(e); }

// XCD-aware tile mapping (bijective for any grid size): the workgroups of one XCD (blockIdx % 8) get a contiguous run of
// tile ids, walked in groups of `group_m` tile-rows so that the run shares A row-panels / B column-panels in that XCD's L2.
__device__ __forceinline__ void tile_coords(int bid, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int group_sz = group_m * tiles_n;
    const int group = bid / group_sz;
    const int first_m = group * group_m;
    const int gm = min(tiles_m - first_m, group_m);
    tm = first_m + (bid % group_sz) % gm;
    tn = (bid % group_sz) / gm;
}

// Warm the Infinity Cache with the B operand (a weight matrix, [N, K] rows of K 16-bit values).  A weight is read once per
// direction per step and the step's working set (~1 GB of weights + activations per pass) is far beyond the 256 MiB cache, so
// every GEMM finds its weight in HBM: measured on the h->4h GEMM (M=8192 N=4096 K=1024, tools/ab_gemm_cold2.py, medians) all
// operands cached 71.6 us, weight cold 79.8, everything cold 90.5, everything cold but A and W read once just before 73.7 -
// the K loop keeps only one or two 64-deep tiles in flight (LDS is full), enough for the cache's latency, not for HBM's.
// So the first `nblk` workgroups of a launch read one dword of every 128-byte line of B between them, before their first tile
// fetch: the whole matrix is on its way into the cache while the first tiles start, at the cost of one or two loads per
// thread (weight cold 79.8 -> 73.2 us, everything cold 90.5 -> 82.2).
// The loads are LDS-DMA into a 256-byte per-wave dump area (`dump`, wave-uniform): a load into a REGISTER would have to keep
// that register reserved until it lands, which plain HIP cannot promise - under register pressure the compiler moves or
// spills the (to it, already valid) value and hands the register to other code, and the late load then overwrites a live
// address (it did: GPU memory fault).  Nothing ever reads the dump area.  Out-of-range lanes re-read line 0.
__device__ __forceinline__ void touch_weight(const u16* __restrict__ B, int N, int K, int ldb, int bid, int nblk, int tid, int nthr,
                                             char* dump) {
    const int lines_per_row = (K * 2 + 127) >> 7;
    const long total_l = (long)N * lines_per_row;
    const int stride = nblk * nthr;
    if (bid >= nblk || total_l > 2l * stride) return;  // (workgroup-uniform) at most two lines per thread: 32 MiB at 256 x 512 threads
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int total = (int)total_l;
        int L = bid * nthr + tid + i * stride;
        if (i == 1 && total <= stride) break;       // uniform
        if (L >= total) L = 0;
        const u16* p = B + (size_t)(L / lines_per_row) * ldb + min((int)(L % lines_per_row) * 64, K - 2);
        __builtin_amdgcn_global_load_lds((const void*)p, (lds_void*)dump, 4, 0, 0);
    }
}

// The same for the second INPUT of a backward epilogue (the saved forward tensor in the residual slot: 64 MiB of pre-activations
// written one forward pass earlier, i.e. in HBM): a 512-thread workgroup reads one dword of every 128-byte line of its own
// 256 x 256 tile (1024 lines, two per thread) before its first tile fetch, so that the epilogue finds the tile in the cache.
__device__ __forceinline__ void touch_tile256(const u16* __restrict__ X, int ld, int m0, int n0, int M, int N, int tid, char* dump) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int L = tid + 512 * i;                  // line id inside the tile: row L >> 2, 128-byte piece L & 3
        const int m = min(m0 + (L >> 2), M - 1), n = min(n0 + (L & 3) * 64, N - 2);
        __builtin_amdgcn_global_load_lds((const void*)(X + (size_t)m * ld + n), (lds_void*)dump, 4, 0, 0);
    }
}

// bias values of 4 consecutive columns
template <typename T> __device__ __forceinline__ void load_bias4(float (&bv)[4], const u16* __restrict__ bias, int n0) {
    const u32x2 braw = *reinterpret_cast<const u32x2*>(bias + n0);
    bv[0] = to_f32<T>((u16)(braw[0] & 0xffffu)); bv[1] = to_f32<T>((u16)(braw[0] >> 16));
    bv[2] = to_f32<T>((u16)(braw[1] & 0xffffu)); bv[3] = to_f32<T>((u16)(braw[1] >> 16));
}

// phase 2 of epilogue_staged (see below): lane = (row l>>3, 16-byte chunk l&7) of each 8-row group of the staged image.
// INTERIOR: the wave's whole (MI*16) x 64 tile lies inside C.
template <typename T, int EPI, bool RES, int MI, bool INTERIOR>
__device__ __forceinline__ void epilogue_phase2(const char* wl, int m_base, int n, int M, int N, int ldc, u16* __restrict__ C,
                                                u16* __restrict__ aux, const u16* __restrict__ bias,
                                                const u16* __restrict__ Rsd, int lane) {
    constexpr bool GELU = epi_is_gelu_fwd(EPI), GBWD = epi_is_gelu_bwd(EPI), RBWD = (EPI == MM_EPI_RELU_BWD);
    const int c = lane & 7;
    const bool full = INTERIOR || n + 8 <= N, half = !full && n + 4 <= N;  // N % 4 == 0 is a precondition of the MFMA kernels
    float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (GELU || GBWD) {
        if (full) {
            unpack8<T>(*reinterpret_cast<const u32x4*>(bias + n), b8);
        } else if (half) {
            float b4[4];
            load_bias4<T>(b4, bias, n);
#pragma unroll
            for (int j = 0; j < 4; ++j) b8[j] = b4[j];
        }
    }
#pragma unroll
    for (int i = 0; i < MI * 2; ++i) {
        const int row = i * 8 + (lane >> 3);
        const int m = m_base + row;
        u32x4 v = *reinterpret_cast<const u32x4*>(wl + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
        if (!INTERIOR && (m >= M || !(full || half))) continue;
        const size_t off = (size_t)m * ldc + n;
        if (GELU) {
            // the saved pre-activation is next read one whole forward pass later: a streaming store keeps it from
            // displacing the operands in L2
            if (full) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(aux + off));
            else *reinterpret_cast<u32x2*>(aux + off) = u32x2{v[0], v[1]};
        }
        if (GELU || RES) {
            float f[8], r8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            unpack8<T>(v, f);
            if (RES) {
                if (full) {
                    // a backward epilogue's saved forward tensor is read exactly once: streaming load
                    unpack8<T>(GBWD ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Rsd + off))
                                    : *reinterpret_cast<const u32x4*>(Rsd + off), r8);
                } else {
                    const u32x2 rr = *reinterpret_cast<const u32x2*>(Rsd + off);
                    unpack8<T>(u32x4{rr[0], rr[1], 0u, 0u}, r8);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = f[j];
                if (GELU) {
                    x += b8[j];
                    x = (EPI == MM_EPI_BIAS_GELU_TANH) ? gelu_tanh_dev(x) : gelu_erf_dev(rnd<T>(x));
                }
                if (RBWD) {
                    x = r8[j] > 0.f ? x : 0.f;  // r8 = the forward ReLU output
                } else if (GBWD) {
                    float u = r8[j] + b8[j];    // r8 = the forward pre-bias product
                    if (EPI == MM_EPI_GELU_ERF_BWD) u = rnd<T>(u);
                    x *= (EPI == MM_EPI_GELU_TANH_BWD) ? gelu_tanh_grad_dev(u) : gelu_erf_grad_dev(u);
                } else if (RES) {
                    x = (GELU ? rnd<T>(x) : x) + r8[j];
                }
                f[j] = x;
            }
            v = pack8<T>(f);
        }
        if (full) *reinterpret_cast<u32x4*>(C + off) = v;
        else *reinterpret_cast<u32x2*>(C + off) = u32x2{v[0], v[1]};
    }
}

// Rotary embedding on a wave's staged (MI*16) x 64 tile, between the two phases of epilogue_staged.  Head size 64 and
// 64-column wave tiles make every tile exactly one q, k or v slice of one head (fused layout [np][q | k | v],
// megatron/model/transformer.py:555-564); v tiles skip this (wave-uniform).  Work item = (row, 16-byte piece pc of the
// first half) together with its partner piece in the second half, one item per lane and pass, so every lane does useful
// work: with d = 16 a 64-row tile is ONE pass.  Arithmetic and rounding are those of rotary_qk_k (elementwise.hip), i.e. of
// apply_rotary_pos_emb (megatron/model/positional_embeddings.py:76-92) on the rounded GEMM output.
template <typename T, int MI>
__device__ __forceinline__ void epilogue_rotary_pass(char* wl, int m_base, int n_base, int M, const u16* __restrict__ cos_t,
                                                     const u16* __restrict__ sin_t, const EpiExtra& ex, int lane) {
    // No fma contraction here: for fp16 the compiler narrows these fp32 expressions to native half arithmetic and would then
    // fuse rnd(a*b) +- rnd(c*d) into one half fma, skipping a rounding the eager reference performs.
#pragma clang fp contract(off)
    if ((n_base >> 6) % 3 == 2) return;  // a v slice
    const int half = ex.d >> 1;
    const int lp = (half == 8) ? 0 : (half == 16 ? 1 : 2);  // log2(pieces per half): d in {16, 32, 64}
    const int items = (MI * 16) << lp;
    for (int it = lane; it < items; it += 64) {
        const int row = it >> lp, pc = it & ((1 << lp) - 1);
        const int m = min(m_base + row, M - 1);
        const int pos = (int)(ex.b == 1 ? (uint32_t)m : __umulhi((uint32_t)m, ex.magic)) + ex.offset;
        const int sw = (row >> 1) & 7;
        u32x4* p1 = reinterpret_cast<u32x4*>(wl + row * 128 + ((pc ^ sw) << 4));
        u32x4* p2 = reinterpret_cast<u32x4*>(wl + row * 128 + (((pc + (1 << lp)) ^ sw) << 4));
        const u16* cp = cos_t + (size_t)pos * ex.d + pc * 8;
        const u16* sp = sin_t + (size_t)pos * ex.d + pc * 8;
        float x1[8], x2[8], c1[8], c2[8], s1[8], s2[8], o1[8], o2[8];
        unpack8<T>(*reinterpret_cast<const u32x4*>(cp), c1);
        unpack8<T>(*reinterpret_cast<const u32x4*>(cp + half), c2);
        unpack8<T>(*reinterpret_cast<const u32x4*>(sp), s1);
        unpack8<T>(*reinterpret_cast<const u32x4*>(sp + half), s2);
        unpack8<T>(*p1, x1);
        unpack8<T>(*p2, x2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o1[j] = rnd<T>(x1[j] * c1[j]) + rnd<T>(-x2[j] * s1[j]);
            o2[j] = rnd<T>(x2[j] * c2[j]) + rnd<T>(x1[j] * s2[j]);
        }
        *p1 = pack8<T>(o1);
        *p2 = pack8<T>(o2);
    }
}

// ---- coalesced epilogue of one wave's (MI*16) x 64 output tile, staged through a wave-private LDS region.
// The MFMA accumulator layout gives a lane 4 consecutive columns (8 B) of 16 different rows: stored directly, one wave
// instruction writes 16 separate 32-byte pieces and every 128-byte line of C is touched by 4 instructions - measured, that
// store pattern costs as much as the whole K loop at K = 1024.  Instead:
//   phase 1: epilogue math up to the first rounding, 8-byte ds_write into a [rows][128 B] image (16-byte chunk index
//            XOR (row>>1)&7: conflict-free for these writes and for the reads below);
//   phase 2: lane = (row l>>3, chunk l&7) reads 16 B back; one wave instruction now covers 8 full 128-byte rows of C.
//            Residual add, the GeLU activation and the second output (aux) happen here, on 16-byte coalesced accesses.
// Backward epilogues (MM_EPI_RELU_BWD / MM_EPI_GELU_*_BWD) take the saved forward tensor in the residual slot and apply the
// activation's derivative to the rounded product, i.e. they are the data-gradient GEMM + the elementwise backward kernel.
// Rounding order is the reference's eager order (product rounded, + bias rounded, activation, + residual rounded).  `wl` must not be shared with another wave; all
// LDS traffic of the K loop must have completed for every wave (barrier) before the call.
// MI0 / MIC: stage and store only the 16-row blocks MI0 .. MI0 + MIC - 1 of the accumulator (m_base = the first row of THAT part; the
// staged image is MIC*16 rows), for callers whose staging area is smaller than the wave's whole tile.
template <typename T, int EPI, bool RES, int MI, int MI0 = 0, int MIC = MI>
__device__ __forceinline__ void epilogue_staged(const f32x4 (&acc)[4][MI], char* wl, int m_base, int n_base, int M, int N, int ldc,
                                                u16* __restrict__ C, u16* __restrict__ aux, const u16* __restrict__ bias,
                                                const u16* __restrict__ Rsd, int lane, const EpiExtra& ex = EpiExtra{}) {
    static_assert(!epi_is_bwd(EPI) || RES, "backward epilogues take their second input in the residual slot");
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n0 = n_base + ni * 16 + fq * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if ((EPI == MM_EPI_BIAS || EPI == MM_EPI_BIAS_RELU || EPI == MM_EPI_BIAS_ROTARY) && n0 < N) load_bias4<T>(bv, bias, n0);
#pragma unroll
        for (int mi = 0; mi < MIC; ++mi) {
            const int row = mi * 16 + frow;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = acc[ni][MI0 + mi][j];
                if (EPI == MM_EPI_BIAS || EPI == MM_EPI_BIAS_ROTARY) o[j] += bv[j];
                if (EPI == MM_EPI_BIAS_RELU) o[j] = fmaxf(rnd<T>(o[j] + bv[j]), 0.f);
            }
            const int phys = (ni * 2 + (fq >> 1)) ^ ((row >> 1) & 7);
            *reinterpret_cast<u32x2*>(wl + row * 128 + phys * 16 + (fq & 1) * 8) =
                u32x2{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
        }
    }
    if constexpr (EPI == MM_EPI_BIAS_ROTARY) epilogue_rotary_pass<T, MIC>(wl, m_base, n_base, M, aux, Rsd, ex, lane);
    const int c = lane & 7, n = n_base + c * 8;
    // interior tiles (all of M = 8192-class problems) take a predicate-free copy of phase 2: no exec-mask branches, so the
    // compiler batches the LDS reads / residual loads of several row groups ahead of the math
    if (m_base + MIC * 16 <= M && n_base + 64 <= N)
        epilogue_phase2<T, EPI, RES, MIC, true>(wl, m_base, n, M, N, ldc, C, aux, bias, Rsd, lane);
    else
        epilogue_phase2<T, EPI, RES, MIC, false>(wl, m_base, n, M, N, ldc, C, aux, bias, Rsd, lane);
}

// gemm256.hip: returns MM_OK when it ran the problem, 1 when the shape is not one it takes (caller falls back), <0 on error.
// gemm_pp.hip (the ping-pong 256 x {256,128} kernel): same return convention; gemm_pp_tile = the tile width it would use (0: declines)
int gemm_pp_tile(int M, int N, int K, int lda, int ldb);
int gemm_pp_try(void* C, void* aux, const void* A, const void* B, const void* bias, const void* residual, int M, int N, int K,
                int lda, int ldb, int ldc, int epilogue, int dtype, hipStream_t st, const EpiExtra& ex = EpiExtra{});
// split-K form of the ping-pong kernel (two K-slices per 256 x 256 tile): workspace bytes it needs (0: not a split-K shape)
size_t gemm_pp_splitk_workspace(int M, int N, int K, int lda, int ldb);
int gemm_pp_splitk(void* C, const void* A, const void* B, const void* bias, int M, int N, int K, int lda, int ldb, int ldc,
                   void* ws, int epilogue, int dtype, hipStream_t st);
bool gemm256_takes(int M, int N, int K, int lda, int ldb);
int gemm256_try(void* C, void* aux, const void* A, const void* B, const void* bias, const void* residual, int M, int N, int K,
                int lda, int ldb, int ldc, int epilogue, int dtype, hipStream_t st, const EpiExtra& ex = EpiExtra{});

}  // namespace mm
