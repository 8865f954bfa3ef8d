// gemm.hip - bf16/f16 MFMA GEMM  C[M,N] = A[M,K] * B[N,K]^T  with fused epilogues (gfx950).
//
// Replaces F.linear inside ColumnParallelLinear / RowParallelLinear
// (reference megatron/mpu/layers.py:557,745) and the adapter's nn.Linear (megatron/model/adapter.py:18-24).
//
// Structure (v1, "128x128x64, one barrier per K-tile"):
//   * 256 threads = 4 waves, wave (wr, wc) owns a 64x64 sub-tile as 4x4 MFMA 16x16x32 accumulators;
//   * A and B K-tiles are staged global -> LDS by global_load_lds (16 B per lane, no VGPR round trip),
//     double-buffered: tile t+1 is in flight while tile t feeds the MFMAs;
//   * LDS image is lane-linear per wave-instruction (8 rows x 128 B); bank conflicts of the ds_read_b128
//     fragment reads are removed by XOR-swizzling the 16-byte chunk index with (row>>1)&7 on the
//     SOURCE address and on the READ address (both sides or neither);
//   * operands are swapped (MFMA A-operand = B tile) so that each lane ends up with 4 consecutive
//     output columns of one row -> 8-byte epilogue stores and 8-byte bias loads;
//   * blockIdx is remapped so that the blocks of one XCD (blockIdx % 8) walk neighbouring tiles
//     (shared A row-panels / B column-panels stay in that XCD's L2).
#include "gemm_common.h"

namespace mm {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

// stage one 128x64 tile (rows row0.., k-offset k0) of X (ld elements per row, nrows valid rows) into lds_tile
template <typename T>
__device__ __forceinline__ void stage_tile(char* lds_tile, const u16* __restrict__ X, int ld, int row0, int nrows, int k0,
                                           int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 32 + i * 8 + (lane >> 3);      // row inside the tile
        const int cl = lane & 7;                            // chunk position in LDS
        const int cg = cl ^ ((r >> 1) & 7);                 // global chunk that belongs there
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;                   // clamp: edge tiles re-read the last row (never stored)
        const u16* src = X + (size_t)gr * ld + k0 + cg * 8;
        char* dst = lds_tile + (wave * 32 + i * 8) * (BK * 2);  // wave-uniform; HW adds lane*16
        __builtin_amdgcn_global_load_lds((const void*)src, (lds_void*)dst, 16, 0, 0);
    }
}

template <typename T, int EPI, bool RES>
__global__ __launch_bounds__(256) void gemm_nt_128_k(u16* __restrict__ C, u16* __restrict__ aux, const u16* __restrict__ A,
                                                     const u16* __restrict__ B, const u16* __restrict__ bias,
                                                     const u16* __restrict__ Rsd, int M, int N, int K, int lda, int ldb,
                                                     int ldc, int tiles_m, int tiles_n, EpiExtra ex) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE_BYTES + 4 * 256];  // stages + the warming loads' dump area
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    int tm, tn;
    tile_coords(blockIdx.x, tiles_m, tiles_n, 8, tm, tn);
    const int bm = tm * BM, bn = tn * BN;

    const int wr = wave >> 1, wc = wave & 1;
    f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = K / BK;
    // the weight matrix on its way into the Infinity Cache (first 512 workgroups of the launch; gemm_common.h)
    touch_weight(B, N, K, ldb, blockIdx.x, min((int)gridDim.x, 512), tid, 256, smem + 2 * 2 * TILE_BYTES + wave * 256);
    // stage s: A tile at smem + s*2*TILE_BYTES, B tile right behind it
    stage_tile<T>(smem, A, lda, bm, M, 0, wave, lane);
    stage_tile<T>(smem + TILE_BYTES, B, ldb, bn, N, 0, wave, lane);
    __syncthreads();  // vmcnt(0) + barrier: tile 0 landed

    // per-lane fragment read offsets (bytes) inside a tile, for kk = 0; kk = 1 flips chunk bit 2
    const int frow = lane & 15, fq = lane >> 4;
    int offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + i * 16 + frow;
        const int rb = wc * 64 + i * 16 + frow;
        offA[i] = ra * (BK * 2) + ((fq ^ ((ra >> 1) & 7)) << 4);
        offB[i] = rb * (BK * 2) + ((fq ^ ((rb >> 1) & 7)) << 4);
    }

    typedef typename Mfma<T>::frag frag;
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            char* nxt = smem + (cur ^ 1) * (2 * TILE_BYTES);
            stage_tile<T>(nxt, A, lda, bm, M, (t + 1) * BK, wave, lane);
            stage_tile<T>(nxt + TILE_BYTES, B, ldb, bn, N, (t + 1) * BK, wave, lane);
        }
        const char* a_base = smem + cur * (2 * TILE_BYTES);
        const char* b_base = a_base + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            frag fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *reinterpret_cast<const frag*>(a_base + (offA[i] ^ (kk << 6)));
                fb[i] = *reinterpret_cast<const frag*>(b_base + (offB[i] ^ (kk << 6)));
            }
            // two workgroups share a CU: the MFMA block outranks the partner wave's address / epilogue vector work for the issue
            // port (one slot per 16 pipe cycles keeps the matrix pipe fed); -4..7 % on the N = h shapes, tools/ab_gemm_x.py
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = Mfma<T>::run(fb[ni], fa[mi], acc[ni][mi]);
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();  // next tile landed (vmcnt(0)) and everyone is done reading `cur`
    }

    // ---- epilogue (coalesced through a wave-private 8 KiB LDS region; the loop's last barrier retired all LDS traffic)
    epilogue_staged<T, EPI, RES, 4>(acc, smem + wave * 8192, bm + wr * 64, bn + wc * 64, M, N, ldc, C, aux, bias, Rsd, lane, ex);
}


// ---- generic fallback: any M, N, K, any leading dimension / alignment (tiny or ragged problems such as the image-prefix
// projection's weight gradient, whose contraction length is the 4-row micro-batch).  16x16 output tile, fp32 FMA.
template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_generic_k(u16* __restrict__ C, u16* __restrict__ aux, const u16* __restrict__ A,
                                                         const u16* __restrict__ B, const u16* __restrict__ bias,
                                                         const u16* __restrict__ Rsd, int M, int N, int K, int lda, int ldb,
                                                         int ldc, int epi) {
    __shared__ float sa[16][17], sb[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const int ka = k0 + tx;
        const int am = blockIdx.y * 16 + ty, bn_ = blockIdx.x * 16 + ty;
        sa[ty][tx] = (am < M && ka < K) ? to_f32<T>(A[(size_t)am * lda + ka]) : 0.f;
        sb[ty][tx] = (bn_ < N && ka < K) ? to_f32<T>(B[(size_t)bn_ * ldb + ka]) : 0.f;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc += sa[ty][kk] * sb[tx][kk];
        __syncthreads();
    }
    if (m >= M || n >= N) return;
    const size_t off = (size_t)m * ldc + n;
    float o = acc;
    const float bv = (epi != MM_EPI_NONE && epi != MM_EPI_RELU_BWD) ? to_f32<T>(bias[n]) : 0.f;
    if (epi == MM_EPI_BIAS_GELU_TANH || epi == MM_EPI_BIAS_GELU_ERF) {
        const u16 u = from_f32<T>(o);
        aux[off] = u;
        const float x = to_f32<T>(u) + bv;
        o = (epi == MM_EPI_BIAS_GELU_TANH) ? gelu_tanh_dev(x) : gelu_erf_dev(rnd<T>(x));
    } else if (epi == MM_EPI_BIAS) {
        o += bv;
    } else if (epi == MM_EPI_BIAS_RELU) {
        o = fmaxf(rnd<T>(o + bv), 0.f);
    }
    if (epi == MM_EPI_RELU_BWD) {
        o = to_f32<T>(Rsd[off]) > 0.f ? rnd<T>(o) : 0.f;
    } else if (epi == MM_EPI_GELU_TANH_BWD) {
        o = rnd<T>(o) * gelu_tanh_grad_dev(to_f32<T>(Rsd[off]) + bv);
    } else if (epi == MM_EPI_GELU_ERF_BWD) {
        o = rnd<T>(o) * gelu_erf_grad_dev(rnd<T>(to_f32<T>(Rsd[off]) + bv));
    } else if (Rsd) {
        o = rnd<T>(o) + to_f32<T>(Rsd[off]);
    }
    C[off] = from_f32<T>(o);
}

template <typename T, int EPI>
static int launch_gemm(void* C, void* aux, const void* A, const void* B, const void* bias, const void* residual, int M, int N,
                       int K, int lda, int ldb, int ldc, hipStream_t st, const EpiExtra& ex = EpiExtra{}) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n), block(256);
    // backward epilogues: the residual slot carries the saved forward tensor (always present); the rotary epilogue: the
    // aux / residual slots carry the cos / sin tables and there is no residual add
    if (EPI != MM_EPI_BIAS_ROTARY && (epi_is_bwd(EPI) || residual)) {
        if constexpr (EPI != MM_EPI_BIAS_ROTARY)
            gemm_nt_128_k<T, EPI, true><<<grid, block, 0, st>>>((u16*)C, (u16*)aux, (const u16*)A, (const u16*)B, (const u16*)bias,
                                                                (const u16*)residual, M, N, K, lda, ldb, ldc, tiles_m, tiles_n, ex);
    } else if constexpr (!epi_is_bwd(EPI)) {
        gemm_nt_128_k<T, EPI, false><<<grid, block, 0, st>>>((u16*)C, (u16*)aux, (const u16*)A, (const u16*)B, (const u16*)bias,
                                                             (const u16*)residual, M, N, K, lda, ldb, ldc, tiles_m, tiles_n, ex);
    }
    MM_CHECK_LAUNCH();
    return MM_OK;
}

}  // namespace mm

using namespace mm;

extern "C" int mm_gemm_nt(void* C, void* aux, const void* A, const void* B, const void* bias, const void* residual, int M,
                          int N, int K, int lda, int ldb, int ldc, int epilogue, int dtype, void* stream) {
    MM_CHECK_ARG(C && A && B, "mm_gemm_nt: null pointer");
    MM_CHECK_ARG(M > 0 && N > 0 && K > 0, "mm_gemm_nt: M=%d N=%d K=%d must be positive", M, N, K);
    MM_CHECK_ARG(lda >= K && ldb >= K && ldc >= N, "mm_gemm_nt: leading dimensions too small (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    MM_CHECK_ARG(epilogue >= MM_EPI_NONE && epilogue <= MM_EPI_GELU_ERF_BWD, "mm_gemm_nt: unknown epilogue %d", epilogue);
    if (epilogue != MM_EPI_NONE && epilogue != MM_EPI_RELU_BWD) MM_CHECK_ARG(bias != nullptr, "mm_gemm_nt: epilogue %d needs a bias", epilogue);
    if (epi_is_bwd(epilogue))
        MM_CHECK_ARG(residual != nullptr, "mm_gemm_nt: backward epilogue %d needs the saved forward tensor in `residual`", epilogue);
    if (epilogue == MM_EPI_BIAS_GELU_TANH || epilogue == MM_EPI_BIAS_GELU_ERF)
        MM_CHECK_ARG(aux != nullptr, "mm_gemm_nt: GeLU epilogues need an aux buffer");
    const bool al8 = ((((uintptr_t)C) | ((uintptr_t)aux) | ((uintptr_t)bias) | ((uintptr_t)residual)) & 7) == 0;
    const bool mfma_ok = (K % 64 == 0) && (N % 4 == 0) && (lda % 8 == 0) && (ldb % 8 == 0) && (ldc % 4 == 0) &&
                         aligned16(A) && aligned16(B) && al8;
    if (!mfma_ok) {
        hipStream_t gst = (hipStream_t)stream;
        return dispatch_dtype(dtype, [&](auto t) -> int {
            using T = decltype(t);
            gemm_nt_generic_k<T><<<dim3((N + 15) / 16, (M + 15) / 16), 256, 0, gst>>>(
                (u16*)C, (u16*)aux, (const u16*)A, (const u16*)B, (const u16*)bias, (const u16*)residual, M, N, K, lda, ldb, ldc, epilogue);
            MM_CHECK_LAUNCH();
            return (int)MM_OK;
        });
    }
    hipStream_t st = (hipStream_t)stream;
    {   // large shapes: the ping-pong 256-row kernel (gemm_pp.hip); it declines (returns 1) what it does not take
        const int rc = gemm_pp_try(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, epilogue, dtype, st);
        if (rc <= 0) return rc;
    }
#ifdef MM_DEV
    {   // the earlier pipelined 256x256 kernel (gemm256.hip): developer library only, for same-process A/B (MM_GEMMPP=-1)
        const int rc = gemm256_try(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, epilogue, dtype, st);
        if (rc <= 0) return rc;
    }
#endif
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        switch (epilogue) {
            case MM_EPI_NONE: return launch_gemm<T, MM_EPI_NONE>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st);
            case MM_EPI_BIAS: return launch_gemm<T, MM_EPI_BIAS>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st);
            case MM_EPI_BIAS_GELU_TANH: return launch_gemm<T, MM_EPI_BIAS_GELU_TANH>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st);
            case MM_EPI_BIAS_GELU_ERF: return launch_gemm<T, MM_EPI_BIAS_GELU_ERF>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st);
            case MM_EPI_BIAS_RELU: return launch_gemm<T, MM_EPI_BIAS_RELU>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st);
            case MM_EPI_RELU_BWD: return launch_gemm<T, MM_EPI_RELU_BWD>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st);
            case MM_EPI_GELU_TANH_BWD: return launch_gemm<T, MM_EPI_GELU_TANH_BWD>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st);
            default: return launch_gemm<T, MM_EPI_GELU_ERF_BWD>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st);
        }
    });
}

// C = A B^T (+ bias) by the split-K form of the ping-pong kernel (csrc/gemm_pp.hip): for the shapes whose 256 x 256 tiles would
// fill half the chip or less (the N = h class at M = 8192, the LM head's data gradient with K = 50304).
extern "C" size_t mm_gemm_nt_splitk_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return gemm_pp_splitk_workspace(M, N, K, K, K);
}

extern "C" int mm_gemm_nt_splitk(void* C, const void* A, const void* B, const void* bias, void* workspace, int M, int N, int K,
                                 int lda, int ldb, int ldc, int epilogue, int dtype, void* stream) {
    MM_CHECK_ARG(C && A && B && workspace, "mm_gemm_nt_splitk: null pointer");
    MM_CHECK_ARG(M > 0 && N > 0 && K > 0, "mm_gemm_nt_splitk: M=%d N=%d K=%d must be positive", M, N, K);
    MM_CHECK_ARG(lda >= K && ldb >= K && ldc >= N, "mm_gemm_nt_splitk: leading dimensions too small (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    MM_CHECK_ARG(epilogue == MM_EPI_NONE && bias == nullptr, "mm_gemm_nt_splitk: epilogue %d not supported (MM_EPI_NONE without a bias only)", epilogue);
    MM_CHECK_ARG(gemm_pp_splitk_workspace(M, N, K, lda, ldb) != 0,
                 "mm_gemm_nt_splitk: M=%d N=%d K=%d is not a split-K shape (mm_gemm_nt_splitk_workspace_bytes returns 0): use mm_gemm_nt", M, N, K);
    MM_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && aligned16(A) && aligned16(B) && aligned16(workspace) &&
                     ((((uintptr_t)C) | ((uintptr_t)bias)) & 7) == 0, "mm_gemm_nt_splitk: operands must be 16-byte aligned");
    return gemm_pp_splitk(C, A, B, bias, M, N, K, lda, ldb, ldc, workspace, epilogue, dtype, (hipStream_t)stream);
}

extern "C" const char* mm_gemm_nt_kernel_name(int M, int N, int K, int epilogue) {
    (void)epilogue;
    if (M <= 0 || N <= 0 || K <= 0) return "none";
    if (K % 64 || N % 4) return "gemm_nt_generic_k";
    const int bn = gemm_pp_tile(M, N, K, K, K);
    if (bn) return bn == 256 ? "gemm_nt_pp_k<256>" : "gemm_nt_pp_k<128>";
#ifdef MM_DEV
    if (gemm256_takes(M, N, K, K, K)) return "gemm_nt_256_k";
#endif
    return "gemm_nt_128_k";
}

extern "C" int mm_gemm_nt_qkv_rotary(void* C, const void* A, const void* B, const void* bias, const void* cos, const void* sin,
                                     int M, int N, int K, int lda, int ldb, int ldc, int b, int hn, int d, int offset, int dtype,
                                     void* stream) {
    MM_CHECK_ARG(C && A && B && bias && cos && sin, "mm_gemm_nt_qkv_rotary: null pointer");
    MM_CHECK_ARG(M > 0 && N > 0 && K > 0 && b > 0, "mm_gemm_nt_qkv_rotary: M=%d N=%d K=%d b=%d must be positive", M, N, K, b);
    MM_CHECK_ARG(lda >= K && ldb >= K && ldc >= N, "mm_gemm_nt_qkv_rotary: leading dimensions too small (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    MM_CHECK_ARG(offset >= 0 && d > 0 && d <= hn, "mm_gemm_nt_qkv_rotary: rotary dims %d / offset %d out of range (hn=%d)", d, offset, hn);
    const bool al = ((((uintptr_t)C) | ((uintptr_t)bias)) & 7) == 0 && aligned16(A) && aligned16(B) && aligned16(cos) && aligned16(sin);
    const bool ok = hn == 64 && (d == 16 || d == 32 || d == 64) && N % 192 == 0 && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
                    ldc % 4 == 0 && al && (uint64_t)M * (uint64_t)b < (1ull << 32);
    if (!ok) {
        mm::set_error("mm_gemm_nt_qkv_rotary: not a fused shape (hn=%d d=%d N=%d K=%d b=%d): use mm_gemm_nt + mm_rotary_qk_inplace", hn, d, N, K, b);
        return MM_EUNSUPPORTED;
    }
    const EpiExtra ex{b > 1 ? (uint32_t)((1ull << 32) / (uint64_t)b) + 1u : 0u, b, d, offset};
    hipStream_t st = (hipStream_t)stream;
    int rc = gemm_pp_try(C, const_cast<void*>(cos), A, B, bias, sin, M, N, K, lda, ldb, ldc, MM_EPI_BIAS_ROTARY, dtype, st, ex);
    if (rc <= 0) return rc;
#ifdef MM_DEV
    rc = gemm256_try(C, const_cast<void*>(cos), A, B, bias, sin, M, N, K, lda, ldb, ldc, MM_EPI_BIAS_ROTARY, dtype, st, ex);
    if (rc <= 0) return rc;
#endif
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        return launch_gemm<T, MM_EPI_BIAS_ROTARY>(C, const_cast<void*>(cos), A, B, bias, sin, M, N, K, lda, ldb, ldc, st, ex);
    });
}
