// gemm_tn.hip - weight-gradient GEMM  C[N, Kp] = A[M, N]^T * B[M, Kp]  (contraction over the TOKEN dimension M), split-K.
//
// Replaces the dW = dY^T X half of F.linear's backward (ATen, used by megatron/mpu/layers.py:557,745 and the adapter's
// nn.Linear, megatron/model/adapter.py:18-24).  Both operands are row-major with the contraction index on the ROW, i.e.
// strided for an MFMA fragment.  Instead of transposing dY and X in HBM (two extra 16 MB round trips per GEMM), the
// 64-row tiles are staged row-major into the padded LDS image of mfma32.h and BOTH MFMA operands are read with
// ds_read_b64_tr_b16 (hardware transpose, conflict-free on that image).  Both reads use the same k-order, so the
// permutation inside a 16-row k-step cancels in the sum.
//
// With the LM frozen the only weight gradients are the adapters' [h/8, h] and [h, h/8] matrices: 8 output tiles against a
// contraction of 8192 tokens.  The kernel therefore splits M over `splits` workgroups per tile (grid = tiles x splits,
// >= 256 workgroups for 256 CUs), each writing an fp32 slab; a second pass sums the slabs in a fixed order
// (bit-reproducible, no atomics) and rounds once to the 16-bit output.
//
// The same backward also needs db = column sums of dY (= A here).  The workgroups of the first column-tile of C compute
// them from the A tiles they have staged anyway, as one more MFMA per A fragment against an all-ones operand (every
// column of that product is the column sum), so the separate column-sum kernels and their pass over dY disappear.
#include "common.h"
#include "mfma32.h"

namespace mm {

constexpr int TN_BT = 128;  // output tile (rows of C = columns of A) x (cols of C = columns of B)

// stage rows [row0, row0+64) x cols [col0, col0+128) of X (ld elements per row) into a Tile<128> image
__device__ __forceinline__ void tn_tile_load(u32x4 (&regs)[4], const u16* __restrict__ X, int ld, int row0, int col0, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        const int row = c >> 4, ch = c & 15;
        regs[i] = *reinterpret_cast<const u32x4*>(X + (size_t)(row0 + row) * ld + col0 + ch * 8);
    }
}

template <typename T> struct Ones;
template <> struct Ones<BF16> { static __device__ __forceinline__ bf16x8 get() { const __bf16 o = (__bf16)1.0f; return bf16x8{o, o, o, o, o, o, o, o}; } };
template <> struct Ones<F16> { static __device__ __forceinline__ f16x8 get() { const _Float16 o = (_Float16)1.0f; return f16x8{o, o, o, o, o, o, o, o}; } };

// SLAB: fp32 partials to ws [gridDim.y][N*Kp (+ N column sums when CS)]; otherwise direct 16-bit output.
template <typename T, bool SLAB, bool CS>
__global__ __launch_bounds__(256) void gemm_tn_k(u16* __restrict__ C, u16* __restrict__ colsum, float* __restrict__ ws,
                                                 const u16* __restrict__ A, const u16* __restrict__ B, int M, int N, int Kp,
                                                 int lda, int ldb, int ldc, int tiles_k, int rows_per_split) {
    typedef Tile<128> TL;
    typedef typename M32<T>::frag frag;
    __shared__ __attribute__((aligned(16))) char smem[4 * TL::BYTES];  // [buf][A | B]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
    const int n0 = tn * TN_BT, k0 = tk * TN_BT;
    const int m_begin = blockIdx.y * rows_per_split;
    const int ntile = rows_per_split / 64;
    const int wr = wave >> 1, wc = wave & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const bool do_cs = CS && tk == 0 && wc == 0;  // wave-uniform
    f32x16 cs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) cs[i][e] = 0.f;
    const frag ones = Ones<T>::get();

    u32x4 areg[4], breg[4];
    tn_tile_load(areg, A, lda, m_begin, n0, tid);
    tn_tile_load(breg, B, ldb, m_begin, k0, tid);
    tile_store<128>(smem, areg, tid);
    tile_store<128>(smem + TL::BYTES, breg, tid);
    __syncthreads();

    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1;
        const char* as = smem + cur * 2 * TL::BYTES;
        const char* bs = as + TL::BYTES;
        if (t + 1 < ntile) {
            tn_tile_load(areg, A, lda, m_begin + (t + 1) * 64, n0, tid);
            tn_tile_load(breg, B, ldb, m_begin + (t + 1) * 64, k0, tid);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = tr_frag<T, 128>(as, 16 * ks, wr * 2 + i, lane);
                fb[i] = tr_frag<T, 128>(bs, 16 * ks, wc * 2 + i, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = M32<T>::run(fa[i], fb[j], acc[i][j]);
            if (do_cs) {
#pragma unroll
                for (int i = 0; i < 2; ++i) cs[i] = M32<T>::run(fa[i], ones, cs[i]);
            }
        }
        if (t + 1 < ntile) {
            char* nx = smem + (cur ^ 1) * 2 * TL::BYTES;
            tile_store<128>(nx, areg, tid);
            tile_store<128>(nx + TL::BYTES, breg, tid);
        }
        __syncthreads();
    }
    // D[row = n (registers)][col = kp (lane)]
    const size_t slab = (size_t)N * Kp + (CS ? N : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kp = k0 + wc * 64 + 32 * j + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wr * 64 + 32 * i + acc_row(e, hh);
                if (SLAB) ws[(size_t)blockIdx.y * slab + (size_t)n * Kp + kp] = acc[i][j][e];
                else C[(size_t)n * ldc + kp] = from_f32<T>(acc[i][j][e]);
            }
        }
    if (do_cs && r == 0) {  // every column of cs holds the sums: lanes 0 and 32 own all 32 rows between them
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wr * 64 + 32 * i + acc_row(e, hh);
                if (SLAB) ws[(size_t)blockIdx.y * slab + (size_t)N * Kp + n] = cs[i][e];
                else colsum[n] = from_f32<T>(cs[i][e]);
            }
    }
}

// C[n, kp] = sum_s ws[s][n][kp] (fixed order), 4 elements per thread; elements past N*Kp are the column sums
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_k(u16* __restrict__ C, u16* __restrict__ colsum, const float* __restrict__ ws,
                                                       int N, int Kp, int ldc, int splits, int ncs) {
    const size_t nk = (size_t)N * Kp, slab = nk + ncs;
    const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= slab) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(ws + i4);
    for (int k = 1; k < splits; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ws + (size_t)k * slab + i4);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    u16* o;
    if (i4 < nk) {
        const int n = (int)(i4 / Kp), kp = (int)(i4 % Kp);
        o = C + (size_t)n * ldc + kp;
    } else {
        o = colsum + (i4 - nk);
    }
    *reinterpret_cast<u32x2*>(o) = u32x2{(uint32_t)from_f32<T>(s[0]) | ((uint32_t)from_f32<T>(s[1]) << 16),
                                        (uint32_t)from_f32<T>(s[2]) | ((uint32_t)from_f32<T>(s[3]) << 16)};
}

}  // namespace mm

using namespace mm;

extern "C" int mm_gemm_tn_splits(int M, int N, int Kp) {
    if (M <= 0 || N <= 0 || Kp <= 0 || N % TN_BT || Kp % TN_BT || M % 64) return 0;  // 0: shape not supported by this kernel
    const int tiles = (N / TN_BT) * (Kp / TN_BT);
    int splits = 1;
    while (tiles * splits * 2 <= 256 && M % (64 * splits * 2) == 0 && M / (splits * 2) >= 128) splits *= 2;
    return splits;
}

extern "C" int mm_gemm_tn(void* C, void* a_colsum, float* workspace, const void* A, const void* B, int M, int N, int Kp, int lda,
                          int ldb, int ldc, int splits, int dtype, void* stream) {
    MM_CHECK_ARG(C && A && B, "mm_gemm_tn: null pointer");
    MM_CHECK_ARG(M > 0 && N > 0 && Kp > 0, "mm_gemm_tn: M=%d N=%d Kp=%d must be positive", M, N, Kp);
    MM_CHECK_ARG(N % TN_BT == 0 && Kp % TN_BT == 0, "mm_gemm_tn: N=%d and Kp=%d must be multiples of %d", N, Kp, TN_BT);
    MM_CHECK_ARG(splits >= 1 && M % (64 * splits) == 0, "mm_gemm_tn: M=%d must be a multiple of 64*splits (splits=%d)", M, splits);
    MM_CHECK_ARG(lda >= N && ldb >= Kp && ldc >= Kp && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0,
                 "mm_gemm_tn: bad leading dimensions lda=%d ldb=%d ldc=%d", lda, ldb, ldc);
    MM_CHECK_ARG(aligned16(A) && aligned16(B) && (((uintptr_t)C) & 7) == 0 && (((uintptr_t)a_colsum) & 7) == 0,
                 "mm_gemm_tn: A/B must be 16-byte, C/a_colsum 8-byte aligned");
    MM_CHECK_ARG(splits == 1 || (workspace && aligned16(workspace)),
                 "mm_gemm_tn: splits > 1 needs a 16-byte aligned fp32 workspace [splits, N*Kp + N]");
    MM_CHECK_ARG(splits <= 65535, "mm_gemm_tn: too many splits");
    hipStream_t st = (hipStream_t)stream;
    const int tiles_k = Kp / TN_BT;
    dim3 grid((N / TN_BT) * tiles_k, splits), block(256);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        u16* cs = (u16*)a_colsum;
        if (splits == 1) {
            if (cs) gemm_tn_k<T, false, true><<<grid, block, 0, st>>>((u16*)C, cs, nullptr, (const u16*)A, (const u16*)B, M, N, Kp, lda, ldb, ldc, tiles_k, M);
            else gemm_tn_k<T, false, false><<<grid, block, 0, st>>>((u16*)C, nullptr, nullptr, (const u16*)A, (const u16*)B, M, N, Kp, lda, ldb, ldc, tiles_k, M);
            MM_CHECK_LAUNCH();
        } else {
            if (cs) gemm_tn_k<T, true, true><<<grid, block, 0, st>>>((u16*)C, cs, workspace, (const u16*)A, (const u16*)B, M, N, Kp, lda, ldb, ldc, tiles_k, M / splits);
            else gemm_tn_k<T, true, false><<<grid, block, 0, st>>>((u16*)C, nullptr, workspace, (const u16*)A, (const u16*)B, M, N, Kp, lda, ldb, ldc, tiles_k, M / splits);
            MM_CHECK_LAUNCH();
            const size_t slab = (size_t)N * Kp + (cs ? N : 0);
            splitk_reduce_k<T><<<(unsigned)((slab / 4 + 255) / 256), 256, 0, st>>>((u16*)C, cs, workspace, N, Kp, ldc, splits, cs ? N : 0);
            MM_CHECK_LAUNCH();
        }
        return (int)MM_OK;
    });
}
