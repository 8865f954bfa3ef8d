// gemm256.hip - pipelined bf16/f16 MFMA GEMM  C[M,N] = A[M,K] * B[N,K]^T  with 256x256 tiles (gfx950).
//
// Same contract and epilogues as gemm.hip (reference megatron/mpu/layers.py:557,745 F.linear); this kernel takes the
// shapes that fill the chip with 256x256 tiles (the h->4h / 4h->h data-gradient GEMMs and the LM head at the 410M
// configuration).  Both kernels are bound by the L2 -> LDS fill rate of a CU (~35 B/clk measured), not by the MFMA
// pipe: a 256x256x64 step moves 64 KiB per 8.4 MFLOP where the 128x128 tile moves 32 KiB per 2.1 MFLOP.
//
//   * one 8-wave workgroup per CU (two waves per SIMD), waves 2(M) x 4(N), each wave owns a 128x64 output tile;
//   * A/B K-tiles go global -> LDS with global_load_lds (16 B per lane, lane-linear image, XOR chunk swizzle applied on
//     the source address and on the read address), two 64 KiB stages, and STAY IN FLIGHT across the per-K-tile barrier:
//     the only wait on them is the s_waitcnt just before the barrier that publishes a tile, one whole K-tile after
//     the issue;
//   * fragment reads are inline-asm ds_read_b128, so the compiler does not serialise them behind the in-flight LDS-DMA,
//     and they are register double-buffered: the 12 reads of work item i+1 (a 32-deep k-step) are issued before the
//     32 MFMAs of item i and waited for with a counted lgkmcnt;
//   * one raw s_barrier per K-tile; MFMA clusters run at raised wave priority;
//   * operands swapped (MFMA A-operand = B tile), epilogue staged through LDS for full-line stores (gemm_common.h).
// Hazard bookkeeping (why each wait is where it is) is written next to the loop.
// (A 256x128 variant with the two K halves of each tile on the two waves of a SIMD was built and measured: it ties the
//  128x128 kernel at N = 1024 - both sit on the same L2 fill bound - and was dropped.)
#include "gemm_common.h"

namespace mm {

constexpr int G_BM = 256;
constexpr int G_ROWB = 128;  // bytes per tile row (64 k-elements)

constexpr int G_BN = 256;
struct G {
    static constexpr int A_BYTES = G_BM * G_ROWB;
    static constexpr int B_BYTES = G_BN * G_ROWB;
    static constexpr int STAGE = A_BYTES + B_BYTES;  // 64 KiB
    static constexpr int LDS = STAGE * 2;
    static constexpr int NLA = G_BM / 64, NLB = G_BN / 64, NL = NLA + NLB;  // global_load_lds per thread per K-tile
};

// s_waitcnt immediate (gfx9 encoding): vm = 63 / lgkm = 15 mean "do not wait on this counter"
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 0xF) | (7 << 4) | ((lgkm & 0xF) << 8) | (((vm >> 4) & 3) << 14); }
#define MM_WAIT(vm, lgkm) __builtin_amdgcn_s_waitcnt(waitcnt_imm(vm, lgkm))

template <int OFF> __device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
    u32x4 d;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
    return d;
}

// one work item = the fragments of a wave's 128x64 tile for one 32-deep k-step: 4 B-operand + 8 A-operand reads
__device__ __forceinline__ void read_item(u32x4 (&fa)[8], u32x4 (&fb)[4], uint32_t a_addr, uint32_t b_addr) {
    fb[0] = lds_read16<0 * 2048>(b_addr);
    fb[1] = lds_read16<1 * 2048>(b_addr);
    fb[2] = lds_read16<2 * 2048>(b_addr);
    fb[3] = lds_read16<3 * 2048>(b_addr);
    fa[0] = lds_read16<0 * 2048>(a_addr);
    fa[1] = lds_read16<1 * 2048>(a_addr);
    fa[2] = lds_read16<2 * 2048>(a_addr);
    fa[3] = lds_read16<3 * 2048>(a_addr);
    fa[4] = lds_read16<4 * 2048>(a_addr);
    fa[5] = lds_read16<5 * 2048>(a_addr);
    fa[6] = lds_read16<6 * 2048>(a_addr);
    fa[7] = lds_read16<7 * 2048>(a_addr);
}

template <typename T>
__device__ __forceinline__ void mfma_item(f32x4 (&acc)[4][8], const u32x4 (&fa)[8], const u32x4 (&fb)[4]) {
    typedef typename Mfma<T>::frag frag;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
            acc[ni][mi] = Mfma<T>::run(__builtin_bit_cast(frag, fb[ni]), __builtin_bit_cast(frag, fa[mi]), acc[ni][mi]);
    __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ void wg_barrier() {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// DBG (developer decomposition builds only, mm_debug_gemm256): bit 0 = no global->LDS loads inside the K loop, bit 1 = no LDS
// fragment reads inside the loop, bit 2 = no epilogue stores, bit 3 = no MFMAs.  DBG = 0 everywhere in the product.
template <typename T, int EPI, bool RES, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_256_k(u16* __restrict__ C, u16* __restrict__ aux, const u16* __restrict__ A,
                                                        const u16* __restrict__ B, const u16* __restrict__ bias,
                                                        const u16* __restrict__ Rsd, int M, int N, int K, int lda, int ldb,
                                                        int ldc, int tiles_m, int tiles_n, int group_m, EpiExtra ex) {
    __shared__ __attribute__((aligned(16))) char smem[G::LDS];  // the ONLY LDS object of the kernel
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int tm, tn;
    tile_coords(blockIdx.x, tiles_m, tiles_n, group_m, tm, tn);
    const int bm = tm * G_BM, bn = tn * G_BN;
    const int wr = wave >> 2, wc = wave & 3;
    const int frow = lane & 15, fq = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;

    // ---- global -> LDS staging: instruction i of a tile covers rows i*64 + wave*8 + (lane>>3), LDS chunk lane&7
    uint32_t goffA[G::NLA], goffB[G::NLB];
    {
        const int rl = wave * 8 + (lane >> 3), cl = lane & 7;
#pragma unroll
        for (int i = 0; i < G::NLA; ++i) {
            const int r = i * 64 + rl;
            const int gr = min(bm + r, M - 1);  // edge tiles re-read the last row (never stored)
            goffA[i] = (uint32_t)gr * (uint32_t)(lda * 2) + (uint32_t)((cl ^ ((r >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < G::NLB; ++i) {
            const int r = i * 64 + rl;
            const int gr = min(bn + r, N - 1);
            goffB[i] = (uint32_t)gr * (uint32_t)(ldb * 2) + (uint32_t)((cl ^ ((r >> 1) & 7)) << 4);
        }
    }
    auto issue = [&](int t, int stage) {
        const char* ag = (const char*)A + (size_t)t * G_ROWB;
        const char* bg = (const char*)B + (size_t)t * G_ROWB;
        char* la = smem + stage * G::STAGE + wave * 8 * G_ROWB;  // wave-uniform; HW adds lane*16
        char* lb = la + G::A_BYTES;
#pragma unroll
        for (int i = 0; i < G::NLA; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(ag + goffA[i]), (lds_void*)(la + i * 64 * G_ROWB), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < G::NLB; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(bg + goffB[i]), (lds_void*)(lb + i * 64 * G_ROWB), 16, 0, 0);
    };

    // ---- fragment read addresses (bytes from the stage base): row*128 + ((k-chunk) ^ ((row>>1)&7))*16; the tile-row
    // increments of 16 rows do not change the swizzle term, so they are immediate offsets of the ds_read
    const uint32_t sw = (uint32_t)((frow >> 1) & 7);
    const uint32_t arow = (uint32_t)((wr * 128 + frow) * G_ROWB);
    const uint32_t brow = (uint32_t)(G::A_BYTES + (wc * 64 + frow) * G_ROWB);
    auto a_addr = [&](int stage, int kk) { return lds0 + stage * G::STAGE + arow + ((((uint32_t)(kk * 4 + fq)) ^ sw) << 4); };
    auto b_addr = [&](int stage, int kk) { return lds0 + stage * G::STAGE + brow + ((((uint32_t)(kk * 4 + fq)) ^ sw) << 4); };

    f32x4 acc[4][8];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = K / 64;
    u32x4 fa0[8], fb0[4], fa1[8], fb1[4];

    {
        // Two stages.  Invariant at the top of iteration t: tile t has landed and is visible to every wave; the reads
        // of item (t, k-half 0) are in flight into F0; the loads of tile t+1 (issued after barrier t-1) are in flight.
        issue(0, 0);
        if (nt > 1) {
            issue(1, 1);
            MM_WAIT(G::NL, 15);
        } else {
            MM_WAIT(0, 15);
        }
        wg_barrier();
        read_item(fa0, fb0, a_addr(0, 0), b_addr(0, 0));
        if (DBG & 2) read_item(fa1, fb1, a_addr(0, 1), b_addr(0, 1));
        for (int t = 0; t < nt; ++t) {
            const int st = t & 1;
            if (!(DBG & 2)) read_item(fa1, fb1, a_addr(st, 1), b_addr(st, 1));
            if (DBG & 2) MM_WAIT(63, 0); else
            MM_WAIT(63, 12);  // F0 complete (LDS returns in order; only the 12 reads just issued may be outstanding)
            __builtin_amdgcn_sched_barrier(0);
            if (!(DBG & 8)) mfma_item<T>(acc, fa0, fb0);
            // my share of tile t+1 has landed; all my reads of tile t (F1 included) are done
            MM_WAIT(0, 0);
            wg_barrier();  // => tile t+1 visible to everyone, stage st free for everyone
            if (!(DBG & 1)) if (t + 2 < nt) issue(t + 2, st);
            if (!(DBG & 2)) read_item(fa0, fb0, a_addr(st ^ 1, 0), b_addr(st ^ 1, 0));  // item (t+1, 0); harmless stale read after the last tile
            __builtin_amdgcn_sched_barrier(0);
            if (!(DBG & 8)) mfma_item<T>(acc, fa1, fb1);
        }
        MM_WAIT(0, 0);
        wg_barrier();  // every wave is done with the stages: LDS is reused by the epilogue
    }
    __builtin_amdgcn_sched_barrier(0);

    if (DBG & 4) {  // keep the accumulators (and, without MFMAs, the fragments) alive without storing anything real
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][3];
        sum += __uint_as_float(fa0[0][0] ^ fb0[0][0] ^ fa1[7][3] ^ fb1[3][3]);
        if (sum == 12345.678f) C[0] = (u16)lane;
        return;
    }
    // ---- epilogue (coalesced through a wave-private 16 KiB LDS region)
    epilogue_staged<T, EPI, RES, 8>(acc, smem + wave * 16384, bm + wr * 128, bn + wc * 64, M, N, ldc, C, aux, bias, Rsd, lane, ex);
}

template <typename T, int EPI>
static int launch256(void* C, void* aux, const void* A, const void* B, const void* bias, const void* residual, int M, int N, int K,
                     int lda, int ldb, int ldc, hipStream_t st, const EpiExtra& ex) {
    const int tiles_m = (M + G_BM - 1) / G_BM, tiles_n = (N + G_BN - 1) / G_BN;
    const int nwg = tiles_m * tiles_n;
    int group_m = (nwg / 8) / tiles_n;  // one XCD's run of tiles spans whole tile-rows when it can
    group_m = group_m < 1 ? 1 : (group_m > 8 ? 8 : group_m);
    dim3 grid(nwg), block(512);
    // (the rotary epilogue carries its tables in the aux / residual slots: never a residual add)
    if (EPI != MM_EPI_BIAS_ROTARY && (epi_is_bwd(EPI) || residual)) {
        if constexpr (EPI != MM_EPI_BIAS_ROTARY)
            gemm_nt_256_k<T, EPI, true><<<grid, block, 0, st>>>((u16*)C, (u16*)aux, (const u16*)A, (const u16*)B, (const u16*)bias,
                                                                (const u16*)residual, M, N, K, lda, ldb, ldc, tiles_m, tiles_n, group_m, ex);
    } else if constexpr (!epi_is_bwd(EPI)) {
        gemm_nt_256_k<T, EPI, false><<<grid, block, 0, st>>>((u16*)C, (u16*)aux, (const u16*)A, (const u16*)B, (const u16*)bias,
                                                             (const u16*)residual, M, N, K, lda, ldb, ldc, tiles_m, tiles_n, group_m, ex);
    }
    MM_CHECK_LAUNCH();
    return MM_OK;
}

template <typename T>
static int launch256_epi(void* C, void* aux, const void* A, const void* B, const void* bias, const void* residual, int M, int N,
                         int K, int lda, int ldb, int ldc, int epilogue, hipStream_t st, const EpiExtra& ex) {
    switch (epilogue) {
        case MM_EPI_BIAS_ROTARY: return launch256<T, MM_EPI_BIAS_ROTARY>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_NONE: return launch256<T, MM_EPI_NONE>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_BIAS: return launch256<T, MM_EPI_BIAS>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_BIAS_GELU_TANH: return launch256<T, MM_EPI_BIAS_GELU_TANH>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_BIAS_GELU_ERF: return launch256<T, MM_EPI_BIAS_GELU_ERF>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_BIAS_RELU: return launch256<T, MM_EPI_BIAS_RELU>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_RELU_BWD: return launch256<T, MM_EPI_RELU_BWD>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_GELU_TANH_BWD: return launch256<T, MM_EPI_GELU_TANH_BWD>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        default: return launch256<T, MM_EPI_GELU_ERF_BWD>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
    }
}

template <int EPI, int DBG>
static void launch256_dbg(void* C, void* aux, const void* A, const void* B, const void* bias, int M, int N, int K, hipStream_t st) {
    const int tiles_m = (M + G_BM - 1) / G_BM, tiles_n = (N + G_BN - 1) / G_BN;
    const int nwg = tiles_m * tiles_n;
    int group_m = (nwg / 8) / tiles_n;
    group_m = group_m < 1 ? 1 : (group_m > 8 ? 8 : group_m);
    gemm_nt_256_k<BF16, EPI, false, DBG><<<dim3(nwg), dim3(512), 0, st>>>((u16*)C, (u16*)aux, (const u16*)A, (const u16*)B, (const u16*)bias,
                                                                         nullptr, M, N, K, K, K, N, tiles_m, tiles_n, group_m, EpiExtra{});
}

// Developer decomposition of the 256x256 kernel (bf16, contiguous operands, M, N multiples of 256, K of 64): see DBG above.
int gemm256_debug(void* C, void* aux, const void* A, const void* B, const void* bias, int M, int N, int K, int gelu, int dbg,
                  hipStream_t st) {
    if (M % 256 || N % 256 || K % 64 || K < 128) return MM_EINVAL;
#define MM_DBG_CASE(D)                                                                         \
    case D:                                                                                     \
        if (gelu) launch256_dbg<MM_EPI_BIAS_GELU_TANH, D>(C, aux, A, B, bias, M, N, K, st);    \
        else launch256_dbg<MM_EPI_NONE, D>(C, aux, A, B, bias, M, N, K, st);                   \
        break;
    switch (dbg) {
        MM_DBG_CASE(0) MM_DBG_CASE(1) MM_DBG_CASE(2) MM_DBG_CASE(3) MM_DBG_CASE(4) MM_DBG_CASE(5) MM_DBG_CASE(6) MM_DBG_CASE(7)
        MM_DBG_CASE(8) MM_DBG_CASE(12) MM_DBG_CASE(14) MM_DBG_CASE(13)
        default: return MM_EINVAL;
    }
#undef MM_DBG_CASE
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ELAUNCH;
}

// Shape policy: the 256x256 tile pays when its workgroups fill the 256 CUs in whole rounds (or in so many rounds that the
// partial last one does not matter); otherwise the 128x128 kernel with ~3 workgroups per CU balances better.  Measured at
// M = 8192 (TFLOP/s, this kernel vs gemm.hip): N=4096 K=1024 1058 vs 879, N=50432 K=1024 999 vs 810, N=3072 (1.5 rounds) 791 vs 807,
// N=1024 (half a round) 690 vs 990.
bool gemm256_takes(int M, int N, int K, int lda, int ldb) {
    int force = 0;
    if (const char* e = getenv("MM_GEMM256")) force = atoi(e);  // developer switch: 1 = always (when legal), -1 = never
    if (force < 0 || M < 256 || N < 256 || K < 128) return false;
    // staging uses 32-bit byte offsets from the operand base
    if ((size_t)M * (size_t)lda * 2 >= (1ull << 32) || (size_t)N * (size_t)ldb * 2 >= (1ull << 32)) return false;
    const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
    if (force == 0 && !(tiles % 256 == 0 || tiles >= 1024)) return false;
    return true;
}

int gemm256_try(void* C, void* aux, const void* A, const void* B, const void* bias, const void* residual, int M, int N, int K,
                int lda, int ldb, int ldc, int epilogue, int dtype, hipStream_t st, const EpiExtra& ex) {
    if (!gemm256_takes(M, N, K, lda, ldb)) return 1;
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        return launch256_epi<T>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, epilogue, st, ex);
    });
}

}  // namespace mm
