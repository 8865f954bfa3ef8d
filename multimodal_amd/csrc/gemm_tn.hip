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
#include <mutex>

namespace mm {

#ifndef MM_AB_TN_PRE
#define MM_AB_TN_PRE 4
#endif
constexpr int TN_PRE = MM_AB_TN_PRE;  // register sets of tiles in flight (even)
static_assert(TN_PRE % 2 == 0, "set parity = LDS buffer parity");
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

// One weight-gradient problem.  Two of them can share a launch (mm_gemm_tn_pair): a linear layer pair such as the
// adapter's has two small weight gradients over the same tokens, and one grid of 256 workgroups covers both.
struct TnProb {
    u16* C;          // [N, Kp] output (ldc)
    u16* colsum;     // [N] column sums of A, or null
    const u16* A;    // [M, N]  (lda)
    const u16* B;    // [M, Kp] (ldb)
    int N, Kp, lda, ldb, ldc, tiles_k, tiles;
    size_t ws_off;   // offset of this problem's part inside one split's workspace slab (floats)
};

// SLAB: fp32 partials to ws [gridDim.y][slab] (slab = both problems' N*Kp (+ N column sums)); otherwise direct 16-bit output.
template <typename T, bool SLAB>
__global__ __launch_bounds__(256) void gemm_tn_k(TnProb p0, TnProb p1, float* __restrict__ ws, size_t slab, int rows_per_split) {
    typedef Tile<128> TL;
    typedef typename M32<T>::frag frag;
    __shared__ __attribute__((aligned(16))) char smem[4 * TL::BYTES];  // [buf][A | B]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    // Workgroups are dealt to the 8 XCDs round-robin in launch order, and each XCD has its own L2.  The adapter problems are
    // one tile wide in N or in Kp, so all 8 tiles of a problem in one token slice read the SAME narrow operand slice
    // (dpre / h1): launch order would put them on 8 different XCDs and fetch that slice 8 times.  Within every block of 64
    // launch ids, XCD x takes work ids 8x .. 8x+7 (tile fastest), i.e. one (problem, slice) group per XCD at a time.
    const unsigned lid = blockIdx.x + gridDim.x * blockIdx.y, total = gridDim.x * gridDim.y;
    const unsigned wid = lid < (total & ~63u) ? (lid & ~63u) + (lid & 7u) * 8u + ((lid >> 3) & 7u) : lid;
    const int bx = (int)(wid % gridDim.x), by = (int)(wid / gridDim.x);
    const bool second = bx >= p0.tiles;  // workgroup-uniform
    const TnProb& P = second ? p1 : p0;
    const int tile = second ? bx - p0.tiles : bx;
    const u16* __restrict__ A = P.A;
    const u16* __restrict__ B = P.B;
    const int N = P.N, Kp = P.Kp, lda = P.lda, ldb = P.ldb, ldc = P.ldc, tiles_k = P.tiles_k;
    const bool CS = P.colsum != nullptr;
    const int tn = tile / tiles_k, tk = tile % tiles_k;
    const int n0 = tn * TN_BT, k0 = tk * TN_BT;
    const int m_begin = by * rows_per_split;
    const int ntile = rows_per_split / 64;
    const int wr = wave >> 1, wc = wave & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const bool do_cs = CS && tk == 0 && wc == 0;  // wave-uniform (CS: this problem wants column sums)
    f32x16 cs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) cs[i][e] = 0.f;
    const frag ones = Ones<T>::get();

    // Tiles come from HBM (dY, X: 16 MiB each) and a 64-token step is only 16 MFMAs per wave, so the loop is bound by how many
    // bytes are in flight: TN_PRE - 1 tiles are prefetched in registers beyond the one being multiplied (32 KiB each; one
    // workgroup per CU has the registers for it).  Register set (t % TN_PRE) holds tile t from its load until it is written to
    // LDS buffer (t & 1) at the end of iteration t-1.
    u32x4 areg[TN_PRE][4], breg[TN_PRE][4];
#pragma unroll
    for (int d = 0; d < TN_PRE; ++d)
        if (d < ntile) {
            tn_tile_load(areg[d], A, lda, m_begin + d * 64, n0, tid);
            tn_tile_load(breg[d], B, ldb, m_begin + d * 64, k0, tid);
        }
    tile_store<128>(smem, areg[0], tid);
    tile_store<128>(smem + TL::BYTES, breg[0], tid);
    __syncthreads();

    // flags: 0 = decide at run time (guards), 1 = always, -1 = never.  The guarded form is only for token counts that are not a
    // multiple of TN_PRE tiles: behind a conditional load the compiler's s_waitcnt pass no longer knows how many loads are
    // outstanding and waits for vmcnt(0) before every LDS store (what this kernel did until round 3's third set, with two
    // register sets).  Measured, tools/ab_tn_pair.py: the straight-line form with four sets and counted waits takes the adapter
    // pair from 23.8 to 23.2 us - the launch is one round of 256 workgroups whose first fetch, 256 KiB stream and 64 KiB slab
    // store run one after the other, and the per-step wait was the smallest of the three.
    auto body = [&](int t, auto phase, auto load_f, auto store_f) {
        constexpr int set = decltype(phase)::value;  // == t % TN_PRE
        constexpr int cur = set & 1;                 // == t & 1 (TN_PRE is even)
        constexpr int nset = (set + 1) % TN_PRE;
        constexpr int LOAD = decltype(load_f)::value, STORE = decltype(store_f)::value;
        const char* as = smem + cur * 2 * TL::BYTES;
        const char* bs = as + TL::BYTES;
        if (LOAD == 1 || (LOAD == 0 && t + TN_PRE < ntile)) {  // set `set` is free: tile t went to LDS at the end of the previous iteration
            tn_tile_load(areg[set], A, lda, m_begin + (t + TN_PRE) * 64, n0, tid);
            tn_tile_load(breg[set], B, ldb, m_begin + (t + TN_PRE) * 64, k0, tid);
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
        if (STORE == 1 || (STORE == 0 && t + 1 < ntile)) {
            char* nx = smem + (cur ^ 1) * 2 * TL::BYTES;
            tile_store<128>(nx, areg[nset], tid);
            tile_store<128>(nx + TL::BYTES, breg[nset], tid);
        }
        __syncthreads();
    };
    using std::integral_constant;
    if (ntile % TN_PRE == 0) {
        // straight-line groups of TN_PRE tiles: every wait in them is a counted vmcnt
        int t = 0;
        for (; t + TN_PRE < ntile; t += TN_PRE)
            static_for<0, TN_PRE>([&](auto d) { body(t + d.value, d, integral_constant<int, 1>{}, integral_constant<int, 1>{}); });
        static_for<0, TN_PRE>([&](auto d) {
            body(t + d.value, d, integral_constant<int, -1>{}, integral_constant<int, (d.value + 1 < TN_PRE) ? 1 : -1>{});
        });
    } else {
        for (int t = 0; t < ntile; t += TN_PRE)
            static_for<0, TN_PRE>([&](auto d) {
                if (t + d.value < ntile) body(t + d.value, d, integral_constant<int, 0>{}, integral_constant<int, 0>{});
            });
    }
    // D[row = n (registers)][col = kp (lane)]
    float* wsp = ws + (size_t)by * slab + P.ws_off;
    u16* C = P.C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kp = k0 + wc * 64 + 32 * j + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wr * 64 + 32 * i + acc_row(e, hh);
                if (SLAB) wsp[(size_t)n * Kp + kp] = acc[i][j][e];
                else C[(size_t)n * ldc + kp] = from_f32<T>(acc[i][j][e]);
            }
        }
    if (do_cs && r == 0) {  // every column of cs holds the sums: lanes 0 and 32 own all 32 rows between them
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wr * 64 + 32 * i + acc_row(e, hh);
                if (SLAB) wsp[(size_t)N * Kp + n] = cs[i][e];
                else P.colsum[n] = from_f32<T>(cs[i][e]);
            }
    }
}

// out = sum_s ws[s][.] (fixed order: k ascending), 2 elements per thread (twice the waves of a 4-element version: the
// slab pair is only ~1 MiB wide, so the kernel lives on loads in flight); a slab is [p0: N*Kp, N column sums | p1: the same]
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_k(TnProb p0, TnProb p1, const float* __restrict__ ws, size_t slab, int splits) {
    const size_t i2 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i2 >= slab) return;
    const float* w = ws + i2;
    auto ld = [&](int k) { return *reinterpret_cast<const f32x2*>(w + (size_t)k * slab); };
    f32x2 s = ld(0);
    int k = 1;
    if (splits >= 8) {  // eight independent loads in flight, then the adds in order
        f32x2 v[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) v[u] = ld(1 + u);
#pragma unroll
        for (int u = 0; u < 7; ++u) s += v[u];
        k = 8;
    }
    for (; k + 8 <= splits; k += 8) {
        f32x2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld(k + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < splits; ++k) s += ld(k);
    const bool second = p1.tiles > 0 && i2 >= p1.ws_off;
    const TnProb& P = second ? p1 : p0;
    const size_t j = i2 - P.ws_off, nk = (size_t)P.N * P.Kp;
    u16* o;
    if (j < nk) o = P.C + (j / P.Kp) * P.ldc + (j % P.Kp);
    else o = P.colsum + (j - nk);
    *reinterpret_cast<uint32_t*>(o) = pack2<T>(s[0], s[1]);
}

static int tn_check(const char* fn, const void* C, const void* cs, const void* A, const void* B, int M, int N, int Kp, int lda, int ldb,
                    int ldc, int splits) {
    MM_CHECK_ARG(C && A && B, "%s: null pointer", fn);
    MM_CHECK_ARG(M > 0 && N > 0 && Kp > 0, "%s: M=%d N=%d Kp=%d must be positive", fn, M, N, Kp);
    MM_CHECK_ARG(N % TN_BT == 0 && Kp % TN_BT == 0, "%s: N=%d and Kp=%d must be multiples of %d", fn, N, Kp, TN_BT);
    MM_CHECK_ARG(splits >= 1 && splits <= 65535 && M % (64 * splits) == 0, "%s: M=%d must be a multiple of 64*splits (splits=%d)", fn, M, splits);
    MM_CHECK_ARG(lda >= N && ldb >= Kp && ldc >= Kp && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0,
                 "%s: bad leading dimensions lda=%d ldb=%d ldc=%d", fn, lda, ldb, ldc);
    MM_CHECK_ARG(aligned16(A) && aligned16(B) && (((uintptr_t)C) & 7) == 0 && (((uintptr_t)cs) & 7) == 0,
                 "%s: A/B must be 16-byte, C/a_colsum 8-byte aligned", fn);
    return MM_OK;
}

static TnProb tn_prob(void* C, void* cs, const void* A, const void* B, int N, int Kp, int lda, int ldb, int ldc, size_t ws_off) {
    TnProb p;
    p.C = (u16*)C; p.colsum = (u16*)cs; p.A = (const u16*)A; p.B = (const u16*)B;
    p.N = N; p.Kp = Kp; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.tiles_k = Kp / TN_BT; p.tiles = (N / TN_BT) * p.tiles_k; p.ws_off = ws_off;
    return p;
}

template <typename T>
static int tn_launch(const TnProb& p0, const TnProb& p1, float* ws, size_t slab, int M, int splits, hipStream_t st) {
    dim3 grid(p0.tiles + p1.tiles, splits), block(256);
    if (splits == 1) {
        gemm_tn_k<T, false><<<grid, block, 0, st>>>(p0, p1, nullptr, 0, M);
        MM_CHECK_LAUNCH();
    } else {
        gemm_tn_k<T, true><<<grid, block, 0, st>>>(p0, p1, ws, slab, M / splits);
        MM_CHECK_LAUNCH();
        splitk_reduce_k<T><<<(unsigned)((slab / 2 + 255) / 256), 256, 0, st>>>(p0, p1, ws, slab, splits);
        MM_CHECK_LAUNCH();
    }
    return MM_OK;
}

}  // namespace mm

using namespace mm;

static int tn_splits_for(int M, int tiles) {
    int splits = 1;
    while (tiles * splits * 2 <= 256 && M % (64 * splits * 2) == 0 && M / (splits * 2) >= 128) splits *= 2;
    return splits;
}

extern "C" int mm_gemm_tn_splits(int M, int N, int Kp) {
    if (M <= 0 || N <= 0 || Kp <= 0 || N % TN_BT || Kp % TN_BT || M % 64) return 0;  // 0: shape not supported by this kernel
    return tn_splits_for(M, (N / TN_BT) * (Kp / TN_BT));
}

extern "C" int mm_gemm_tn_pair_splits(int M, int N0, int Kp0, int N1, int Kp1) {
    if (M <= 0 || N0 <= 0 || Kp0 <= 0 || N1 <= 0 || Kp1 <= 0 || N0 % TN_BT || Kp0 % TN_BT || N1 % TN_BT || Kp1 % TN_BT || M % 64) return 0;
    return tn_splits_for(M, (N0 / TN_BT) * (Kp0 / TN_BT) + (N1 / TN_BT) * (Kp1 / TN_BT));
}

extern "C" int mm_gemm_tn(void* C, void* a_colsum, float* workspace, const void* A, const void* B, int M, int N, int Kp, int lda,
                          int ldb, int ldc, int splits, int dtype, void* stream) {
    if (int rc = tn_check("mm_gemm_tn", C, a_colsum, A, B, M, N, Kp, lda, ldb, ldc, splits)) return rc;
    MM_CHECK_ARG(splits == 1 || (workspace && aligned16(workspace)),
                 "mm_gemm_tn: splits > 1 needs a 16-byte aligned fp32 workspace [splits, N*Kp + N]");
    const TnProb p0 = tn_prob(C, a_colsum, A, B, N, Kp, lda, ldb, ldc, 0);
    TnProb p1 = p0;
    p1.tiles = 0;
    const size_t slab = (size_t)N * Kp + (a_colsum ? N : 0);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        return tn_launch<T>(p0, p1, workspace, slab, M, splits, (hipStream_t)stream);
    });
}

extern "C" int mm_gemm_tn_pair(void* C0, void* a_colsum0, const void* A0, const void* B0, int N0, int Kp0, void* C1,
                               void* a_colsum1, const void* A1, const void* B1, int N1, int Kp1, int M, float* workspace,
                               int splits, int dtype, void* stream) {
    if (int rc = tn_check("mm_gemm_tn_pair", C0, a_colsum0, A0, B0, M, N0, Kp0, N0, Kp0, Kp0, splits)) return rc;
    if (int rc = tn_check("mm_gemm_tn_pair", C1, a_colsum1, A1, B1, M, N1, Kp1, N1, Kp1, Kp1, splits)) return rc;
    MM_CHECK_ARG(splits == 1 || (workspace && aligned16(workspace)),
                 "mm_gemm_tn_pair: splits > 1 needs a 16-byte aligned fp32 workspace [splits, N0*Kp0 + N0 + N1*Kp1 + N1]");
    const size_t slab0 = (size_t)N0 * Kp0 + (a_colsum0 ? N0 : 0);
    const size_t slab = slab0 + (size_t)N1 * Kp1 + (a_colsum1 ? N1 : 0);
    const TnProb p0 = tn_prob(C0, a_colsum0, A0, B0, N0, Kp0, N0, Kp0, Kp0, 0);
    const TnProb p1 = tn_prob(C1, a_colsum1, A1, B1, N1, Kp1, N1, Kp1, Kp1, slab0);
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        return tn_launch<T>(p0, p1, workspace, slab, M, splits, (hipStream_t)stream);
    });
}

// ---------------------------------------------------------------- the entry point that allocates nothing
// mm_gemm_tn with its split-K slabs in a file-scope device array instead of a caller workspace, for callers that cannot
// allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc).  `scratch` below is a host-side stand-in
// for the symbol, not a device address: host code only ever uses the per-device pointer hipGetSymbolAddress returned.
constexpr size_t TN_SCRATCH_FLOATS = MM_GEMM_TN_STATIC_MAX_FLOATS;  // 24 MiB per GPU, held until the process ends
static __device__ __align__(256) float scratch[TN_SCRATCH_FLOATS];

constexpr int TN_MAX_DEVICES = 64;
static float* g_scratch_ptr[TN_MAX_DEVICES];  // each GPU holds its own copy of the array at its own address
static std::mutex g_scratch_mu;

// per-device address of `scratch`, resolved on the first call on that device (with that device current)
static int tn_scratch_for_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= TN_MAX_DEVICES) {
        mm::set_error("mm_gemm_tn_static: hipGetDevice failed or device index %d out of range: %s", dev, hipGetErrorString(e));
        return MM_ELAUNCH;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        float* ptr = nullptr;
        e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            mm::set_error("mm_gemm_tn_static: hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
            return MM_ELAUNCH;
        }
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return MM_OK;
}

extern "C" int mm_gemm_tn_static(void* C, void* a_colsum, const void* A, const void* B, int M, int N, int Kp, int lda, int ldb,
                                 int ldc, int splits, int dtype, void* stream) {
    if (int rc = tn_check("mm_gemm_tn_static", C, a_colsum, A, B, M, N, Kp, lda, ldb, ldc, splits)) return rc;
    const size_t slab = (size_t)N * Kp + (a_colsum ? N : 0);
    MM_CHECK_ARG(splits == 1 || slab * (size_t)splits <= TN_SCRATCH_FLOATS,
                 "mm_gemm_tn_static: splits * (N*Kp + N) = %zu fp32 exceeds the built-in scratch of %zu fp32 (24 MiB): lower "
                 "`splits` or call mm_gemm_tn with a workspace", slab * (size_t)splits, TN_SCRATCH_FLOATS);
    float* ws = nullptr;
    if (splits > 1)  // every slab word the reduction reads is written by this call's first kernel: nothing to clear
        if (int rc = tn_scratch_for_current_device(&ws)) return rc;
    const TnProb p0 = tn_prob(C, a_colsum, A, B, N, Kp, lda, ldb, ldc, 0);
    TnProb p1 = p0;
    p1.tiles = 0;
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        return tn_launch<T>(p0, p1, ws, slab, M, splits, (hipStream_t)stream);
    });
}
