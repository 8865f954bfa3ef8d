// adapter.hip - the MAGMA bottleneck adapter as ONE kernel per direction (gfx950).
//
// Reference: megatron/model/adapter.py:18-25,109-117   out = o + W2 relu(W1 o + b1) + b2   (o = the wrapped block's
// pre-bias output [tokens, h], W1 [r, h], W2 [h, r], r = h / downsample_factor).  The adapter is 2.5 % of the layer's
// flops but, run as two skinny GEMMs per direction (N = r = 128 gives 64 workgroups; 16 dependent K-steps each), it was
// 10 % of the step: latency-bound, not bandwidth- or MFMA-bound.  Both directions are the same computation,
//
//     H = f1(X Wa^T)  [tokens, r]          Y = f2(H Wb^T) + X  [tokens, h]
//
//   forward :  X = o,  Wa = W1, f1 = relu(. + b1),          Wb = W2,   f2 = . + b2      (H = the saved ReLU output)
//   backward:  X = dY, Wa = W2^T, f1 = mask by saved H > 0, Wb = W1^T, f2 = identity    (H = d(pre-activation), Y = dX)
//              (the kernel takes W2 and W1 as stored: their tiles are staged row-major and the MFMA fragments are read
//               TRANSPOSED from LDS with ds_read_b64_tr_b16 - no per-step W^T copies of the trainable weights)
//
// so one kernel template serves both.  Structure: a workgroup owns 32 token rows (256 workgroups at 8192 tokens);
//   * eight waves (two per SIMD): wave (w & 3, w >> 2) owns a quarter of the columns and one of the two 16-row MFMA tiles;
//   * the 32 x h block of X is staged into LDS (XOR-swizzled 16-byte chunks) in 64-column chunks that travel with the
//     weight tiles of the first product - it is the MFMA operand of the first product AND the residual of the second; the
//     output block is assembled in place and each finished 64-column chunk is stored while the next one is computed;
//   * the weights (2 x 256 KiB, L2-resident, shared by every workgroup) stream through a three-buffer LDS ring in 32 steps
//     (16 K-tiles of Wa, 16 column chunks of Wb), fetched with COALESCED 16-byte loads into registers several steps ahead
//     (fully unrolled, so the prefetch distance is just register liveness) - fragment-shaped loads straight from global
//     memory were tried first and are 8x more L1 accesses per instruction (one line per lane): 26 us, L1-issue-bound;
//   * H goes through an 8 KiB LDS tile between the two products (and to HBM once, coalesced: the backward needs it);
//   * one barrier per step; within a step the LDS write of tile s+2 and the fragment reads of tile s+1 overlap the
//     MFMAs of tile s.  Measured (s_memtime stamps, -DMM_AD_STAMP diagnostic build): prologue 4.8k cycles, 16 steps of the
//     first product 10.8k, 16 chunks of the second 12.6k - ~700 cycles per step against a 470-cycle L2 -> CU fill bound
//     (16 KiB per step): the LDS pipe (ds_write of the tile + fragment reads, ~400 cycles per step) is the next limit.
// Rounding points are exactly those of the unfused path (mm_gemm_nt with MM_EPI_BIAS_RELU / MM_EPI_BIAS + residual /
// MM_EPI_RELU_BWD / residual), and each output element's K-order is the same, so the results are bit-identical to it.
#include "gemm_common.h"
#include "mfma32.h"

namespace mm {

constexpr int AD_BM = 32;
constexpr int AD_THREADS = 512;  // 8 waves: two per SIMD, so one wave's LDS / L2 latency hides behind its partner's MFMAs

template <typename T, int HID, int R, int MODE>  // MODE 0: forward, 1: backward (data gradient)
__global__ __launch_bounds__(AD_THREADS) void bottleneck_k(u16* __restrict__ Y, u16* __restrict__ Hout, const u16* __restrict__ X,
                                                    const u16* __restrict__ Wa, const u16* __restrict__ ba,
                                                    const u16* __restrict__ Wb, const u16* __restrict__ bb,
                                                    const u16* __restrict__ Hsaved, int rows) {
    typedef typename Mfma<T>::frag frag;
    constexpr int XROW = HID * 2, HROW = R * 2;     // bytes per LDS row of the X block / the H tile
    constexpr int XCPR = HID / 8, HCPR = R / 8;     // 16-byte chunks per row
    constexpr int HSW = HCPR >= 16 ? 15 : HCPR - 1; // swizzle mask for rows of HCPR chunks
    constexpr int NI1 = R / 64;                     // first product: wave (w & 3, w >> 2) owns R/4 columns = NI1 MFMA tiles of 16 rows
    constexpr int S1 = HID / 64;                    // ... in S1 steps of 64 k (one Wa tile [R][64] each)
    constexpr int CH2 = 64;                         // second product: chunks of CH2 output columns (Wb rows), 16 per wave
    constexpr int S2 = HID / CH2;
    constexpr int NI2 = CH2 / 64;                   // MFMA tiles per wave per chunk
    constexpr int KS2 = R / 32;
    constexpr int L1N = R * 8 / AD_THREADS;         // 16-byte loads per thread per Wa tile
    constexpr int L2N = CH2 * HCPR / AD_THREADS;    // ... per Wb chunk
    constexpr int LMAX = L1N > L2N ? L1N : L2N;
    constexpr int RING = (R * 128 > CH2 * HROW) ? R * 128 : CH2 * HROW;  // bytes per ring buffer
    constexpr int D1 = 8;                           // prefetch distance in steps (Wa tiles and Wb chunks alike): ~128 KiB in flight per CU
    __shared__ __attribute__((aligned(16))) char smem[AD_BM * XROW + AD_BM * HROW + 3 * RING + HID * 2];
    char* Xs = smem;
    char* Hs = smem + AD_BM * XROW;
    char* Ws = Hs + AD_BM * HROW;
    char* Bs = Ws + 3 * RING;  // bb staged once (forward)
    const int tid = threadIdx.x, wave = (tid >> 6) & 3, wmi = tid >> 8, lane = tid & 63, frow = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * AD_BM;
#ifdef MM_AD_STAMP
    const uint64_t st0 = __builtin_amdgcn_s_memtime();
    uint64_t st1 = 0, st2 = 0;
#endif

    // ---- weight staging: step s < S1 is Wa tile s, step S1 + c is Wb chunk c
    // Forward: Wa = W1 [R][HID] and Wb = W2 [HID][R] are K-contiguous: tiles [n][k], fragments by ds_read_b128.
    // Backward: the operands are W2^T and W1^T of the weights as stored, so a tile is staged as [k][n] (rows of the stored
    // matrix) and fragments are read transposed; 32-byte segments are XOR-swizzled so that the 8 rows one half-wave
    // touches per ds_read_b64_tr_b16 fall in different banks.
    constexpr int T1SEG = HROW / 32;  // backward Wa tile: [64 k][R n], row = HROW bytes
    auto tsw = [](int row, int nseg) { return nseg >= 8 ? ((row & 3) | (((row >> 3) & 1) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 1) << 1)); };
    u32x4 wreg[S1 + S2][LMAX];
    auto w_issue = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s < S1) {
#pragma unroll
            for (int i = 0; i < L1N; ++i) {
                const int c = tid + AD_THREADS * i;
                if (MODE == 0) {
                    const int row = c >> 3, ch = c & 7;
                    wreg[s][i] = *reinterpret_cast<const u32x4*>(Wa + (size_t)row * HID + s * 64 + ch * 8);
                } else {  // Wa = stored W2 [HID][R]: rows 64s .. 64s+63
                    const int row = c / HCPR, ch = c % HCPR;
                    wreg[s][i] = *reinterpret_cast<const u32x4*>(Wa + (size_t)(s * 64 + row) * R + ch * 8);
                }
            }
        } else if constexpr (s < S1 + S2) {
#pragma unroll
            for (int i = 0; i < L2N; ++i) {
                const int c = tid + AD_THREADS * i;
                if (MODE == 0) {
                    const int row = c / HCPR, ch = c % HCPR;
                    wreg[s][i] = *reinterpret_cast<const u32x4*>(Wb + (size_t)((s - S1) * CH2 + row) * R + ch * 8);
                } else {  // Wb = stored W1 [R][HID]: all R rows, columns CH2*c .. +CH2-1
                    const int row = c >> 3, ch = c & 7;
                    wreg[s][i] = *reinterpret_cast<const u32x4*>(Wb + (size_t)row * HID + (s - S1) * CH2 + ch * 8);
                }
            }
        }
    };
    auto w_store = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        char* buf = Ws + (s % 3) * RING;
        if constexpr (s >= S1 + S2) {
        } else if constexpr (s < S1) {
#pragma unroll
            for (int i = 0; i < L1N; ++i) {
                const int c = tid + AD_THREADS * i;
                if (MODE == 0) {
                    const int row = c >> 3, ch = c & 7;
                    *reinterpret_cast<u32x4*>(buf + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = wreg[s][i];
                } else {
                    const int row = c / HCPR, ch = c % HCPR;
                    *reinterpret_cast<u32x4*>(buf + row * HROW + (((ch >> 1) ^ tsw(row, T1SEG)) << 5) + ((ch & 1) << 4)) = wreg[s][i];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < L2N; ++i) {
                const int c = tid + AD_THREADS * i;
                if (MODE == 0) {
                    const int row = c / HCPR, ch = c % HCPR;
                    *reinterpret_cast<u32x4*>(buf + row * HROW + ((ch ^ (row & HSW)) << 4)) = wreg[s][i];
                } else {  // [R k][CH2 n], rows of 128 bytes (4 segments)
                    const int row = c >> 3, ch = c & 7;
                    *reinterpret_cast<u32x4*>(buf + row * 128 + (((ch >> 1) ^ tsw(row, 4)) << 5) + ((ch & 1) << 4)) = wreg[s][i];
                }
            }
        }
    };
    // transposed fragment of a [k][n] tile (row bytes rb, nseg 32-byte segments per row): lane (i = lane&15, g = lane>>4)
    // gets tile[kb + 8g + j][n0 + i], j = 0..7 (lane 4q+p of a 16-lane group supplies the address of row q, columns 4p..4p+3)
    auto tr_frag16 = [&](const char* buf, int rb, int nseg, int kb, int n0) {
        const int q = frow >> 2, pp = frow & 3;
        const int r0 = kb + 8 * fq + q, r1 = r0 + 4;
        const char* pa = buf + r0 * rb + (((n0 >> 4) ^ tsw(r0, nseg)) << 5) + 8 * pp;
        const char* pb = buf + r1 * rb + (((n0 >> 4) ^ tsw(r1, nseg)) << 5) + 8 * pp;
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
        const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
        const s16x8 v = s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        frag f;
        __builtin_memcpy(&f, &v, 16);
        return f;
    };

    // ---- small operands first: a load that is issued late is waited for with vmcnt(0), i.e. it drains the whole weight
    // prefetch queue behind which it sits (vmcnt retires in order) - so nothing but weight tiles is loaded after this point.
    const int nb1 = wave * (R / 4);
    const int mrow = wmi * 16 + frow;  // the block row this lane owns in both products (wave half wmi = MFMA row tile)
    u32x2 small1[NI1];     // forward: b1 of the lane's H columns; backward: the saved H values of (mrow, those columns)
    u32x4 bbreg = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int ni = 0; ni < NI1; ++ni) {
        const int n0 = nb1 + ni * 16 + fq * 4;
        if (MODE == 0) {
            small1[ni] = *reinterpret_cast<const u32x2*>(ba + n0);
        } else {
            small1[ni] = *reinterpret_cast<const u32x2*>(Hsaved + (size_t)min(m0 + mrow, rows - 1) * R + n0);
        }
    }
    if (MODE == 0 && tid < XCPR) bbreg = *reinterpret_cast<const u32x4*>(bb + tid * 8);

    // ---- X block: staged per 64-column chunk, chunk s travelling with weight tile s (issued D1 steps ahead, written to
    // LDS two steps ahead, published by the same barrier): the first product starts when the FIRST chunk has arrived, not
    // the whole 64 KiB block.  Thread -> (row tid>>3, 16-byte piece tid&7 of the chunk): 8 lanes cover one 128-byte line.
    // Waves 0-3 move the X chunks in and the output chunks out (256 pieces per chunk); waves 4-7 only move weights.
    static_assert(AD_BM * 8 == 256, "one 16-byte piece per thread of the first four waves per X chunk");
    u32x4 xr[S1];
    const bool xmover = tid < 256;  // wave-uniform
    const int xrow = (tid >> 3) & 31, xpc = tid & 7;
    const u16* xsrc = X + (size_t)min(m0 + xrow, rows - 1) * HID + xpc * 8;
    auto x_issue = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s < S1)
            if (xmover) xr[s] = *reinterpret_cast<const u32x4*>(xsrc + s * 64);
    };
    auto x_store = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s < S1)
            if (xmover) *reinterpret_cast<u32x4*>(Xs + xrow * XROW + (((s * 8 + xpc) ^ (xrow & 15)) << 4)) = xr[s];
    };
    static_for<0, D1>([&](auto sc) {
        x_issue(sc);
        w_issue(sc);
    });
    if (MODE == 0 && tid < XCPR) *reinterpret_cast<u32x4*>(Bs + tid * 16) = bbreg;

    // ---- first product: H[32, R], wave w owns columns [w*R/4, (w+1)*R/4)
    // Ring protocol (3 buffers): during step s the wave (1) writes the tile of step s+2, (2) reads the FRAGMENTS of step
    // s+1 into registers, (3) runs the MFMAs of step s on fragments it read one step earlier - with one wave per SIMD
    // nothing else hides the LDS latency - then one barrier: it publishes the write of (1) and retires the reads of (2),
    // whose buffer is rewritten in step s+2.
    f32x4 acc1[NI1];
#pragma unroll
    for (int i = 0; i < NI1; ++i) acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag FA[S1][2], FB[S1][2][NI1], FB2[S2][KS2][NI2];
    auto frags1 = [&](auto sc) {  // fragments of first-product step s
        constexpr int s = decltype(sc)::value;
        if constexpr (s < S1) {
            const char* buf = Ws + (s % 3) * RING;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                FA[s][kk] = *reinterpret_cast<const frag*>(Xs + mrow * XROW + (((s * 8 + kk * 4 + fq) ^ frow) << 4));
#pragma unroll
                for (int ni = 0; ni < NI1; ++ni) {
                    if (MODE == 0) {
                        const int n = nb1 + ni * 16 + frow;
                        FB[s][kk][ni] = *reinterpret_cast<const frag*>(buf + n * 128 + (((kk * 4 + fq) ^ ((n >> 1) & 7)) << 4));
                    } else {
                        FB[s][kk][ni] = tr_frag16(buf, HROW, T1SEG, kk * 32, nb1 + ni * 16);
                    }
                }
            }
        }
    };
    auto frags2 = [&](auto cc) {  // weight fragments of second-product chunk c
        constexpr int c = decltype(cc)::value;
        if constexpr (c < S2) {
            const char* buf = Ws + ((S1 + c) % 3) * RING;
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
                for (int ni = 0; ni < NI2; ++ni) {
                    if (MODE == 0) {
                        const int n = wave * (CH2 / 4) + ni * 16 + frow;  // row of the chunk
                        FB2[c][ks][ni] = *reinterpret_cast<const frag*>(buf + n * HROW + (((ks * 4 + fq) ^ (n & HSW)) << 4));
                    } else {
                        FB2[c][ks][ni] = tr_frag16(buf, 128, 4, ks * 32, wave * (CH2 / 4) + ni * 16);
                    }
                }
        }
    };
    x_store(std::integral_constant<int, 0>{});
    w_store(std::integral_constant<int, 0>{});
    x_store(std::integral_constant<int, 1>{});
    w_store(std::integral_constant<int, 1>{});
    __syncthreads();
#ifdef MM_AD_STAMP
    st1 = __builtin_amdgcn_s_memtime();
#endif
    frags1(std::integral_constant<int, 0>{});
    static_for<0, S1>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        x_store(std::integral_constant<int, s + 2>{});
        w_store(std::integral_constant<int, s + 2>{});
        static_for<s + D1, s + D1 + 1>([&](auto nx) {  // (a no-op past the last step)
            x_issue(nx);
            w_issue(nx);
        });
        frags1(std::integral_constant<int, s + 1>{});
        if constexpr (s + 1 == S1) frags2(std::integral_constant<int, 0>{});
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < NI1; ++ni) acc1[ni] = Mfma<T>::run(FB[s][kk][ni], FA[s][kk], acc1[ni]);
        if constexpr (s + 1 < S1) __syncthreads();  // (the last step's barrier comes after the H tile is written)
    });
    // lane holds H[mrow][n0 .. n0+3], n0 = nb1 + ni*16 + fq*4
#pragma unroll
    for (int ni = 0; ni < NI1; ++ni) {
        const int n0 = nb1 + ni * 16 + fq * 4;
        float o[4], sv[4];
        unpack4<T>(small1[ni], sv);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = (MODE == 0) ? fmaxf(rnd<T>(acc1[ni][j] + sv[j]), 0.f) : (sv[j] > 0.f ? acc1[ni][j] : 0.f);
        *reinterpret_cast<u32x2*>(Hs + mrow * HROW + (((n0 >> 3) ^ (mrow & HSW)) << 4) + (fq & 1) * 8) =
            u32x2{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
    }

    __syncthreads();  // publishes the H tile (and retires the last Wa tile's fragment reads)
#ifdef MM_AD_STAMP
    st2 = __builtin_amdgcn_s_memtime();
#endif

    // ---- second product: Y[32, HID] = H Wb^T (+ bb) + X in S2 chunks of CH2 columns; wave w owns 16 columns of each chunk
    frag fh[KS2];
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks)
        fh[ks] = *reinterpret_cast<const frag*>(Hs + mrow * HROW + (((ks * 4 + fq) ^ (frow & HSW)) << 4));
    // H to HBM, 16 bytes per lane, full rows
    static_assert(AD_BM * HCPR <= AD_THREADS, "one 16-byte piece of the H tile per thread");
    {
        const int row = tid / HCPR, ch = tid % HCPR;
        if (tid < AD_BM * HCPR && m0 + row < rows)
            *reinterpret_cast<u32x4*>(Hout + (size_t)(m0 + row) * R + ch * 8) =
                *reinterpret_cast<const u32x4*>(Hs + row * HROW + ((ch ^ (row & HSW)) << 4));
    }
    // a finished 64-column chunk of the output block is stored while the next chunk is computed (one 16-byte piece per
    // thread, 8 lanes per 128-byte line) instead of the whole 64 KiB block after the last chunk
    static_assert(CH2 == 64, "y_store pieces assume 64-column chunks");
    auto y_store = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if (xmover && m0 + xrow < rows)
            *reinterpret_cast<u32x4*>(Y + (size_t)(m0 + xrow) * HID + c * CH2 + xpc * 8) =
                *reinterpret_cast<const u32x4*>(Xs + xrow * XROW + (((c * 8 + xpc) ^ (xrow & 15)) << 4));
    };
    static_for<0, S2>([&](auto cc) {
        constexpr int c = decltype(cc)::value, s = S1 + c;
        w_store(std::integral_constant<int, s + 2>{});
        static_for<s + D1, s + D1 + 1>([&](auto nx) { w_issue(nx); });
        frags2(std::integral_constant<int, c + 1>{});
        if constexpr (c > 0) y_store(std::integral_constant<int, c - 1>{});  // final since the previous step's barrier
        f32x4 acc2[NI2];
#pragma unroll
        for (int i = 0; i < NI2; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
            for (int ni = 0; ni < NI2; ++ni) acc2[ni] = Mfma<T>::run(FB2[c][ks][ni], fh[ks], acc2[ni]);
#pragma unroll
        for (int ni = 0; ni < NI2; ++ni) {
            const int n0 = c * CH2 + wave * (CH2 / 4) + ni * 16 + fq * 4;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 0) unpack4<T>(*reinterpret_cast<const u32x2*>(Bs + n0 * 2), bv);
            u32x2* px = reinterpret_cast<u32x2*>(Xs + mrow * XROW + (((n0 >> 3) ^ (mrow & 15)) << 4) + (fq & 1) * 8);
            const u32x2 xv = *px;  // each X element is read and replaced by the one lane that owns it
            float o[4];
            o[0] = rnd<T>(acc2[ni][0] + bv[0]) + to_f32<T>((u16)(xv[0] & 0xffffu));
            o[1] = rnd<T>(acc2[ni][1] + bv[1]) + to_f32<T>((u16)(xv[0] >> 16));
            o[2] = rnd<T>(acc2[ni][2] + bv[2]) + to_f32<T>((u16)(xv[1] & 0xffffu));
            o[3] = rnd<T>(acc2[ni][3] + bv[3]) + to_f32<T>((u16)(xv[1] >> 16));
            *px = u32x2{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
        }
        __syncthreads();
    });
    y_store(std::integral_constant<int, S2 - 1>{});
#ifdef MM_AD_STAMP
    {
        const uint64_t st3 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (tid == 0) {
            uint32_t* dbg = reinterpret_cast<uint32_t*>(Y + (size_t)m0 * HID);
            dbg[0] = (uint32_t)(st1 - st0); dbg[1] = (uint32_t)(st2 - st1); dbg[2] = (uint32_t)(st3 - st2); dbg[3] = (uint32_t)(st0 & 0xffffffffu);
        }
    }
#endif
}

template <typename T, int MODE>
static int launch_bottleneck(void* Y, void* Hout, const void* X, const void* Wa, const void* ba, const void* Wb, const void* bb,
                             const void* Hsaved, int rows, int h, int r, hipStream_t st) {
    dim3 grid((rows + AD_BM - 1) / AD_BM), block(AD_THREADS);
#define MM_AD_CASE(HID, RR)                                                                                             \
    if (h == HID && r == RR) {                                                                                          \
        bottleneck_k<T, HID, RR, MODE><<<grid, block, 0, st>>>((u16*)Y, (u16*)Hout, (const u16*)X, (const u16*)Wa,       \
                                                               (const u16*)ba, (const u16*)Wb, (const u16*)bb,          \
                                                               (const u16*)Hsaved, rows);                               \
        MM_CHECK_LAUNCH();                                                                                              \
        return MM_OK;                                                                                                   \
    }
    MM_AD_CASE(1024, 128)
    MM_AD_CASE(512, 64)
#undef MM_AD_CASE
    return MM_EUNSUPPORTED;
}

}  // namespace mm

using namespace mm;

extern "C" int mm_adapter_supported(int h, int r) { return (h == 1024 && r == 128) || (h == 512 && r == 64); }

static int check_adapter(const char* fn, int rows, int h, int r) {
    MM_CHECK_ARG(rows > 0, "%s: rows=%d must be positive", fn, rows);
    if (!mm_adapter_supported(h, r)) {
        mm::set_error("%s: (h=%d, r=%d) is not a fused shape (see mm_adapter_supported); use the mm_gemm_nt path", fn, h, r);
        return MM_EUNSUPPORTED;
    }
    return MM_OK;
}

extern "C" int mm_adapter_fwd(void* out, void* h1, const void* o, const void* w1, const void* b1, const void* w2, const void* b2,
                              int rows, int h, int r, int dtype, void* stream) {
    MM_CHECK_ARG(out && h1 && o && w1 && b1 && w2 && b2, "mm_adapter_fwd: null pointer");
    if (int rc = check_adapter("mm_adapter_fwd", rows, h, r)) return rc;
    MM_CHECK_ARG(aligned16(out) && aligned16(h1) && aligned16(o) && aligned16(w1) && aligned16(w2) && aligned16(b1) && aligned16(b2),
                 "mm_adapter_fwd: pointers must be 16-byte aligned");
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        return launch_bottleneck<T, 0>(out, h1, o, w1, b1, w2, b2, nullptr, rows, h, r, (hipStream_t)stream);
    });
}

extern "C" int mm_adapter_bwd_dx(void* dx, void* dpre, const void* dy, const void* w2, const void* w1, const void* h1, int rows,
                                 int h, int r, int dtype, void* stream) {
    MM_CHECK_ARG(dx && dpre && dy && w2 && w1 && h1, "mm_adapter_bwd_dx: null pointer");
    if (int rc = check_adapter("mm_adapter_bwd_dx", rows, h, r)) return rc;
    MM_CHECK_ARG(aligned16(dx) && aligned16(dpre) && aligned16(dy) && aligned16(w2) && aligned16(w1) && aligned16(h1),
                 "mm_adapter_bwd_dx: pointers must be 16-byte aligned");
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        return launch_bottleneck<T, 1>(dx, dpre, dy, w2, nullptr, w1, nullptr, h1, rows, h, r, (hipStream_t)stream);
    });
}
