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
// so one kernel template serves both.  Structure: a workgroup owns 32 token rows (256 workgroups at 8192 tokens), eight
// waves (two per SIMD): wave (w & 3, w >> 2) owns a quarter of the columns and one of the two 16-row MFMA tiles;
//   * the 32 x h block of X lives in LDS - it is the MFMA operand of the first product AND the residual of the second; it
//     arrives in 64-column chunks that travel with the weight tiles, the output block is assembled in place and each
//     finished 64-column chunk is stored (full 128-byte lines) while the next one is computed;
//   * the weights (2 x 256 KiB, L2-resident, shared by every workgroup) stream through an LDS ring in 32 steps (16 K-tiles of
//     Wa, 16 column chunks of Wb), one barrier per step; H goes through an 8 KiB LDS tile between the two products (and to HBM
//     once, coalesced: the backward needs it);
//   * backward: the operands are W2^T and W1^T of the weights AS STORED: tiles are staged [k][n] and the MFMA fragments
//     are read transposed with ds_read_b64_tr_b16 (32-byte segments XOR-swizzled so the 8 rows a half-wave touches fall in
//     different banks) - no per-step W^T copies of the trainable weights.
// Rounding points are exactly those of the unfused path (mm_gemm_nt with MM_EPI_BIAS_RELU / MM_EPI_BIAS + residual /
// MM_EPI_RELU_BWD / residual), and each output element's K-order is the same, so the results are bit-identical to it.
// History (device time per launch at 8192 x 1024 x 128): two skinny GEMMs 27 us -> register-staged ring, 4 waves 19.2 ->
// 8 waves, X / output streamed per chunk 16.8 -> LDS-DMA ring (below) 14.6.  Stamps (-DMM_AD_STAMP): prologue 3.2k cycles,
// first product 490 cycles per step, second 720 (its per-chunk barrier + in-place epilogue chain is the next thing to cut).
#include "gemm_common.h"
#include "mfma32.h"

namespace mm {

constexpr int AD_BM = 32;
constexpr int AD_THREADS = 512;  // 8 waves: two per SIMD, so one wave's LDS / L2 latency hides behind its partner's MFMAs

// =====================================================================================================================
// bottleneck_dma_k - the weight tiles and the X chunks are staged global -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no VGPR round trip and no ds_write issue cost), a five-buffer
// ring filled three steps ahead, and EVERY LDS access written as inline asm: the compiler orders any LDS access it can see
// behind all in-flight LDS-DMA (s_waitcnt vmcnt(0)), which would drain the ring every step.  The waits are therefore
// counted by hand: `younger()` replays the wave's VMEM issue order at compile time (the kernel is fully unrolled) and
// returns how many operations were issued after a tile's last DMA, i.e. the vmcnt that proves the tile has landed.
// LDS images: weight tiles [n][k] rows of 128 / 2R bytes with the XOR chunk swizzles of the fragment-read formulas below
// (applied on the DMA SOURCE address); the X block is chunk-major ([chunk][32 rows][128 B], the GEMM kernels' A-tile
// image) because one DMA instruction writes 1 KiB of contiguous LDS.
constexpr int AD_PD = 3;           // tiles in flight beyond the one being published
constexpr int AD_NB = AD_PD + 2;   // ring buffers

constexpr int ad_waitcnt_imm(int vm, int lgkm) { return (vm & 0xF) | (7 << 4) | ((lgkm & 0xF) << 8) | (((vm >> 4) & 3) << 14); }
#define AD_WAIT(vm, lgkm)                                          \
    do {                                                           \
        __builtin_amdgcn_s_waitcnt(ad_waitcnt_imm(vm, lgkm));      \
        __builtin_amdgcn_sched_barrier(0);                         \
    } while (0)

__device__ __forceinline__ u32x4 ad_rd16(uint32_t a) {
    u32x4 d;
    asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ u32x2 ad_rd8(uint32_t a) {
    u32x2 d;
    asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ u32x2 ad_rd8_tr(uint32_t a) {
    u32x2 d;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ void ad_wr8(uint32_t a, u32x2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void ad_wr16(uint32_t a, u32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void ad_barrier() {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// VMEM operations a wave has issued AFTER the last DMA instruction of tile `tile`, counted at the end of step `step`
// (step -1 = right after the prologue); 63 ("no wait") when the wave issues no DMA for that tile.  The waves split roles so
// that the in-order VMEM queue of the waves that confirm the weight tiles holds nothing slow: waves 0-3 ("movers") bring
// the X chunks in (one DMA instruction per first-product tile), load the residual-tail operands (CMB) and store the output
// chunks - HBM-latency traffic; waves 4-7 ("loaders") issue ALL weight DMAs (L2-resident, NWL instructions per tile) and
// nothing else but their piece of the H-tile store.  Issue order per wave: prologue tiles 0..PD; then in step s the DMA
// of tile s+PD+1, then (CMB movers, from step S1-2 on) operand loads of later chunks, then that step's global
// store (H tile in step S1, output chunk c-1 in step S1+c: movers only).
constexpr int ad_younger(int tile, int step, bool mover, int NWL, int S1, int S, bool hstore, bool cmb) {
    auto ops_of = [&](int t) { return mover ? (t < S1 ? 1 : 0) : NWL; };
    if (tile >= S || ops_of(tile) == 0) return 63;
    int count = -1;  // -1: the tile's DMA not seen yet
    auto emit_tile = [&](int t) {
        if (count >= 0) count += ops_of(t);
        if (t == tile) count = 0;
    };
    for (int t = 0; t <= AD_PD; ++t)
        if (t < S) emit_tile(t);
    for (int s = 0; s <= step; ++s) {
        if (s + AD_PD + 1 < S) emit_tile(s + AD_PD + 1);
        // (CMB: the movers' residual-tail operand loads start in step S1-2, after their last X-chunk wait (end of step S1-3),
        //  and the movers confirm no tile after that: the loads never enter a counted wait)
        const bool st = (s == S1) ? hstore : (s > S1 ? mover : false);
        if (st && count >= 0) count += 1;
    }
    return count < 0 ? 0 : (count > 63 ? 63 : count);
}

// CMB (forward only): the GPT-J residual tail rides on the output store - Y = [res +] ((attn + attn_b) + (adapter(X) + mlp_b))
// with the eager roundings of mm_gptj_combine (megatron/model/transformer.py:747-766 at hidden_dropout = 0).
struct AdCombine {
    const u16* res;     // [rows, HID] or nullptr
    const u16* attn;    // [rows, HID]
    const u16* attn_b;  // [HID]
    const u16* mlp_b;   // [HID]
};

template <typename T, int HID, int R, int MODE, bool CMB>  // MODE 0: forward, 1: backward (data gradient)
__global__ __launch_bounds__(AD_THREADS) void bottleneck_dma_k(u16* __restrict__ Y, u16* __restrict__ Hout, const u16* __restrict__ X,
                                                               const u16* __restrict__ Wa, const u16* __restrict__ ba,
                                                               const u16* __restrict__ Wb, const u16* __restrict__ bb,
                                                               const u16* __restrict__ Hsaved, int rows, AdCombine cmb) {
    static_assert(!CMB || MODE == 0, "the residual tail is fused into the forward only");
    static_assert(!CMB || (HID / 64 >= 5 && AD_PD == 3), "CMB operand schedule assumes >= 5 chunks and the last X wait in step S1-3");
    typedef typename Mfma<T>::frag frag;
    constexpr int HROW = R * 2, HCPR = R / 8;
    constexpr int HSW = HCPR >= 16 ? 15 : HCPR - 1;
    constexpr int NI1 = R / 64, S1 = HID / 64, S2 = HID / 64, S = S1 + S2, KS2 = R / 32;
    constexpr int RING = R * 128;              // bytes per tile: Wa tile [R][64] / [64][R], Wb chunk [64][R] / [R][64]
    constexpr int NW = RING / 1024 / 4;        // weight DMA instructions per LOADER wave (waves 4-7) per tile
    static_assert(RING % 4096 == 0 && NW >= 1, "a tile must be a whole number of DMA instructions per loader wave");
    constexpr int XS_BYTES = S1 * 4096, HS_BYTES = AD_BM * HROW;
    constexpr int T1SEG = HROW / 32;
    constexpr bool HSTORE_ALL = (AD_BM * HCPR == AD_THREADS);  // do the loader waves store a piece of the H tile too?
    static_assert(AD_BM * HCPR == AD_THREADS || AD_BM * HCPR == 256, "H tile store: all waves or exactly the X movers");
    __shared__ __attribute__((aligned(1024))) char smem[XS_BYTES + HS_BYTES + AD_NB * RING + HID * 2 * (CMB ? 3 : 1)];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
    const uint32_t XS = lds0, HS = lds0 + XS_BYTES, WS = HS + HS_BYTES, BS = WS + AD_NB * RING;
    const uint32_t BSA = BS + HID * 2, BSM = BS + HID * 4;  // CMB: attention / MLP output biases
    const int tid = threadIdx.x, lane = tid & 63, frow = lane & 15, fq = lane >> 4;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index as a scalar: uniform branches, scalar LDS bases
    const int wave = w8 & 3, wmi = w8 >> 2;
    const int m0 = blockIdx.x * AD_BM;
    const int mrow = wmi * 16 + frow;
    const bool xmover = w8 < 4;
#ifdef MM_AD_STAMP
    const uint64_t st0 = __builtin_amdgcn_s_memtime();
    uint64_t st1 = 0, st2 = 0;
#endif
    auto tsw = [](int row, int nseg) { return nseg >= 8 ? ((row & 3) | (((row >> 3) & 1) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 1) << 1)); };

    // ---- small operands first (they sit in front of every DMA in the in-order VMEM queue)
    const int nb1 = wave * (R / 4);
    u32x2 small1[NI1];
    u32x4 bbreg = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int ni = 0; ni < NI1; ++ni) {
        const int n0 = nb1 + ni * 16 + fq * 4;
        if (MODE == 0) small1[ni] = *reinterpret_cast<const u32x2*>(ba + n0);
        else small1[ni] = *reinterpret_cast<const u32x2*>(Hsaved + (size_t)min(m0 + mrow, rows - 1) * R + n0);
    }
    if (MODE == 0 && tid < HID / 8) bbreg = *reinterpret_cast<const u32x4*>(bb + tid * 8);
    if (CMB && tid >= HID / 8 && tid < 3 * (HID / 8)) {
        const int t2 = tid - HID / 8;
        bbreg = (t2 < HID / 8) ? *reinterpret_cast<const u32x4*>(cmb.attn_b + t2 * 8) : *reinterpret_cast<const u32x4*>(cmb.mlp_b + (t2 - HID / 8) * 8);
    }

    // ---- per-lane DMA sources.  Physical 16-byte slot P of a tile image holds logical piece (row, ch) with the swizzle of
    // the matching read formula below; instruction j of a tile fills slots 64j .. 64j+63.
    const u16* wa_src[NW];
    const u16* wb_src[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int j = (w8 & 3) * NW + i;  // (only the loader waves, w8 >= 4, use these)
        if (MODE == 0) {
            {   // Wa = W1 [R][HID]: tile s = columns 64s..: image [R rows][128 B], piece ^ ((row >> 1) & 7)
                const int row = 8 * j + (lane >> 3), q = lane & 7;
                wa_src[i] = Wa + (size_t)row * HID + ((q ^ ((row >> 1) & 7)) << 3);
            }
            {   // Wb = W2 [HID][R]: chunk c = rows 64c..: image [64 rows][HROW], piece ^ (row & HSW)
                const int row = (1024 / HROW) * j + lane / HCPR, q = lane % HCPR;
                wb_src[i] = Wb + (size_t)row * R + ((q ^ (row & HSW)) << 3);
            }
        } else {
            {   // Wa = stored W2 [HID][R]: tile s = rows 64s..: image [64 k][HROW], 32-byte segments ^ tsw(row, T1SEG)
                const int row = (1024 / HROW) * j + lane / HCPR, q = lane % HCPR;
                wa_src[i] = Wa + (size_t)row * R + (((((q >> 1) ^ tsw(row, T1SEG)) << 1) | (q & 1)) << 3);
            }
            {   // Wb = stored W1 [R][HID]: chunk c = columns 64c..: image [R k][128 B], segments ^ tsw(row, 4)
                const int row = 8 * j + (lane >> 3), q = lane & 7;
                wb_src[i] = Wb + (size_t)row * HID + (((((q >> 1) ^ tsw(row, 4)) << 1) | (q & 1)) << 3);
            }
        }
    }
    constexpr size_t WA_STRIDE = (MODE == 0) ? 64 : (size_t)64 * R;   // elements between consecutive Wa tiles
    constexpr size_t WB_STRIDE = (MODE == 0) ? (size_t)64 * R : 64;  // ... Wb chunks
    // X chunk s = columns 64s.. of the 32 rows: image [32 rows][128 B], piece ^ ((row >> 1) & 7); wave w < 4 fills rows 8w..8w+7
    const int xrow = (tid >> 3) & 31, xpc = tid & 7;
    const u16* x_src = X + (size_t)min(m0 + xrow, rows - 1) * HID + ((xpc ^ ((xrow >> 1) & 7)) << 3);
    auto fill = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t < S) {
            if constexpr (t < S1) {
                if (xmover)
                    __builtin_amdgcn_global_load_lds((const void*)(x_src + t * 64), (lds_void*)(smem + t * 4096 + w8 * 1024), 16, 0, 0);
            }
            if (!xmover) {
                char* buf = smem + XS_BYTES + HS_BYTES + (t % AD_NB) * RING + (w8 & 3) * NW * 1024;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    const u16* src;
                    if constexpr (t < S1) src = wa_src[i] + (size_t)t * WA_STRIDE;
                    else src = wb_src[i] + (size_t)(t - S1) * WB_STRIDE;
                    __builtin_amdgcn_global_load_lds((const void*)src, (lds_void*)(buf + i * 1024), 16, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the DMA block where the wait arithmetic (ad_younger) assumes it is
    };
    auto wait_tile = [&](auto tc, auto sc) {  // tile `t` landed (this wave's share), counted at the end of step `st`; + all LDS ops
        constexpr int t = decltype(tc)::value, st = decltype(sc)::value;
        constexpr int ym = ad_younger(t, st, true, NW, S1, S, true, CMB), yo = ad_younger(t, st, false, NW, S1, S, HSTORE_ALL, CMB);
        if (xmover) AD_WAIT(ym, 0);
        else AD_WAIT(yo, 0);
    };

    static_for<0, AD_PD + 1>([&](auto tc) { fill(tc); });
    if (MODE == 0 && tid < (CMB ? 3 : 1) * (HID / 8)) ad_wr16(BS + tid * 16, bbreg);  // [bb | attn_b | mlp_b], 2 KiB each
    static_assert(3 * (HID / 8) <= AD_THREADS, "one 16-byte bias piece per thread");
    // CMB: operands of the residual tail for output chunk k (this thread's piece), loaded two steps before they are used
    u32x4 c_res[S2], c_att[S2];
    const size_t c_off = (size_t)min(m0 + xrow, rows - 1) * HID + xpc * 8;
    auto cmb_issue = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (CMB && k >= 0 && k < S2) {
            if (xmover) {
                c_att[k] = *reinterpret_cast<const u32x4*>(cmb.attn + c_off + k * 64);
                c_res[k] = cmb.res ? *reinterpret_cast<const u32x4*>(cmb.res + c_off + k * 64) : u32x4{0u, 0u, 0u, 0u};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto cmb_apply = [&](auto kc, u32x4 yv, u32x4 bav, u32x4 bmv) {  // eager roundings of gptj_combine_k (elementwise.hip)
        constexpr int k = decltype(kc)::value;
        if constexpr (CMB) {
            float a[8], at[8], xr[8], b1[8], b2[8], o[8];
            unpack8<T>(yv, a);
            unpack8<T>(c_att[k], at);
            unpack8<T>(c_res[k], xr);
            unpack8<T>(bav, b1);
            unpack8<T>(bmv, b2);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t3 = rnd<T>(rnd<T>(at[j] + b1[j]) + rnd<T>(a[j] + b2[j]));
                o[j] = cmb.res ? xr[j] + t3 : t3;
            }
            return pack8<T>(o);
        } else {
            return yv;
        }
    };

    // transposed fragment of a [k][n] tile image (row bytes rb, nseg 32-byte segments per row): lane (i = lane&15, g = lane>>4)
    // gets tile[kb + 8g + j][n0 + i], j = 0..7 (lane 4q+p of a 16-lane group supplies the address of row q, columns 4p..4p+3)
    auto tr_frag16 = [&](uint32_t buf, int rb, int nseg, int kb, int n0) {
        const int q = frow >> 2, pp = frow & 3;
        const int r0 = kb + 8 * fq + q, r1 = r0 + 4;
        const u32x2 a = ad_rd8_tr(buf + r0 * rb + (((n0 >> 4) ^ tsw(r0, nseg)) << 5) + 8 * pp);
        const u32x2 b = ad_rd8_tr(buf + r1 * rb + (((n0 >> 4) ^ tsw(r1, nseg)) << 5) + 8 * pp);
        return u32x4{a[0], a[1], b[0], b[1]};
    };
    u32x4 FA[S1][2], FB[S1][2][NI1], FB2[S2][KS2];
    auto frags1 = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s < S1) {
            const uint32_t buf = WS + (s % AD_NB) * RING;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                FA[s][kk] = ad_rd16(XS + s * 4096 + mrow * 128 + (((kk * 4 + fq) ^ ((mrow >> 1) & 7)) << 4));
#pragma unroll
                for (int ni = 0; ni < NI1; ++ni) {
                    if (MODE == 0) {
                        const int n = nb1 + ni * 16 + frow;
                        FB[s][kk][ni] = ad_rd16(buf + n * 128 + (((kk * 4 + fq) ^ ((n >> 1) & 7)) << 4));
                    } else {
                        FB[s][kk][ni] = tr_frag16(buf, HROW, T1SEG, kk * 32, nb1 + ni * 16);
                    }
                }
            }
        }
    };
    auto frags2 = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if constexpr (c < S2) {
            const uint32_t buf = WS + ((S1 + c) % AD_NB) * RING;
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                if (MODE == 0) {
                    const int n = wave * 16 + frow;
                    FB2[c][ks] = ad_rd16(buf + n * HROW + (((ks * 4 + fq) ^ (n & HSW)) << 4));
                } else {
                    FB2[c][ks] = tr_frag16(buf, 128, 4, ks * 32, wave * 16);
                }
            }
        }
    };

    // ---- first product
    f32x4 acc1[NI1];
#pragma unroll
    for (int i = 0; i < NI1; ++i) acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    wait_tile(std::integral_constant<int, 1>{}, std::integral_constant<int, -1>{});  // tiles 0 and 1 (and the bias image)
    ad_barrier();
#ifdef MM_AD_STAMP
    st1 = __builtin_amdgcn_s_memtime();
#endif
    frags1(std::integral_constant<int, 0>{});
    AD_WAIT(63, 0);
    static_for<0, S1>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        frags1(std::integral_constant<int, s + 1>{});
        if constexpr (s + 1 == S1) frags2(std::integral_constant<int, 0>{});
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < NI1; ++ni)
                acc1[ni] = Mfma<T>::run(__builtin_bit_cast(frag, FB[s][kk][ni]), __builtin_bit_cast(frag, FA[s][kk]), acc1[ni]);
        fill(std::integral_constant<int, s + AD_PD + 1>{});  // issued behind the MFMAs: the DMA issue slots overlap their execution
        // CMB operand schedule (HBM latency is several steps): chunks 0-3 in step S1-2, chunk 4 in S1-1, then chunk c+5 in
        // second-product step c; chunk k is consumed in second-product step k+1
        if constexpr (s + 2 == S1) static_for<0, 4>([&](auto kc) { cmb_issue(kc); });
        if constexpr (s + 1 == S1) cmb_issue(std::integral_constant<int, 4>{});
        if constexpr (s + 1 == S1) {  // the H tile: lane holds H[mrow][n0 .. n0+3], n0 = nb1 + ni*16 + fq*4
#pragma unroll
            for (int ni = 0; ni < NI1; ++ni) {
                const int n0 = nb1 + ni * 16 + fq * 4;
                float o[4], sv[4];
                unpack4<T>(small1[ni], sv);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = (MODE == 0) ? fmaxf(rnd<T>(acc1[ni][j] + sv[j]), 0.f) : (sv[j] > 0.f ? acc1[ni][j] : 0.f);
                ad_wr8(HS + mrow * HROW + (((n0 >> 3) ^ (mrow & HSW)) << 4) + (fq & 1) * 8, u32x2{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])});
            }
        }
        wait_tile(std::integral_constant<int, s + 2>{}, sc);
        ad_barrier();
    });

#ifdef MM_AD_STAMP
    st2 = __builtin_amdgcn_s_memtime();
#endif
    // ---- second product: wave (wave, wmi) owns 16 columns of each 64-column chunk for its 16 rows
    u32x4 fh[KS2];
    static_for<0, S2>([&](auto cc) {
        constexpr int c = decltype(cc)::value, s = S1 + c;
        u32x4 hv = u32x4{0u, 0u, 0u, 0u}, yv = u32x4{0u, 0u, 0u, 0u};
        if constexpr (c == 0) {
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) fh[ks] = ad_rd16(HS + mrow * HROW + (((ks * 4 + fq) ^ (frow & HSW)) << 4));
            if (tid < AD_BM * HCPR) {
                const int row = tid / HCPR, ch = tid % HCPR;
                hv = ad_rd16(HS + row * HROW + ((ch ^ (row & HSW)) << 4));
            }
        }
        frags2(std::integral_constant<int, c + 1>{});
        u32x4 ybav = u32x4{0u, 0u, 0u, 0u}, ybmv = u32x4{0u, 0u, 0u, 0u};
        if constexpr (c > 0) {
            if (xmover) {
                yv = ad_rd16(XS + (c - 1) * 4096 + xrow * 128 + ((xpc ^ ((xrow >> 1) & 7)) << 4));
                if (CMB) {
                    ybav = ad_rd16(BSA + ((c - 1) * 64 + xpc * 8) * 2);
                    ybmv = ad_rd16(BSM + ((c - 1) * 64 + xpc * 8) * 2);
                }
            }
        }
        const int n0 = c * 64 + wave * 16 + fq * 4;
        const uint32_t xaddr = XS + c * 4096 + mrow * 128 + (((wave * 2 + (fq >> 1)) ^ ((mrow >> 1) & 7)) << 4) + (fq & 1) * 8;
        const u32x2 xv = ad_rd8(xaddr);  // each X element is read and replaced by the one lane that owns it
        u32x2 bvr = u32x2{0u, 0u};
        if (MODE == 0) bvr = ad_rd8(BS + n0 * 2);
        if constexpr (c == 0) AD_WAIT(63, 0);  // fh (later chunks: everything these MFMAs read landed before the last barrier)
        f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
            acc2 = Mfma<T>::run(__builtin_bit_cast(frag, FB2[c][ks]), __builtin_bit_cast(frag, fh[ks]), acc2);
        fill(std::integral_constant<int, s + AD_PD + 1>{});
        cmb_issue(std::integral_constant<int, c + 5>{});
        AD_WAIT(63, 0);  // xv, bvr, yv, hv, the next chunk's fragments
        if constexpr (c == 0) {
            if (tid < AD_BM * HCPR) {
                const int row = tid / HCPR, ch = tid % HCPR;
                if (m0 + row < rows) *reinterpret_cast<u32x4*>(Hout + (size_t)(m0 + row) * R + ch * 8) = hv;
            }
        } else {
            if (xmover && m0 + xrow < rows)
                *reinterpret_cast<u32x4*>(Y + (size_t)(m0 + xrow) * HID + (c - 1) * 64 + xpc * 8) =
                    cmb_apply(std::integral_constant<int, c - 1>{}, yv, ybav, ybmv);
        }
        __builtin_amdgcn_sched_barrier(0);
        float bv[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
        if (MODE == 0) unpack4<T>(bvr, bv);
        o[0] = rnd<T>(acc2[0] + bv[0]) + to_f32<T>((u16)(xv[0] & 0xffffu));
        o[1] = rnd<T>(acc2[1] + bv[1]) + to_f32<T>((u16)(xv[0] >> 16));
        o[2] = rnd<T>(acc2[2] + bv[2]) + to_f32<T>((u16)(xv[1] & 0xffffu));
        o[3] = rnd<T>(acc2[3] + bv[3]) + to_f32<T>((u16)(xv[1] >> 16));
        ad_wr8(xaddr, u32x2{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])});
        wait_tile(std::integral_constant<int, s + 2>{}, std::integral_constant<int, s>{});
        ad_barrier();
    });
    if (xmover) {
        const u32x4 yv = ad_rd16(XS + (S2 - 1) * 4096 + xrow * 128 + ((xpc ^ ((xrow >> 1) & 7)) << 4));
        u32x4 ybav = u32x4{0u, 0u, 0u, 0u}, ybmv = u32x4{0u, 0u, 0u, 0u};
        if (CMB) {
            ybav = ad_rd16(BSA + ((S2 - 1) * 64 + xpc * 8) * 2);
            ybmv = ad_rd16(BSM + ((S2 - 1) * 64 + xpc * 8) * 2);
        }
        AD_WAIT(63, 0);
        if (m0 + xrow < rows)
            *reinterpret_cast<u32x4*>(Y + (size_t)(m0 + xrow) * HID + (S2 - 1) * 64 + xpc * 8) =
                cmb_apply(std::integral_constant<int, S2 - 1>{}, yv, ybav, ybmv);
    }
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

template <typename T, int MODE, bool CMB = false>
static int launch_bottleneck(void* Y, void* Hout, const void* X, const void* Wa, const void* ba, const void* Wb, const void* bb,
                             const void* Hsaved, int rows, int h, int r, hipStream_t st, AdCombine cmb = AdCombine{}) {
    dim3 grid((rows + AD_BM - 1) / AD_BM), block(AD_THREADS);
#define MM_AD_CASE(HID, RR)                                                                                             \
    if (h == HID && r == RR) {                                                                                          \
        bottleneck_dma_k<T, HID, RR, MODE, CMB><<<grid, block, 0, st>>>((u16*)Y, (u16*)Hout, (const u16*)X,             \
                                                                        (const u16*)Wa, (const u16*)ba, (const u16*)Wb, \
                                                                        (const u16*)bb, (const u16*)Hsaved, rows, cmb); \
        MM_CHECK_LAUNCH();                                                                                              \
        return MM_OK;                                                                                                   \
    }
    MM_AD_CASE(1024, 128)
    MM_AD_CASE(512, 64)
    MM_AD_CASE(512, 128)  // summit-70m-openclipH: h = 512 with the default adapter_downsample_factor 4
#undef MM_AD_CASE
    return MM_EUNSUPPORTED;
}

}  // namespace mm

using namespace mm;

extern "C" int mm_adapter_supported(int h, int r) { return (h == 1024 && r == 128) || (h == 512 && (r == 64 || r == 128)); }

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

extern "C" int mm_adapter_fwd_combine(void* out, void* h1, const void* o, const void* w1, const void* b1, const void* w2, const void* b2,
                                      const void* residual, const void* attn, const void* attn_bias, const void* mlp_bias, int rows,
                                      int h, int r, int dtype, void* stream) {
    MM_CHECK_ARG(out && h1 && o && w1 && b1 && w2 && b2 && attn && attn_bias && mlp_bias, "mm_adapter_fwd_combine: null pointer");
    if (int rc = check_adapter("mm_adapter_fwd_combine", rows, h, r)) return rc;
    MM_CHECK_ARG(aligned16(out) && aligned16(h1) && aligned16(o) && aligned16(w1) && aligned16(w2) && aligned16(b1) && aligned16(b2) &&
                     aligned16(attn) && aligned16(attn_bias) && aligned16(mlp_bias) && (!residual || aligned16(residual)),
                 "mm_adapter_fwd_combine: pointers must be 16-byte aligned");
    const AdCombine cmb{(const u16*)residual, (const u16*)attn, (const u16*)attn_bias, (const u16*)mlp_bias};
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        return launch_bottleneck<T, 0, true>(out, h1, o, w1, b1, w2, b2, nullptr, rows, h, r, (hipStream_t)stream, cmb);
    });
}
