// gemm_pp.hip - "ping-pong" bf16/f16 MFMA GEMM  C[M,N] = A[M,K] * B[N,K]^T  with 256 x BN tiles (BN = 256 or 128), gfx950.
//
// Same contract and epilogues as gemm.hip / gemm256.hip (reference megatron/mpu/layers.py:557,745 F.linear).
//
// Why another kernel: tools/decompose_gemm256.py on the 256x256 kernel of gemm256.hip (M=8192 N=4096 K=1024, bias-GeLU):
// MFMAs + barriers alone 41 us, loads + LDS reads + epilogue alone 53 us, the full kernel 82 us - close to the SUM.  In that
// kernel all eight waves run the same program behind one barrier per K-tile, so the two waves of a SIMD reach their
// global->LDS issue (8 LDS-DMA pieces, ~60-180 cycles of issue each), their 24 fragment reads and their 64 MFMAs at the same
// time and take turns instead of overlapping.  Here the two waves of a SIMD run OPPOSITE phases:
//
//   waves 0-3 (group 0) | compute t   | load       | compute t+1 | load       | ...
//   waves 4-7 (group 1) | load        | compute t  | load        | compute t+1| ...
//                       ^ one s_barrier per phase (two per 64-deep K-tile)
//
//   compute phase: the wave's 2*MI*4 MFMAs (16x16x32) of one K-tile, back to back, every operand already in registers;
//   load phase   : the wave's share of the global->LDS DMA of a later tile, then ALL fragment reads of its next tile
//                  (2*(MI+4) ds_read_b128 into the registers the compute phase has just finished with).
//   (A workgroup's waves go to the SIMDs in cyclic order, so every SIMD holds one wave of each group: in every phase each
//    SIMD has exactly one wave issuing MFMAs and one wave issuing memory instructions.)
//
// LDS (round 3): the A operand of a K-tile (256 x 128 B) and the B operand (BN x 128 B) live in SEPARATE rings - A in two
// stages, B in three (BN = 256: 2 x 32 + 3 x 32 = 160 KiB, all of a CU's LDS).  Group 0's waves fetch all of A, group 1's all of
// B.  With two whole-tile stages (round 2) group 1 could only fetch tile t+1 in the phase right before group 0 reads it, and
// waited out the full fetch latency in that phase, every K-tile: measured with that wait removed (results wrong, timing only)
// the K loop ran 18 % faster at K = 16384 and the bias-GeLU GEMM 6 %.  The third B stage lets group 1 fetch B of tile t+2 while
// tile t is still being read; its wait covers the batch issued one K-tile earlier (counted vmcnt).  Hazard notes sit next to
// the code.  Lane-linear 8-row images, 16-byte chunk index XOR (row>>1)&7 on the DMA source address and on the read address.
//
// Persistent (round 3): at most one workgroup per CU; workgroup b walks output tiles b, b + grid, ...  Before the epilogue of a
// tile it issues K-tile 0 of its next tile (A stage 0, B stage 0); the epilogue stages through B stages 1-2 in two passes.
//
// BN = 256: waves 2(M) x 4(N), 128x64 per wave (the h->4h class and the LM head: >= 256 tiles).
// BN = 128: waves 4(M) x 2(N),  64x64 per wave (the N = h class: M=8192, N=1024 is exactly 256 tiles = one per CU, where
//           256x256 tiles would leave half the chip idle and 128x128 tiles move 2x the L2->LDS bytes per flop).
#include "gemm_common.h"

namespace mm {

template <int BN_> struct PP {
    static constexpr int BM = 256, BN = BN_;
    static constexpr int ROWB = 128;  // bytes per tile row (64 k-elements)
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    static constexpr int NSA = 2, NSB = 3;                               // ring depths
    static constexpr int B_BASE = NSA * A_BYTES, LDS = B_BASE + NSB * B_BYTES;
    static constexpr int NPA = BM / 32, NPB = BN / 32;                   // LDS-DMA instructions per wave and K-tile: group 0 (A), group 1 (B)
    static constexpr int NPMAX = NPA > NPB ? NPA : NPB;
    static constexpr int WAVES_N = BN / 64, WAVES_M = 8 / WAVES_N;
    static constexpr int WM = BM / WAVES_M, MI = WM / 16;               // wave tile WM x 64: MI x 4 accumulators
    static constexpr int NREADS = 2 * (MI + 4);                         // ds_read_b128 per wave per K-tile
    static_assert(LDS <= 160 * 1024, "one workgroup per CU: at most the CU's 160 KiB");
};

constexpr int pp_waitcnt(int vm, int lgkm) { return (vm & 0xF) | (7 << 4) | ((lgkm & 0xF) << 8) | (((vm >> 4) & 3) << 14); }
#define PP_WAIT(vm, lgkm) __builtin_amdgcn_s_waitcnt(pp_waitcnt(vm, lgkm))

template <int OFF> __device__ __forceinline__ u32x4 pp_lds_read16(uint32_t addr) {
    u32x4 d;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
    return d;
}

__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// DBG (developer decomposition builds only, mm_debug_gemm256 variants 16..): bit 0 = no global->LDS DMA inside the K loop, bit 1 =
// no LDS fragment reads inside the loop, bit 2 = no epilogue.  DBG = 0 everywhere in the product.
// SPLITK = 2 (BN = 256 only): two workgroups per output tile, each over half of K, for shapes with too few tiles to fill the
// chip (the N = h class at M = 8192: 128 tiles; the LM head's data gradient: K = 50304).  The exchange is symmetric: after its
// K loop a workgroup hands the fp32 accumulators of ONE of its two wave rows (128 x 256, 128 KiB, lane-linear 16-byte stores)
// to its partner and finishes the other wave row with the partner's partials added, so both workgroups run an epilogue over
// half a tile.  Hand-off per cdna_hip_programming.md Guideline 16 (R1): write-through (sc1) stores, every wave drains, workgroup barrier, one
// lane's agent-scope flag store; the same lane polls the partner's flag (relaxed, bounded), one agent-scope acquire, barrier,
// plain loads.  Partners sit 8 apart in dispatch order (same XCD under round-robin placement - speed only), the grid is at most
// one workgroup per CU, and a spin that runs out raises ws_flags[timeout] instead of hanging.  own + partner is one fp32 add
// per element either way round: the result does not depend on which side finishes a row.
struct SplitArgs {
    float* slabs;      // [tiles][2 (receiving slice)][4 (wave column)][128 x 64 fp32, (ni, mi, lane) order]
    unsigned* flags;   // [tiles * 2] "my half for you is published", then [1] timeout marker; zeroed before the launch
};

template <typename T, int EPI, bool RES, int BN, int DBG = 0, int SPLITK = 1>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_k(u16* __restrict__ C, u16* __restrict__ aux, const u16* __restrict__ A,
                                                       const u16* __restrict__ B, const u16* __restrict__ bias,
                                                       const u16* __restrict__ Rsd, int M, int N, int K, int lda, int ldb,
                                                       int ldc, int tiles_m, int tiles_n, int group_m, EpiExtra ex,
                                                       SplitArgs sp = SplitArgs{nullptr, nullptr}) {
    typedef PP<BN> P;
    constexpr int MI = P::MI;
    typedef typename Mfma<T>::frag frag;
    static_assert(SPLITK == 1 || (SPLITK == 2 && BN == 256), "split-K: two slices of a 256 x 256 tile");
    __shared__ __attribute__((aligned(16))) char smem[P::LDS];  // the ONLY LDS object of the kernel
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index as a SCALAR: everything derived from it (LDS-DMA bases in M0, group tests) stays in SGPRs.  As a vector value it
    // was spilled under the epilogue's register pressure and re-loaded inside the K loop - and a scratch reload waits vmcnt(0),
    // i.e. for every DMA in flight, once per phase.
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool g0 = wave < 4;  // group 0: waves 0-3
    const int w4 = wave & 3;
    int tm, tn;
    // split-K: blocks b and b + 8 (b & 8 == 0) are the two K-slices of tile (b & 7) + 8 * (b >> 4)
    const int slice = SPLITK == 2 ? (int)((blockIdx.x >> 3) & 1) : 0;
    // SPLITK == 1 is PERSISTENT: the launch is at most one workgroup per CU (256, a multiple of 8, so all tiles of a workgroup fall
    // into the same XCD run of tile_coords) and workgroup b walks tiles b, b + gridDim.x, ...
    int tidx = SPLITK == 2 ? (int)((blockIdx.x & 7) | ((blockIdx.x >> 4) << 3)) : (int)blockIdx.x;
    tile_coords(tidx, tiles_m, tiles_n, group_m, tm, tn);
    int bm = tm * P::BM, bn = tn * BN;
    const int wr = wave / P::WAVES_N, wc = wave % P::WAVES_N;
    const int frow = lane & 15, fq = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;

    // ---- global -> LDS staging.  A wave of group 0 fills rows i*32 + w4*8 + (lane>>3) of the A image with its instruction i
    // (i < NPA), a wave of group 1 the same rows of the B image (i < NPB); lane & 7 is the 16-byte LDS chunk.
    uint32_t goff[P::NPMAX];
    auto set_tile = [&](int bm_, int bn_) {
        const int rl = w4 * 8 + (lane >> 3), cl = lane & 7;
        const int base = g0 ? bm_ : bn_, lim = g0 ? M : N;
        const uint32_t ldbytes = (uint32_t)((g0 ? lda : ldb) * 2);
#pragma unroll
        for (int i = 0; i < P::NPMAX; ++i) {
            const int r = i * 32 + rl;
            const int gr = min(base + r, lim - 1);  // edge tiles re-read the last row (never stored)
            goff[i] = (uint32_t)gr * ldbytes + (uint32_t)((cl ^ ((r >> 1) & 7)) << 4);
        }
    };
    set_tile(bm, bn);
    const int nt = (K / 64) / SPLITK;     // K-tiles of this workgroup: slice * nt .. slice * nt + nt - 1
    auto issue_a = [&](int t, int stage) {  // group 0: this wave's rows of the A operand of K-tile t
        const char* g = (const char*)A + (size_t)(slice * nt + t) * P::ROWB;
        char* l = smem + stage * P::A_BYTES + w4 * 8 * P::ROWB;  // wave-uniform; HW adds lane*16
#pragma unroll
        for (int i = 0; i < P::NPA; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(g + goff[i]), (lds_void*)(l + i * 32 * P::ROWB), 16, 0, 0);
    };
    auto issue_b = [&](int t, int stage) {  // group 1: this wave's rows of the B operand of K-tile t
        const char* g = (const char*)B + (size_t)(slice * nt + t) * P::ROWB;
        char* l = smem + P::B_BASE + stage * P::B_BYTES + w4 * 8 * P::ROWB;
#pragma unroll
        for (int i = 0; i < P::NPB; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(g + goff[i]), (lds_void*)(l + i * 32 * P::ROWB), 16, 0, 0);
    };

    // ---- fragment reads: byte address row*128 + ((k-chunk) ^ ((row>>1)&7))*16; steps of 16 rows keep the swizzle term, so they
    // are immediate offsets of the ds_read
    const uint32_t sw = (uint32_t)((frow >> 1) & 7);
    const uint32_t arow = (uint32_t)((wr * P::WM + frow) * P::ROWB);
    const uint32_t brow = (uint32_t)(P::B_BASE + (wc * 64 + frow) * P::ROWB);
    u32x4 fa[2][MI], fb[2][4];  // [k-half][16-row block]: the whole K-tile's operands of this wave
    auto read_frags = [&](int sa, int sb) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const uint32_t ch = ((((uint32_t)(kk * 4 + fq)) ^ sw) << 4);
            const uint32_t ba = lds0 + sb * P::B_BYTES + brow + ch;
            const uint32_t aa = lds0 + sa * P::A_BYTES + arow + ch;
            fb[kk][0] = pp_lds_read16<0 * 2048>(ba);
            fb[kk][1] = pp_lds_read16<1 * 2048>(ba);
            fb[kk][2] = pp_lds_read16<2 * 2048>(ba);
            fb[kk][3] = pp_lds_read16<3 * 2048>(ba);
            static_for<0, MI>([&](auto mi) { fa[kk][mi.value] = pp_lds_read16<mi.value * 2048>(aa); });
        }
    };

    // (Measured and not kept, round 3: the wave's DMA pieces interleaved with its fragment reads - piece, 3 reads, piece, ... - instead
    // of all pieces, then all reads: 256-wide tiles unchanged (+-0.5 %), 128-wide 4-6 % slower.  Group 1's pieces at the head of its
    // compute phase, or reads before pieces: 3-8 % slower.  profiles/r03_c_gemm_pp_ab.txt.)

    f32x4 acc[4][MI];  // [ni][mi]
    auto compute = [&]() {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[ni][mi] = Mfma<T>::run(__builtin_bit_cast(frag, fb[kk][ni]), __builtin_bit_cast(frag, fa[kk][mi]), acc[ni][mi]);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto next3 = [](int s) { return s == 2 ? 0 : s + 1; };

    // the weight matrix on its way into the Infinity Cache (first 256 workgroups of the launch; gemm_common.h).  Its dump area is
    // B stage 2, first filled in phase 0: every wave has drained these loads by then (the prologue's counted wait is in issue order).
    static_assert(8 * 256 <= P::B_BYTES, "dump area inside B stage 2");
    touch_weight(B, N, K, ldb, blockIdx.x, min((int)gridDim.x, 256), tid, 512, smem + P::B_BASE + 2 * P::B_BYTES + wave * 256);
    // (Round 2 also warmed the backward epilogues' saved-tensor tile here, one dword per 128-byte line.  Round 3 A/B of exactly that,
    // same process, interleaved, everything cold - tools/ab_two_libs.py, M=8192 N=4096 K=1024 GeLU' epilogue: 95.2 us with the warming
    // loads, 87.8 without; whole step 20.13 -> 19.99 ms.  The 1024 scattered line fetches per workgroup delay its first K-tiles by
    // more than the epilogue's streaming loads gain, and the counters showed the tensor fetched twice: 232 MiB against 88.  Removed.)
    // ---- prologue: K-tiles 0 and 1 (A by group 0, B by group 1); tile 0 visible after the barrier
    if (g0) issue_a(0, 0); else issue_b(0, 0);
    for (;;) {  // one output tile per trip
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nt < 2) {
        PP_WAIT(0, 15);
    } else if (g0) {
        issue_a(1, 1);
        PP_WAIT(P::NPA, 15);
    } else {
        issue_b(1, 1);
        PP_WAIT(P::NPB, 15);
    }
    pp_barrier();
    if (g0) {
        // ================= group 0: compute tile t in phase 2t; in phase 2t+1 fetch A of tile t+2 and read tile t+1
        read_frags(0, 0);
        PP_WAIT(63, 0);
        int sb = 0;  // B stage of tile t
        for (int t = 0; t < nt; ++t) {
            compute();
            // A of tile t+1 (issued in phase 2t-1, or in the prologue) has landed: the barrier publishes it to group 1, which reads
            // it in phase 2t+2
            PP_WAIT(0, 15);
            pp_barrier();                                   // end of phase 2t
            if (t + 1 >= nt) break;                         // nothing left to load: straight to the epilogue (LDS is free:
                                                            // group 1 finished its last fragment reads before this barrier)
            sb = next3(sb);
            if (!(DBG & 1)) if (t + 2 < nt) issue_a(t + 2, t & 1);          // A stage t&1: tile t was last read in phase 2t (group 1), done
            if (!(DBG & 2)) read_frags((t + 1) & 1, sb);                    // tile t+1: its B was published by group 1's wait in phase 2t
            PP_WAIT(63, 0);                                 // fragments in registers; the DMA just issued stays in flight
            pp_barrier();                                   // end of phase 2t+1
        }
    } else {
        // ================= group 1: in phase 2t fetch B of tile t+2 and read tile t; compute it in phase 2t+1
        int sbr = 0, sbi = 2;  // B stages of tile t and of tile t+2
        for (int t = 0; t < nt; ++t) {
            // B stage of tile t+2 = that of tile t-1: last read in phase 2t-2 (this group) and 2t-3 (group 0)
            const bool more = !(DBG & 1) && t + 2 < nt;
            if (more) issue_b(t + 2, sbi);
            if (!(DBG & 2) || t == 0) read_frags(t & 1, sbr);
            // fragments in registers, and B of tile t+1 (issued one K-tile ago, or in the prologue) landed - group 0 reads it right
            // after the barrier; the batch just issued stays in flight for two more phases
            if (more) PP_WAIT(P::NPB, 0); else PP_WAIT(0, 0);
            pp_barrier();                                   // end of phase 2t
            compute();
            if (t + 1 < nt) pp_barrier();                   // end of phase 2t+1
            sbr = next3(sbr), sbi = next3(sbi);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    if (DBG & 4) {  // keep the accumulators alive without storing anything real
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) sum += acc[i][j][0] + acc[i][j][3];
        if (sum == 12345.678f) C[0] = (u16)lane;
        tidx += (int)gridDim.x;
        if (tidx >= tiles_m * tiles_n) return;
        tile_coords(tidx, tiles_m, tiles_n, group_m, tm, tn);
        bm = tm * P::BM, bn = tn * BN;
        set_tile(bm, bn);
        if (g0) issue_a(0, 0); else issue_b(0, 0);
        __syncthreads();
        continue;
    }
    if constexpr (SPLITK == 2) {
        // ---- split-K exchange: wave row `slice` is finished here, wave row 1 - slice goes to the partner
        const size_t slab = (size_t)128 * 64;
        if (wr != slice) {
            // write-through (sc1) 16-byte stores: the partials leave this XCD's L2 as they are written, so the publish needs no
            // agent-scope release (an L2 write-back of 128 KiB per workgroup: MI355X_MICROARCH.md, publish-large 8.2 vs 3.0 us)
            float* dstp = sp.slabs + (((size_t)tidx * 2 + (1 - slice)) * 4 + wc) * slab;
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(dstp, 0, (int)(slab * sizeof(float)), 0x00020000);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[ni][mi]), rsrc,
                                                           ((ni * MI + mi) * 64 + lane) * 16, 0, 16 /* sc1 */);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains before the barrier
        __syncthreads();
        unsigned* lost = reinterpret_cast<unsigned*>(smem + P::LDS - 4);  // last word of B stage 2: beyond the epilogue's staging area
        if (tid == 0) {
            __hip_atomic_store(sp.flags + tidx * 2 + (1 - slice), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0, gave_up = 0;
            while (__hip_atomic_load(sp.flags + tidx * 2 + slice, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1u << 24)) {  // ~seconds: the partner never ran - leave a mark instead of hanging the GPU
                    __hip_atomic_store(sp.flags + 2 * tiles_m * tiles_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gave_up = 1;
                    break;
                }
            }
            *lost = gave_up;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (wr != slice) return;
        const bool poisoned = *lost != 0u;  // the partner's partials never came: the half tile is written as NaN, never as a wrong number
        const f32x4* src = reinterpret_cast<const f32x4*>(sp.slabs + (((size_t)tidx * 2 + slice) * 4 + wc) * slab) + lane;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] += src[(ni * MI + mi) * 64];
        if (poisoned) {
            const float qnan = __builtin_nanf("");
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{qnan, qnan, qnan, qnan};
        }
    }
    // ---- epilogue (coalesced through a wave-private LDS region).  Group 0 starts it one phase before group 1, whose last phase
    // computes from registers; nobody reads the rings any more and every DMA into them has been waited for.
    if constexpr (SPLITK == 2) {
        static_assert(SPLITK != 2 || 8 * MI * 16 * 128 <= P::LDS - 4, "split-K epilogue staging below the `lost` word");
        epilogue_staged<T, EPI, RES, MI>(acc, smem + wave * (MI * 16 * 128), bm + wr * P::WM, bn + wc * 64, M, N, ldc, C, aux, bias, Rsd,
                                         lane, ex);
        return;
    } else {
        // the next tile of this workgroup: its K-tile 0 goes into A stage 0 / B stage 0 now and lands while the epilogue runs in
        // B stages 1-2
        tidx += (int)gridDim.x;
        const bool more = tidx < tiles_m * tiles_n;
        int nbm = 0, nbn = 0;
        if (more) {
            tile_coords(tidx, tiles_m, tiles_n, group_m, tm, tn);
            nbm = tm * P::BM, nbn = tn * BN;
            set_tile(nbm, nbn);
            if (g0) issue_a(0, 0); else issue_b(0, 0);
        }
        constexpr int MH = MI / 2;
        static_assert(8 * MH * 16 * 128 <= 2 * P::B_BYTES, "the two-pass epilogue stages inside B stages 1-2");
        // the epilogue's per-lane addressing is derived from an opaque copy of the thread id: otherwise the compiler hoists it out of
        // the tile loop, keeps it live across the K loop and spills
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        const int wave_e = tid_e >> 6, lane_e = tid_e & 63;
        const int wr_e = wave_e / P::WAVES_N, wc_e = wave_e % P::WAVES_N;
        char* wl = smem + P::B_BASE + P::B_BYTES + wave_e * (MH * 16 * 128);
        epilogue_staged<T, EPI, RES, MI, 0, MH>(acc, wl, bm + wr_e * P::WM, bn + wc_e * 64, M, N, ldc, C, aux, bias, Rsd, lane_e, ex);
        epilogue_staged<T, EPI, RES, MI, MH, MH>(acc, wl, bm + wr_e * P::WM + MH * 16, bn + wc_e * 64, M, N, ldc, C, aux, bias, Rsd, lane_e,
                                                 ex);
        if (!more) return;
        bm = nbm, bn = nbn;
        __syncthreads();  // every wave's staging reads are done: B stage 1 takes K-tile 1 of the next tile
    }
    }
}

template <typename T, int EPI, int BN>
static int launch_pp(void* C, void* aux, const void* A, const void* B, const void* bias, const void* residual, int M, int N, int K,
                     int lda, int ldb, int ldc, hipStream_t st, const EpiExtra& ex) {
    typedef PP<BN> P;
    const int tiles_m = (M + P::BM - 1) / P::BM, tiles_n = (N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int group_m = (nwg / 8) / tiles_n;  // one XCD's run of tiles spans whole tile-rows when it can
    group_m = group_m < 1 ? 1 : (group_m > 8 ? 8 : group_m);
    // persistent: at most one workgroup per CU, each walks its tiles.  mm_set_cu_share (include/mm_hip.h): a caller that runs two
    // of these launches on two streams gives each half the CUs - measured on the 410M step (QKV + rotary beside h->4h + GeLU, GeLU'
    // data gradient beside the attention backward), same box, alternating: 20.34 -> 19.40 ms.  Alone on 128 CUs the h->4h launch
    // takes 4 tiles x 25 us instead of 2 x 34.5: with half the CUs in their epilogue at any time the output burst no longer runs
    // into the HBM write rate (profiles/r03_c_gemm_pp_ab.txt, sections 8-9).  Multiples of 8 keep a workgroup's tiles on one XCD run.
    int cap = 256;
    const int share = cu_share() & ~7;
    if (BN == 256 && nwg > 256 && share >= 8 && share < 256) cap = share;
    dim3 grid(nwg > cap ? cap : nwg), block(512);
    // (the rotary epilogue carries its tables in the aux / residual slots: never a residual add)
    if (EPI != MM_EPI_BIAS_ROTARY && (epi_is_bwd(EPI) || residual)) {
        if constexpr (EPI != MM_EPI_BIAS_ROTARY)
            gemm_nt_pp_k<T, EPI, true, BN><<<grid, block, 0, st>>>((u16*)C, (u16*)aux, (const u16*)A, (const u16*)B, (const u16*)bias,
                                                                   (const u16*)residual, M, N, K, lda, ldb, ldc, tiles_m, tiles_n, group_m, ex);
    } else if constexpr (!epi_is_bwd(EPI)) {
        gemm_nt_pp_k<T, EPI, false, BN><<<grid, block, 0, st>>>((u16*)C, (u16*)aux, (const u16*)A, (const u16*)B, (const u16*)bias,
                                                                (const u16*)residual, M, N, K, lda, ldb, ldc, tiles_m, tiles_n, group_m, ex);
    }
    MM_CHECK_LAUNCH();
    return MM_OK;
}

template <typename T, int BN>
static int launch_pp_epi(void* C, void* aux, const void* A, const void* B, const void* bias, const void* residual, int M, int N,
                         int K, int lda, int ldb, int ldc, int epilogue, hipStream_t st, const EpiExtra& ex) {
    switch (epilogue) {
        case MM_EPI_BIAS_ROTARY: return launch_pp<T, MM_EPI_BIAS_ROTARY, BN>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_NONE: return launch_pp<T, MM_EPI_NONE, BN>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_BIAS: return launch_pp<T, MM_EPI_BIAS, BN>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_BIAS_GELU_TANH: return launch_pp<T, MM_EPI_BIAS_GELU_TANH, BN>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_BIAS_GELU_ERF: return launch_pp<T, MM_EPI_BIAS_GELU_ERF, BN>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_BIAS_RELU: return launch_pp<T, MM_EPI_BIAS_RELU, BN>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_RELU_BWD: return launch_pp<T, MM_EPI_RELU_BWD, BN>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        case MM_EPI_GELU_TANH_BWD: return launch_pp<T, MM_EPI_GELU_TANH_BWD, BN>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
        default: return launch_pp<T, MM_EPI_GELU_ERF_BWD, BN>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, st, ex);
    }
}

// ---- split-K (two K-slices per 256 x 256 tile): shape policy, workspace size, launch
__global__ void pp_zero_flags_k(unsigned* __restrict__ p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}

static size_t pp_splitk_flag_bytes(int tiles) { return (((size_t)tiles * 2 + 1) * 4 + 255) & ~(size_t)255; }

// 0 = not a split-K shape.  Wanted where 256 x 256 tiles are too few for the 256 CUs but two slices each fill them (every
// workgroup resident at once: the partners wait for each other) and each slice still has a K loop worth its prologue.
size_t gemm_pp_splitk_workspace(int M, int N, int K, int lda, int ldb) {
    // K >= 8192: measured on MI355X at M = 8192, N = 1024 (tools/ab_splitk.py, cold operands) the exchange (32 MiB of fp32 partials
    // out and in, all 256 workgroups at the same moment: ~13 us) costs more than the 256 x 256 tiles gain at K = 2048 / 3072 / 4096
    // (42.5 / 57.3 / 72.9 us against 36.3 / 54.5 / 70.3 for the 128 x 128 kernel) and far less at K = 50304 (659 against 865 us)
#ifndef MM_AB_SPLITK_MINK  // (A/B variant builds only: make EXTRA=-DMM_AB_SPLITK_MINK=2048)
#define MM_AB_SPLITK_MINK 8192
#endif
    if (M % 256 || N % 256 || K % 128 || K < MM_AB_SPLITK_MINK) return 0;
    if ((size_t)M * (size_t)lda * 2 >= (1ull << 32) || (size_t)N * (size_t)ldb * 2 >= (1ull << 32)) return 0;
    const long tiles = (long)(M / 256) * (N / 256);
    if (tiles % 8 || tiles < 64 || tiles > 128) return 0;
    return pp_splitk_flag_bytes((int)tiles) + (size_t)tiles * 2 * 4 * 128 * 64 * sizeof(float);
}

template <typename T, int EPI>
static int launch_pp_splitk(void* C, const void* A, const void* B, const void* bias, int M, int N, int K, int lda, int ldb, int ldc,
                            void* ws, hipStream_t st) {
    const int tiles_m = M / 256, tiles_n = N / 256, tiles = tiles_m * tiles_n;
    int group_m = (tiles / 8) / tiles_n;
    group_m = group_m < 1 ? 1 : (group_m > 8 ? 8 : group_m);
    unsigned* flags = (unsigned*)ws;
    // a kernel, not hipMemsetAsync: under stream capture the memset was not replayed (see prefix.hip)
    pp_zero_flags_k<<<1, 256, 0, st>>>(flags, 2 * tiles + 1);
    MM_CHECK_LAUNCH();
    SplitArgs sp{(float*)((char*)ws + pp_splitk_flag_bytes(tiles)), flags};
    gemm_nt_pp_k<T, EPI, false, 256, 0, 2><<<dim3(2 * tiles), dim3(512), 0, st>>>((u16*)C, nullptr, (const u16*)A, (const u16*)B,
                                                                                 (const u16*)bias, nullptr, M, N, K, lda, ldb, ldc,
                                                                                 tiles_m, tiles_n, group_m, EpiExtra{}, sp);
    MM_CHECK_LAUNCH();
    return MM_OK;
}

int gemm_pp_splitk(void* C, const void* A, const void* B, const void* bias, int M, int N, int K, int lda, int ldb, int ldc,
                   void* ws, int epilogue, int dtype, hipStream_t st) {
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        (void)epilogue;  // MM_EPI_NONE only (the bias epilogue's extra registers spill beside the exchange: not instantiated)
        return launch_pp_splitk<T, MM_EPI_NONE>(C, A, B, bias, M, N, K, lda, ldb, ldc, ws, st);
    });
}

#ifdef MM_DEV
template <int EPI, int DBG>
static void launch_pp_dbg(void* C, void* aux, const void* A, const void* B, const void* bias, int M, int N, int K, hipStream_t st) {
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    const int nwg = tiles_m * tiles_n;
    int group_m = (nwg / 8) / tiles_n;
    group_m = group_m < 1 ? 1 : (group_m > 8 ? 8 : group_m);
    gemm_nt_pp_k<BF16, EPI, false, 256, DBG><<<dim3(nwg > 256 ? 256 : nwg), dim3(512), 0, st>>>((u16*)C, (u16*)aux, (const u16*)A, (const u16*)B, (const u16*)bias,
                                                                            nullptr, M, N, K, K, K, N, tiles_m, tiles_n, group_m, EpiExtra{});
}

// Developer decomposition of the 256x256 ping-pong kernel (bf16, contiguous operands, M, N multiples of 256, K of 64).
int gemm_pp_debug(void* C, void* aux, const void* A, const void* B, const void* bias, int M, int N, int K, int gelu, int dbg,
                  hipStream_t st) {
    if (M % 256 || N % 256 || K % 64 || K < 128) return MM_EINVAL;
#define MM_DBG_CASE(D)                                                                          \
    case D:                                                                                      \
        if (gelu) launch_pp_dbg<MM_EPI_BIAS_GELU_TANH, D>(C, aux, A, B, bias, M, N, K, st);     \
        else launch_pp_dbg<MM_EPI_NONE, D>(C, aux, A, B, bias, M, N, K, st);                    \
        break;
    switch (dbg) {
        MM_DBG_CASE(0) MM_DBG_CASE(1) MM_DBG_CASE(2) MM_DBG_CASE(3) MM_DBG_CASE(4) MM_DBG_CASE(5) MM_DBG_CASE(6) MM_DBG_CASE(7)
        default: return MM_EINVAL;
    }
#undef MM_DBG_CASE
    return hipGetLastError() == hipSuccess ? MM_OK : MM_ELAUNCH;
}
#endif  // MM_DEV

// Shape policy (0 = this kernel does not take the problem, else the tile width BN).  MM_GEMMPP (developer library only):
// 256 / 128 force that tile when legal, -1 = never.
int gemm_pp_tile(int M, int N, int K, int lda, int ldb) {
    int force = 0;
#ifdef MM_DEV  // developer library only (make DEV=1 -> libmm_hip_dev.so): the product library has ONE dispatch policy
    if (const char* e = getenv("MM_GEMMPP")) force = atoi(e);
#endif
    if (force < 0 || M < 256 || N < 128 || K < 128) return 0;
    // staging uses 32-bit byte offsets from the operand base
    if ((size_t)M * (size_t)lda * 2 >= (1ull << 32) || (size_t)N * (size_t)ldb * 2 >= (1ull << 32)) return 0;
    if (force == 256 || force == 128) return (force == 256 && N < 256) ? 128 : force;
    // Measured at M = 8192 on one MI355X (us; tools/ab_gemm.py, same process): N=4096 K=1024 bias-GeLU: 128x128 kernel 87-93,
    // gemm256.hip 78-80, this kernel 256x256 68-70, 256x128 88-98; N=3072 K=1024: 52.5 / 57 / 50.6 / 55.8; N=1024 K=4096:
    // 62 / 88 / 75 / 66.5; N=1024 K=1024: 19 / 28 / 24 / 20; N=50304 K=1024: 986 / 807 / 806 / 1116.  So: 256x256 tiles from
    // 1.5 rounds of the 256 CUs up; the N = h class (128 tiles of 256x256) stays on the 128x128 kernel, which two workgroups
    // per CU keep ahead of the 256x128 variant here.
    // Other widths (h = 512, 2304 / TP 2, 4096 / TP 8; same tool, custom shapes): the 256x256 tile also wins at 192 and 256
    // tiles (N=1536 K=4096: 84 vs 102 us on the 128x128 kernel; N=2048 K=512: 21 vs 24) and loses when the last round is
    // mostly empty (288 tiles, N=2304 K=4608: 191 vs 180).  So: at least 3/4 of a round, and rounds at least 70 % full.
    // Round 3, with the separate A / B rings and the persistent tile walk (same tool): the 256x128 form now beats the 128x128 kernel
    // on the N = h class - M=8192 N=1024: K=4096 57.8-61.5 vs 69.7-70.2 us, K=3072 44.2 vs 50.2-51.1, K=1024 19.2-19.4 vs 21.1-22.4 -
    // so it takes what the 256x256 tile leaves, under the same fill rule.
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    const long rounds = (t256 + 255) / 256;
    if (t256 >= 192 && 10 * t256 >= 7 * 256 * rounds) return 256;
    const long t128 = (long)((M + 255) / 256) * ((N + 127) / 128);
    const long rounds128 = (t128 + 255) / 256;
    return (t128 >= 192 && 10 * t128 >= 7 * 256 * rounds128) ? 128 : 0;
}

int gemm_pp_try(void* C, void* aux, const void* A, const void* B, const void* bias, const void* residual, int M, int N, int K,
                int lda, int ldb, int ldc, int epilogue, int dtype, hipStream_t st, const EpiExtra& ex) {
    const int bn = gemm_pp_tile(M, N, K, lda, ldb);
    if (bn == 0) return 1;
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        if (bn == 256) return launch_pp_epi<T, 256>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, epilogue, st, ex);
        return launch_pp_epi<T, 128>(C, aux, A, B, bias, residual, M, N, K, lda, ldb, ldc, epilogue, st, ex);
    });
}

}  // namespace mm
