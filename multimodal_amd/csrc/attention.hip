// attention.hip - fused (flash-style) self-attention forward and backward for gfx950.
//
// Replaces ParallelSelfAttention.attention + the permute behind it (reference
// megatron/model/transformer.py:315-416, 623-630): scores = q.k^T / sqrt(hn), mask, softmax, P.V -
// without ever materialising the [b*np, s, s] score tensor (512 MiB per layer at the 410M shape).
// Reads q/k/v straight out of the fused QKV activation [s, b, np, 3*hn] and writes the context in
// the layer's [s, b, np*hn] layout, so the split / transpose / permute copies disappear as well.
//
// MI355X structure (all three kernels share it):
//   * workgroup = 4 waves; a wave owns 32 rows of the "resident" operand (32 queries in fwd / dQ,
//     32 keys in dK/dV) whose fragments live in registers for the whole kernel;
//   * the streamed operand arrives in 64-row tiles, global -> registers -> LDS (issue early, write late,
//     double-buffered LDS, one barrier per tile);
//   * MFMA 32x32x16 with the resident rows on the LANE index of every score tile ("swapped QK^T"):
//     the softmax row statistics are then lane-local (one cross-half exchange), and the score tile's
//     accumulator registers are directly the B operand of the second product (P.V, dS.K, P^T.dO, dS^T.Q)
//     with no LDS round trip;
//   * the A operand of those second products needs the streamed tile transposed: it is read from the SAME
//     row-major LDS image with ds_read_b64_tr_b16.  An XOR on the 16-byte chunk index (head size 64: unpadded 128-byte
//     rows; head size 128: rows padded by 64 bytes; mfma32.h) makes both the row reads (ds_read_b128) and the transposed
//     reads conflict-free;
//   * softmax in the exp2 domain with fp32 statistics; fp32 accumulation everywhere.
// mask_mode 0 = implicit causal, -inf above the diagonal, fully masked key tiles are skipped
//               (semantics of scaled_upper_triang_masked_softmax.h:143-260);
// mask_mode 1 = explicit byte mask, 1 -> score := -10000 (semantics of scaled_masked_softmax.h:180-184 and of
//               the eager gpt2_attention_mask_func path).
#include "common.h"
#include "mfma32.h"

namespace mm {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kMaskFill = -10000.0f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Make a register-resident operand "used" here: the compiler must finish its global load before this point instead of
// sinking the load to the first MFMA inside the tile loop (where its pending vmcnt would serialise every iteration's
// prefetch behind the MFMAs).
template <typename V> __device__ __forceinline__ void pin(V& v) {
    u32x4 t;
    __builtin_memcpy(&t, &v, 16);
    asm volatile("" : "+v"(t));
    __builtin_memcpy(&v, &t, 16);
}
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }

// Two workgroups share a CU, so every SIMD holds two waves running this same loop.  Left to age-based arbitration the wave in
// its softmax segment keeps the issue port and the partner's MFMAs trickle in behind it (PMC: matrix pipe and vector port
// busy 27 % + 45 % with no overlap).  Raising the priority of a wave for the length of its MFMA groups lets each MFMA take
// its one issue slot per 32 pipe cycles and leaves the rest of the port to the partner's vector work: forward -5.5 %,
// backward -3.7 % (tools/ab_attn.py, same process), results unchanged.
__device__ __forceinline__ void mfma_prio(int on) {
    if (on) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}
// the same on top of a per-workgroup base priority (heavy causal blocks 2, light ones 0): variant builds only
__device__ __forceinline__ void mfma_prio_b(int on, bool heavy) {
    if (heavy) {
        if (on) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(2);
    } else {
        if (on) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
}

// Combine a value with the one held by the lane 32 away (the other half of the wave) without the LDS round trip of
// ds_bpermute: v_permlane32_swap exchanges the upper 32 lanes of its first operand with the lower 32 of its second, so with
// both operands = x every lane ends up holding (lower-half value, upper-half value) of its lane pair.
__device__ __forceinline__ void half_pair(float x, float& lo, float& hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float half_max(float x) {
    float lo, hi;
    half_pair(x, lo, hi);
    return fmaxf(lo, hi);
}
__device__ __forceinline__ float half_sum(float x) {
    float lo, hi;
    half_pair(x, lo, hi);
    return lo + hi;
}

// ---- attention dropout (reference megatron/model/transformer.py:301-302, 383-384: nn.Dropout on the softmax output under the
// model-parallel rng tracker's fork).  The keep decision of probability element (batch*head, q, key) is a pure function of the
// tracker's device-resident Philox-style state {seed, base offset}, the launch's position `pos` inside the step and the element's
// coordinates, so the forward and both backward kernels regenerate the same mask and nothing of size s x s is stored:
//     key64 = mix64(seed, offset + pos, batch * np + head)                       (per workgroup)
//     h     = hash32(hash32((q * ceil(s / 2) + (key >> 1)) ^ key64.lo) + key64.hi) (one keyed 32-bit hash per key PAIR)
//     keep  = ((key & 1) ? h >> 16 : h & 0xffff) >= thr,  thr = round(p * 65536)  (p is quantised to 1/65536)
// and kept probabilities are scaled by 1 / (1 - thr / 65536).  mm_attn_dropout_mask writes the same mask as bytes (the tests'
// oracle applies it explicitly).  The reference draws its mask from the device generator; the two streams cannot match
// element for element (parity: distributional - mean, independence across heads / calls - and exact against this definition).
struct DropArgs {
    const unsigned long long* st;  // {seed, base offset} on the device (mpu/random.py draw_device)
    unsigned long long pos;        // this launch's position inside the step
    uint32_t thr;                  // drop when the 16 random bits are below thr
    float rscale;                  // 1 / (1 - thr / 65536)
};
__device__ __forceinline__ uint32_t drop_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// 64-bit key of a (seed, offset + pos, batch * np + head) triple.  Both halves enter the per-element hash at different rounds
// (below), so two launches / heads whose keys differ in ANY bit draw unrelated masks: with a single 32-bit key XOR-ed into the
// index every mask would be a relabelled view of one 2^32-entry table, and a 32-bit birthday collision (expected after ~65k
// (head, launch) pairs = ~50 steps of the 410M shape) would repeat a whole mask.
struct DropKey {
    uint32_t lo, hi;
};
__device__ __forceinline__ DropKey drop_key(const DropArgs& da, uint32_t bh) {
    unsigned long long z = da.st[0] + 0x9E3779B97F4A7C15ull * (da.st[1] + da.pos + 1ull);
    z ^= (unsigned long long)(bh + 1u) * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return DropKey{(uint32_t)z, (uint32_t)(z >> 32)};
}
// the 32 random bits of key pair (key >> 1) of query row q; qh = q * ceil(s / 2): a keyed two-round hash of the pair's index
__device__ __forceinline__ uint32_t drop_bits(DropKey k, uint32_t qh, int key) {
    return drop_hash(drop_hash((qh + ((uint32_t)key >> 1)) ^ k.lo) + k.hi);
}
__device__ __forceinline__ bool drop_keep(DropKey k, uint32_t qh, int key, uint32_t thr) {
    const uint32_t h = drop_bits(k, qh, key);
    return ((key & 1) ? (h >> 16) : (h & 0xffffu)) >= thr;
}
// zero the dropped ones of 16 accumulator registers whose rows are KEYS kb0 + acc_row(i, hh) of one query (forward / dQ layout:
// registers 4g .. 4g+3 are four consecutive keys = two key pairs = two hashes)
__device__ __forceinline__ void drop_apply_keys(f32x16& v, DropKey key32, uint32_t qh, int kb0, int hh, uint32_t thr) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int kb = kb0 + acc_row(4 * g, hh);
        const uint32_t h0 = drop_bits(key32, qh, kb), h1 = drop_bits(key32, qh, kb + 2);
        if ((h0 & 0xffffu) < thr) v[4 * g] = 0.f;
        if ((h0 >> 16) < thr) v[4 * g + 1] = 0.f;
        if ((h1 & 0xffffu) < thr) v[4 * g + 2] = 0.f;
        if ((h1 >> 16) < thr) v[4 * g + 3] = 0.f;
    }
}

// mask byte for (q, key); mask row base already selected
__device__ __forceinline__ bool mask_at(const uint8_t* __restrict__ mrow, int key) { return mrow[key] == 1; }

// =====================================================================================  forward
template <typename T, int HD, int MODE, bool DROP = false>
__global__ __launch_bounds__(256, (HD == 64 ? 2 : 1)) void attn_fwd_k(u16* __restrict__ ctx, float* __restrict__ lse, const u16* __restrict__ qkv,
                                                  const uint8_t* __restrict__ mask, int mask_batches, int s, int b, int np,
                                                  float scale_log2e, DropArgs da = DropArgs{}) {
    typedef Tile<HD> TL;
    typedef typename M32<T>::frag frag;
    constexpr int NDS = HD / 16, NDB = HD / 32;
    __shared__ __attribute__((aligned(16))) char smem[4 * TL::BYTES];  // [buf][K | V]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int bi = blockIdx.x / np, hi = blockIdx.x % np;  // head-fast grid: see launch_fwd
    // Workgroups of the heavier half of the causal triangle (grid y = 0 is the heaviest block) run at base priority 2, the lighter
    // half at 0, the MFMA groups one above the base (mfma_prio_b): the makespan of the launch is the longest workgroup's
    // chain of tiles, and that chain slows down threefold when its waves share their SIMDs evenly with two light workgroups.
    // Same-box A/B (round 3): 54.2 -> 52.7 us.  (Pairing block y with block nblk-1-y in one workgroup - equal work everywhere,
    // 512 workgroups = two per CU - measured 57.6 us: the third wave per SIMD is worth more than the balance.)
    const bool heavy = MODE == 0 && (int)blockIdx.y * 2 < (int)gridDim.y;
#define FWD_PRIO(x) mfma_prio_b(x, heavy)
    FWD_PRIO(0);
    const int q0 = ((int)gridDim.y - 1 - (int)blockIdx.y) * 128;  // heaviest (latest) query blocks first
    const size_t tok = (size_t)b * np * 3 * HD;                   // elements between sequence positions
    const u16* qbase = qkv + ((size_t)bi * np + hi) * 3 * HD;
    const u16* kbase = qbase + HD;
    const u16* vbase = qbase + 2 * HD;

    const int qrow = q0 + wave * 32 + r;
    const int qrow_c = qrow < s ? qrow : s - 1;
    frag qf[NDS];
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) qf[ds] = *reinterpret_cast<const frag*>(qbase + (size_t)qrow_c * tok + 16 * ds + 8 * hh);
    DropKey dkey{0, 0};
    uint32_t dqh = 0;
    if constexpr (DROP) {
        dkey = drop_key(da, (uint32_t)blockIdx.x);  // blockIdx.x = batch * np + head
        dqh = (uint32_t)qrow_c * (uint32_t)((s + 1) >> 1);
    }

    const int blk_last = min(q0 + 127, s - 1);
#if defined(MM_AB_FWD_DBG) && (MM_AB_FWD_DBG & 8)  // decomposition build: every block walks the AVERAGE number of tiles, unmasked
    const int ntiles = (s / 128 + 1);
    const int wave_last = s - 1;
    const int wave_first = s;
#else
    const int ntiles = (MODE == 0) ? (blk_last / 64 + 1) : ((s + 63) / 64);
    const int wave_last = min(q0 + wave * 32 + 31, s - 1);
    const int wave_first = q0 + wave * 32;
#endif
    const float mask_raw = kMaskFill * kLog2e / scale_log2e;  // raw score whose scaled value is -10000
    const uint8_t* mrow = nullptr;
    if (MODE == 1) mrow = mask + ((size_t)(mask_batches == 1 ? 0 : bi) * s + qrow_c) * s;

    f32x16 o[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    u32x4 kreg[TL::NCH], vreg[TL::NCH];
    tile_load<HD>(kreg, kbase, tok, 0, s, tid);
    tile_load<HD>(vreg, vbase, tok, 0, s, tid);
    tile_store<HD>(smem, kreg, tid);
    tile_store<HD>(smem + TL::BYTES, vreg, tid);
    __syncthreads();
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) pin(qf[ds]);

#ifndef MM_AB_FWD_DBG  // decomposition builds (wrong results, same addresses): 1 = no barrier in the loop, 2 = no tile fetch / LDS
#define MM_AB_FWD_DBG 0  // store in the loop, 4 = no softmax arithmetic (P = S)
#endif
    constexpr int DBG = MM_AB_FWD_DBG;
    // The ring stage is a compile-time constant of each copy of the tile body (the loop walks two tiles per trip): every LDS
    // address is then a loop-invariant per-lane register plus an immediate, where a run-time stage costs 17 v_add_u32 per tile
    // and wave in a loop that the vector port bounds (in-kernel clock 1.98 GHz: 1350 cycles per tile and SIMD for ~1000 cycles of
    // vector + MFMA issue, section 11 of DESIGN.md).
    auto tile_step = [&](const int j, auto parity) {
        constexpr int cur = decltype(parity)::value;
        const char* ks = smem + cur * 2 * TL::BYTES;
        const char* vs = ks + TL::BYTES;
        if (!(DBG & 2) && j + 1 < ntiles) {
            tile_load<HD>(kreg, kbase, tok, (j + 1) * 64, s, tid);
            tile_load<HD>(vreg, vbase, tok, (j + 1) * 64, s, tid);
        }
        const int kv0 = j * 64;
        if (MODE == 1 || kv0 <= wave_last) {  // wave-uniform: skip key tiles entirely above this wave's diagonal
            f32x16 sacc[2];
#ifdef MM_AB_FWD_NO_HOIST  // (A/B variant builds only)
            constexpr bool HOIST = false;
#else
            constexpr bool HOIST = HD == 64;  // head size 128: the fragments of a whole tile do not fit beside three waves per SIMD
#endif
            frag vfr0[2][NDB];
            if constexpr (HOIST) {
                // every K fragment of the tile is requested before the first MFMA (the compiler's own order asks for each one
                // just ahead of its MFMA: four exposed LDS round trips per tile), and the V fragments of the first 32 keys are
                // requested before the softmax, which does not depend on them: 162 VGPRs, still three waves per SIMD;
                // 54.2 -> 53.4 us in the same A/B
                frag kfr[2][NDS];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ds = 0; ds < NDS; ++ds) kfr[t][ds] = row_frag<T, HD>(ks, 32 * t, ds, lane);
                __builtin_amdgcn_sched_barrier(0);
                FWD_PRIO(1);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[t][i] = 0.f;
#pragma unroll
                    for (int ds = 0; ds < NDS; ++ds) sacc[t] = M32<T>::run(kfr[t][ds], qf[ds], sacc[t]);
                }
                FWD_PRIO(0);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int d = 0; d < NDB; ++d) vfr0[s2][d] = tr_frag<T, HD>(vs, 16 * s2, d, lane);
                __builtin_amdgcn_sched_barrier(0);
            } else {
            FWD_PRIO(1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[t][i] = 0.f;
#pragma unroll
                for (int ds = 0; ds < NDS; ++ds) sacc[t] = M32<T>::run(row_frag<T, HD>(ks, 32 * t, ds, lane), qf[ds], sacc[t]);
            }
            FWD_PRIO(0);
            }
            // raw scores stay unscaled: max commutes with the positive scale, and p = exp2(fma(s, c, -m)) is one FMA + one exp.
            // The causal compare/select only runs on tiles that touch this wave's diagonal (wave-uniform test).
            if constexpr ((DBG & 4) == 0) {
            if (MODE == 1) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = kv0 + 32 * t + acc_row(i, hh);
                        if (key >= s) sacc[t][i] = -INFINITY;
                        else if (mask_at(mrow, key)) sacc[t][i] = mask_raw;
                    }
            } else if (kv0 + 63 > wave_first) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (kv0 + 32 * t + acc_row(i, hh) > qrow) sacc[t][i] = -INFINITY;
            }
            // v_max3_f32 by hand: fmaxf on MFMA outputs makes the compiler canonicalise every operand first (8 extra v_max per tile)
            float rmax = vmax3(sacc[0][0], sacc[0][1], sacc[0][2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) rmax = vmax3(rmax, sacc[0][i], sacc[0][i + 1]);
            rmax = vmax3(rmax, sacc[0][15], sacc[1][0]);
#pragma unroll
            for (int i = 1; i < 15; i += 2) rmax = vmax3(rmax, sacc[1][i], sacc[1][i + 1]);
            rmax = vmax2(rmax, sacc[1][15]);
            {
                float lo, hi;
                half_pair(rmax, lo, hi);
                rmax = vmax2(lo, hi);
            }
            const float m_new = vmax2(rmax * scale_log2e, m_run);
            // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32): two scores per VALU issue
            const f32x2 sc2 = {scale_log2e, scale_log2e}, mn2 = {m_new, m_new};
            f32x2 lsum2 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x2 e = f32x2{sacc[t][i], sacc[t][i + 1]} * sc2 - mn2;
                    const f32x2 p = {fast_exp2(e[0]), fast_exp2(e[1])};
                    sacc[t][i] = p[0];
                    sacc[t][i + 1] = p[1];
                    lsum2 += p;
                }
            const float lsum = lsum2[0] + lsum2[1];
            if (__any(m_new > m_run)) {  // the running max moved for some row of this wave: rescale (exact skip otherwise)
                const float alpha = fast_exp2(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < NDB; ++d)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
            }
            l_run += lsum;
            m_run = m_new;
            }
            if constexpr (DROP) {  // the row sum above is over ALL keys (softmax first, then dropout); P.V only sees the kept ones
#pragma unroll
                for (int t = 0; t < 2; ++t) drop_apply_keys(sacc[t], dkey, dqh, kv0 + 32 * t, hh, da.thr);
            }
            frag vfr1[2][NDB];
            if constexpr (HOIST) {  // the second 32 keys' V fragments: requested before the first P.V MFMA, not between them
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int d = 0; d < NDB; ++d) vfr1[s2][d] = tr_frag<T, HD>(vs, 32 + 16 * s2, d, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            FWD_PRIO(1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const frag pf = M32<T>::cvt8(sacc[t][8 * s2], sacc[t][8 * s2 + 1], sacc[t][8 * s2 + 2], sacc[t][8 * s2 + 3],
                                                 sacc[t][8 * s2 + 4], sacc[t][8 * s2 + 5], sacc[t][8 * s2 + 6], sacc[t][8 * s2 + 7]);
#pragma unroll
                    for (int d = 0; d < NDB; ++d)
                        o[d] = M32<T>::run(HOIST ? (t == 0 ? vfr0[s2][d] : vfr1[s2][d]) : tr_frag<T, HD>(vs, 32 * t + 16 * s2, d, lane), pf, o[d]);
                }
            FWD_PRIO(0);
        }
        if (!(DBG & 2) && j + 1 < ntiles) {
            char* nk = smem + (cur ^ 1) * 2 * TL::BYTES;
            tile_store<HD>(nk, kreg, tid);
            tile_store<HD>(nk + TL::BYTES, vreg, tid);
        }
        if (!(DBG & 1)) __syncthreads();
    };
    for (int j = 0; j < ntiles; j += 2) {
        tile_step(j, std::integral_constant<int, 0>{});
        if (j + 1 < ntiles) tile_step(j + 1, std::integral_constant<int, 1>{});
    }

    const float l_tot = half_sum(l_run);
    const float inv = (DROP ? da.rscale : 1.0f) / l_tot;
    if (qrow < s) {
        u16* orow = ctx + ((size_t)qrow * b + bi) * ((size_t)np * HD) + (size_t)hi * HD;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = 32 * d + 8 * g4 + 4 * hh;
                const u16 a0 = from_f32<T>(o[d][4 * g4] * inv), a1 = from_f32<T>(o[d][4 * g4 + 1] * inv);
                const u16 a2 = from_f32<T>(o[d][4 * g4 + 2] * inv), a3 = from_f32<T>(o[d][4 * g4 + 3] * inv);
                *reinterpret_cast<u32x2*>(orow + d0) = u32x2{(uint32_t)a0 | ((uint32_t)a1 << 16), (uint32_t)a2 | ((uint32_t)a3 << 16)};
            }
        if (hh == 0) lse[((size_t)bi * np + hi) * s + qrow] = (m_run + __log2f(l_tot)) * kLn2;
    }
#undef FWD_PRIO
}

// =====================================================================================  backward: gradient row store
// One lane's share of a [HD] gradient row in the 32x32 accumulator layout: register 4*g4 + e of block d is head dim
// 32*d + 8*g4 + 4*hh + e.  The row is scaled and rounded to T; with ROT > 0 the first ROT dims then go through the
// TRANSPOSED rotary rotation (the backward of apply_rotary_pos_emb, megatron/model/positional_embeddings.py:76-92, with
// the roundings of rotary_qk_k(inverse) in elementwise.hip): dim i < ROT/2 pairs with i + ROT/2, and since ROT/2 is a
// multiple of 8 both live in the SAME lane - the rotation is register-local.  cosr / sinr: the row's table lines [ROT].
template <typename T, int HD, int ROT>
__device__ __forceinline__ void store_grad_row(u16* __restrict__ dst, const f32x16 (&acc)[HD / 32], float scale, int hh,
                                               const u16* __restrict__ cosr, const u16* __restrict__ sinr) {
    // No fma contraction here: for fp16 the compiler narrows these fp32 expressions to native half arithmetic and would then
    // fuse rnd(a*b) +- rnd(c*d) into one half fma, skipping a rounding the eager reference performs.
#pragma clang fp contract(off)
    constexpr int NDB = HD / 32, HALF = ROT / 2;
    float g[NDB][16];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) g[d][i] = rnd<T>(acc[d][i] * scale);
    if constexpr (ROT > 0) {
        static_for<0, HALF / 8>([&](auto gc) {  // 8-dim groups of the first half; this lane holds 4 dims of each
            constexpr int cb = decltype(gc)::value * 8, pb = cb + HALF;
            constexpr int d = cb / 32, g4 = (cb % 32) / 8, pd = pb / 32, pg = (pb % 32) / 8;
            float c1[4], s1[4], c2[4], s2[4];
            unpack4<T>(*reinterpret_cast<const u32x2*>(cosr + cb + 4 * hh), c1);
            unpack4<T>(*reinterpret_cast<const u32x2*>(sinr + cb + 4 * hh), s1);
            unpack4<T>(*reinterpret_cast<const u32x2*>(cosr + pb + 4 * hh), c2);
            unpack4<T>(*reinterpret_cast<const u32x2*>(sinr + pb + 4 * hh), s2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g1 = g[d][4 * g4 + e], g2 = g[pd][4 * pg + e];
                g[d][4 * g4 + e] = rnd<T>(g1 * c1[e]) + rnd<T>(g2 * s2[e]);
                g[pd][4 * pg + e] = rnd<T>(g2 * c2[e]) - rnd<T>(g1 * s1[e]);
            }
        });
    }
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<u32x2*>(dst + 32 * d + 8 * g4 + 4 * hh) =
                u32x2{pack2<T>(g[d][4 * g4], g[d][4 * g4 + 1]), pack2<T>(g[d][4 * g4 + 2], g[d][4 * g4 + 3])};
}
// runtime rotary width -> the compile-time variants a head size supports (rot_d in {0, 16, 32, 64, 128}, <= HD)
template <typename T, int HD>
__device__ __forceinline__ void store_grad_row_rt(u16* __restrict__ dst, const f32x16 (&acc)[HD / 32], float scale, int hh,
                                                  const u16* __restrict__ cos_t, const u16* __restrict__ sin_t, int rot_d, int pos) {
    const u16* cr = cos_t + (size_t)pos * rot_d;
    const u16* sr = sin_t + (size_t)pos * rot_d;
    if (rot_d == 0) store_grad_row<T, HD, 0>(dst, acc, scale, hh, nullptr, nullptr);
    else if (rot_d == 16) store_grad_row<T, HD, 16>(dst, acc, scale, hh, cr, sr);
    else if (rot_d == 32) store_grad_row<T, HD, 32>(dst, acc, scale, hh, cr, sr);
    else if (rot_d == 64) store_grad_row<T, HD, 64>(dst, acc, scale, hh, cr, sr);
    else if constexpr (HD == 128) store_grad_row<T, HD, 128>(dst, acc, scale, hh, cr, sr);
}

// =====================================================================================  backward: dQ
template <typename T, int HD, int MODE, bool DROP = false>
__global__ __launch_bounds__(256, (HD == 64 ? 2 : 1)) void attn_bwd_dq_k(u16* __restrict__ dqkv, const u16* __restrict__ dctx,
                                                     const u16* __restrict__ ctx, const float* __restrict__ lse,
                                                     float* __restrict__ delta, const u16* __restrict__ qkv,
                                                     const uint8_t* __restrict__ mask,
                                                     int mask_batches, int s, int b, int np, float scale_log2e, float scale,
                                                     const u16* __restrict__ cos_t, const u16* __restrict__ sin_t, int rot_d,
                                                     int rot_off, DropArgs da = DropArgs{}) {
    typedef Tile<HD> TL;
    typedef typename M32<T>::frag frag;
    constexpr int NDS = HD / 16, NDB = HD / 32;
    __shared__ __attribute__((aligned(16))) char smem[4 * TL::BYTES];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int bi = blockIdx.x / np, hi = blockIdx.x % np;  // head-fast grid: see launch_fwd
    const int q0 = ((int)gridDim.y - 1 - (int)blockIdx.y) * 128;
    // (the forward kernel's base priority by causal weight LOSES here: dq + dkv 152 -> 159-163 us with two waves per SIMD)
#define BWD_PRIO(x) mfma_prio(x)
    const size_t tok = (size_t)b * np * 3 * HD;
    const u16* qbase = qkv + ((size_t)bi * np + hi) * 3 * HD;
    const u16* kbase = qbase + HD;
    const u16* vbase = qbase + 2 * HD;

    const int qrow = q0 + wave * 32 + r;
    const int qrow_c = qrow < s ? qrow : s - 1;
    frag qf[NDS], dof[NDS];
    const u16* dorow = dctx + ((size_t)qrow_c * b + bi) * ((size_t)np * HD) + (size_t)hi * HD;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
        qf[ds] = *reinterpret_cast<const frag*>(qbase + (size_t)qrow_c * tok + 16 * ds + 8 * hh);
        dof[ds] = *reinterpret_cast<const frag*>(dorow + 16 * ds + 8 * hh);
    }
    DropKey dkey{0, 0};
    uint32_t dqh = 0;
    if constexpr (DROP) {
        dkey = drop_key(da, (uint32_t)blockIdx.x);
        dqh = (uint32_t)qrow_c * (uint32_t)((s + 1) >> 1);
    }
    const float lse2_ld = lse[((size_t)bi * np + hi) * s + qrow_c] * kLog2e;
    // delta[q] = sum_d dO[q][d] * O[q][d]: this wave holds its 32 rows of dO as MFMA fragments (half the dims per lane half), so
    // the row sums cost one O-row load and 4*NDS FMAs here and replace a separate pass over dO and O; the dK/dV kernel, which
    // is launched after this one, reads them from `delta`
    float dlt_ld = 0.f;
    {
        const u16* orow = ctx + ((size_t)qrow_c * b + bi) * ((size_t)np * HD) + (size_t)hi * HD;
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
            float a[8], o[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(dorow + 16 * ds + 8 * hh), a);
            unpack8<T>(*reinterpret_cast<const u32x4*>(orow + 16 * ds + 8 * hh), o);
#pragma unroll
            for (int e = 0; e < 8; ++e) dlt_ld += a[e] * o[e];
        }
        dlt_ld = half_sum(dlt_ld);
        if (hh == 0 && qrow < s) delta[((size_t)bi * np + hi) * s + qrow] = dlt_ld;
    }

    const int blk_last = min(q0 + 127, s - 1);
    const int ntiles = (MODE == 0) ? (blk_last / 64 + 1) : ((s + 63) / 64);
    const int wave_last = min(q0 + wave * 32 + 31, s - 1);
    const int wave_first = q0 + wave * 32;
    const uint8_t* mrow = nullptr;
    if (MODE == 1) mrow = mask + ((size_t)(mask_batches == 1 ? 0 : bi) * s + qrow_c) * s;

    f32x16 dq[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) dq[i][j] = 0.f;

    u32x4 kreg[TL::NCH], vreg[TL::NCH];
    tile_load<HD>(kreg, kbase, tok, 0, s, tid);
    tile_load<HD>(vreg, vbase, tok, 0, s, tid);
    tile_store<HD>(smem, kreg, tid);
    tile_store<HD>(smem + TL::BYTES, vreg, tid);
    __syncthreads();
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
        pin(qf[ds]);
        pin(dof[ds]);
    }
    float lse2 = lse2_ld, dlt = dlt_ld;
    pin(lse2);
    pin(dlt);

    for (int j = 0; j < ntiles; ++j) {
        const int cur = j & 1;
        const char* ks = smem + cur * 2 * TL::BYTES;
        const char* vs = ks + TL::BYTES;
        if (j + 1 < ntiles) {
            tile_load<HD>(kreg, kbase, tok, (j + 1) * 64, s, tid);
            tile_load<HD>(vreg, vbase, tok, (j + 1) * 64, s, tid);
        }
        const int kv0 = j * 64;
        if (MODE == 1 || kv0 <= wave_last) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 sacc, dpacc;
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[i] = dpacc[i] = 0.f;
                BWD_PRIO(1);
#pragma unroll
                for (int ds = 0; ds < NDS; ++ds) {
                    sacc = M32<T>::run(row_frag<T, HD>(ks, 32 * t, ds, lane), qf[ds], sacc);
                    dpacc = M32<T>::run(row_frag<T, HD>(vs, 32 * t, ds, lane), dof[ds], dpacc);
                }
                BWD_PRIO(0);
                if constexpr (DROP) {  // dP = keep / (1 - p) * (dO . V): the gradient through the dropped probabilities
                    drop_apply_keys(dpacc, dkey, dqh, kv0 + 32 * t, hh, da.thr);
#pragma unroll
                    for (int i = 0; i < 16; ++i) dpacc[i] *= da.rscale;
                }
                if (MODE == 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = kv0 + 32 * t + acc_row(i, hh);
                        const bool dead = key >= s;
                        const bool mk = !dead && mask_at(mrow, key);
                        const float p = dead ? 0.f : fast_exp2((mk ? kMaskFill * kLog2e : sacc[i] * scale_log2e) - lse2);
                        // a masked score is the constant -10000: no gradient flows into q.k
                        sacc[i] = mk ? 0.f : p * (dpacc[i] - dlt);
                    }
                } else {
                    // packed fp32 math, two scores per VALU issue; the causal select is a separate, wave-uniformly
                    // branched pass that only the tiles on this wave's diagonal execute
                    // scalar fp32 on purpose: v_pk_*_f32 serialise with the partner wave's MFMAs, plain VALU overlaps them
                    // (tools/ubench/coissue.hip)
                    float pp[16];
                    const float nl = -lse2;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float e = __builtin_fmaf(sacc[i], scale_log2e, nl);
                        asm volatile("" : "+v"(e));
                        pp[i] = fast_exp2(e);
                    }
                    if (kv0 + 32 * t + 31 > wave_first) {  // wave-uniform
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            if (kv0 + 32 * t + acc_row(i, hh) > qrow) pp[i] = 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float dsv = pp[i] * (dpacc[i] - dlt);  // dS / scale (applied once at the end)
                        asm volatile("" : "+v"(dsv));
                        sacc[i] = dsv;
                    }
                }
                BWD_PRIO(1);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const frag dsf = M32<T>::cvt8(sacc[8 * s2], sacc[8 * s2 + 1], sacc[8 * s2 + 2], sacc[8 * s2 + 3],
                                                  sacc[8 * s2 + 4], sacc[8 * s2 + 5], sacc[8 * s2 + 6], sacc[8 * s2 + 7]);
#pragma unroll
                    for (int d = 0; d < NDB; ++d) dq[d] = M32<T>::run(tr_frag<T, HD>(ks, 32 * t + 16 * s2, d, lane), dsf, dq[d]);
                }
                BWD_PRIO(0);
            }
        }
        if (j + 1 < ntiles) {
            char* nk = smem + (cur ^ 1) * 2 * TL::BYTES;
            tile_store<HD>(nk, kreg, tid);
            tile_store<HD>(nk + TL::BYTES, vreg, tid);
        }
        __syncthreads();
    }
    if (qrow < s) {
        u16* orow = dqkv + (size_t)qrow * tok + ((size_t)bi * np + hi) * 3 * HD;
        store_grad_row_rt<T, HD>(orow, dq, scale, hh, cos_t, sin_t, rot_d, qrow + rot_off);
    }
}

// =====================================================================================  backward: dK, dV
template <typename T, int HD, int MODE, bool DROP = false>
__global__ __launch_bounds__(256, (HD == 64 ? 2 : 1)) void attn_bwd_dkv_k(u16* __restrict__ dqkv, const u16* __restrict__ dctx,
                                                      const float* __restrict__ lse, const float* __restrict__ delta,
                                                      const u16* __restrict__ qkv, const uint8_t* __restrict__ mask,
                                                      int mask_batches, int s, int b, int np, float scale_log2e, float scale,
                                                      const u16* __restrict__ cos_t, const u16* __restrict__ sin_t, int rot_d,
                                                      int rot_off, DropArgs da = DropArgs{}) {
    typedef Tile<HD> TL;
    typedef typename M32<T>::frag frag;
    constexpr int NDS = HD / 16, NDB = HD / 32;
    // [buf][Q | dO] tiles, then [buf][lse2[64] | -delta[64]] floats
    __shared__ __attribute__((aligned(16))) char smem[4 * TL::BYTES + 2 * 2 * 64 * 4];
    float* stat = reinterpret_cast<float*>(smem + 4 * TL::BYTES);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int bi = blockIdx.x / np, hi = blockIdx.x % np;  // head-fast grid: see launch_fwd
    const int k0 = blockIdx.y * 128;  // early key blocks see the most queries: they come first
    const size_t tok = (size_t)b * np * 3 * HD;
    const size_t ctok = (size_t)b * np * HD;
    const u16* qbase = qkv + ((size_t)bi * np + hi) * 3 * HD;
    const u16* kbase = qbase + HD;
    const u16* vbase = qbase + 2 * HD;
    const u16* dobase = dctx + (size_t)bi * np * HD + (size_t)hi * HD;
    const float* lse_bh = lse + ((size_t)bi * np + hi) * s;
    const float* dlt_bh = delta + ((size_t)bi * np + hi) * s;

    const int krow = k0 + wave * 32 + r;
    const int krow_c = krow < s ? krow : s - 1;
    DropKey dkey{0, 0};
    if constexpr (DROP) dkey = drop_key(da, (uint32_t)blockIdx.x);
    frag kf[NDS], vf[NDS];
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
        kf[ds] = *reinterpret_cast<const frag*>(kbase + (size_t)krow_c * tok + 16 * ds + 8 * hh);
        vf[ds] = *reinterpret_cast<const frag*>(vbase + (size_t)krow_c * tok + 16 * ds + 8 * hh);
    }
    const int nqt = (s + 63) / 64;
    const int jt0 = (MODE == 0) ? (k0 / 64) : 0;
    const int wave_first = k0 + wave * 32;  // smallest key of this wave
    const uint8_t* mbase = nullptr;
    if (MODE == 1) mbase = mask + (size_t)(mask_batches == 1 ? 0 : bi) * s * s;

    f32x16 dk[NDB], dv[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) dk[i][j] = dv[i][j] = 0.f;

    u32x4 qreg[TL::NCH], oreg[TL::NCH];
    float st_l = 0.f, st_d = 0.f;
    auto load_stats = [&](int row0) {
        if (tid < 64) {
            int q = row0 + tid;
            q = q < s ? q : s - 1;
            st_l = lse_bh[q];  // scaled at store time: touching it here would wait for the load
            st_d = dlt_bh[q];
        }
    };
    auto store_stats = [&](int buf) {
        if (tid < 64) {
            stat[buf * 128 + tid] = st_l * kLog2e;
            stat[buf * 128 + 64 + tid] = -st_d;  // negated: it is the INITIAL ACCUMULATOR of the dP product (below)
        }
    };
    tile_load<HD>(qreg, qbase, tok, jt0 * 64, s, tid);
    tile_load<HD>(oreg, dobase, ctok, jt0 * 64, s, tid);
    load_stats(jt0 * 64);
    tile_store<HD>(smem, qreg, tid);
    tile_store<HD>(smem + TL::BYTES, oreg, tid);
    store_stats(0);
    __syncthreads();
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
        pin(kf[ds]);
        pin(vf[ds]);
    }

    for (int j = jt0; j < nqt; ++j) {
        const int cur = (j - jt0) & 1;
        const char* qs = smem + cur * 2 * TL::BYTES;
        const char* os = qs + TL::BYTES;
        const float* sl = stat + cur * 128;
        if (j + 1 < nqt) {
            tile_load<HD>(qreg, qbase, tok, (j + 1) * 64, s, tid);
            tile_load<HD>(oreg, dobase, ctok, (j + 1) * 64, s, tid);
            load_stats((j + 1) * 64);
        }
        const int q0t = j * 64;
        // (Round 3 experiment, measured and removed: for full tiles the two 32-query halves software-pipelined inside the wave - the
        // S / dP MFMAs of half 1 issued between the softmax / dS arithmetic of half 0, the dV / dK MFMAs of half 0 between the
        // arithmetic of half 1, order pinned with sched_group_barrier; 236 VGPRs, same two waves per SIMD, bit-identical results:
        // dq + dkv 153 -> 159 us.  tools/ubench/coissue2.hip shows one wave CAN overlap the two pipes (16 MFMAs + 128 v_fma in
        // 801 cycles against 520 + 731 apart) - with register operands.  Here every MFMA's A operand is an LDS read that hipcc
        // places just ahead of its use, so the interleaved stream stops at an s_waitcnt lgkmcnt in front of each MFMA and each
        // statistics load, and an in-order wave that waits there issues nothing on either pipe.)
        if (MODE == 1 || q0t + 63 >= wave_first) {  // some query of the tile can see a key of this wave
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 sacc, dpacc;
                // Row constants as the initial accumulator: the registers of a lane are QUERIES here (4 consecutive rows per
                // group), so -delta[q] comes straight from the tile's statistics in LDS into the accumulator of the dP chain
                // and dP - delta leaves the MFMAs ready-made - 16 fewer subtractions per 32 x 32 block in a loop that the
                // vector port, not the matrix pipe, bounds.  (Dropout scales dP before the subtraction: it keeps the zero.)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 nd4 = *reinterpret_cast<const f32x4*>(sl + 64 + 32 * t + 8 * g4 + 4 * hh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sacc[4 * g4 + e] = 0.f;
                        dpacc[4 * g4 + e] = DROP ? 0.f : nd4[e];
                    }
                }
                BWD_PRIO(1);
#pragma unroll
                for (int ds = 0; ds < NDS; ++ds) {
                    sacc = M32<T>::run(row_frag<T, HD>(qs, 32 * t, ds, lane), kf[ds], sacc);    // S[q][key]
                    dpacc = M32<T>::run(row_frag<T, HD>(os, 32 * t, ds, lane), vf[ds], dpacc);  // dP[q][key] - delta[q]
                }
                BWD_PRIO(0);
                float pv[16];
                bool keep[16];
                if constexpr (DROP) {  // here the lane is a KEY and the registers are queries: one hash per element
                    const uint32_t half_s = (uint32_t)((s + 1) >> 1);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int q = min(q0t + 32 * t + acc_row(i, hh), s - 1);
                        keep[i] = drop_keep(dkey, (uint32_t)q * half_s, krow_c, da.thr);
                        dpacc[i] = keep[i] ? dpacc[i] * da.rscale : 0.f;
                    }
                }
                // wave-uniform: only tiles on this wave's diagonal (or the ragged last tile) pay for the compare/select
                const bool need_mask = (wave_first + 31 > q0t + 32 * t) || (q0t + 32 * t + 32 > s);
                if (MODE == 0) {
                    // packed fp32 math, two scores per VALU issue; the causal select is its own wave-uniformly branched pass
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int ql = 32 * t + 8 * g4 + 4 * hh;  // local row of register 4*g4
                        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sl + ql);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {  // scalar fp32 on purpose (see the dQ kernel)
                            float x = __builtin_fmaf(sacc[4 * g4 + e], scale_log2e, -l4[e]);
                            asm volatile("" : "+v"(x));
                            pv[4 * g4 + e] = fast_exp2(x);
                        }
                    }
                    if (need_mask) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int q = q0t + 32 * t + acc_row(i, hh);
                            if (krow > q || q >= s) pv[i] = 0.f;
                        }
                    }
                    if constexpr (DROP) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const f32x4 nd4 = *reinterpret_cast<const f32x4*>(sl + 64 + 32 * t + 8 * g4 + 4 * hh);
#pragma unroll
                            for (int e = 0; e < 4; ++e) dpacc[4 * g4 + e] += nd4[e];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float dsv = pv[i] * dpacc[i];  // dS / scale (dpacc = dP - delta); the scale is applied once to dK at the end
                        asm volatile("" : "+v"(dsv));
                        sacc[i] = dsv;
                    }
                } else {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int ql = 32 * t + 8 * g4 + 4 * hh;
                        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sl + ql);
                        f32x4 nd4 = {0.f, 0.f, 0.f, 0.f};
                        if constexpr (DROP) nd4 = *reinterpret_cast<const f32x4*>(sl + 64 + ql);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = 4 * g4 + e;
                            const int q = q0t + ql + e;
                            const bool dead = (q >= s) || (krow >= s);
                            const bool mk = !dead && mask_at(mbase + (size_t)q * s, krow);
                            const float p = dead ? 0.f : fast_exp2((mk ? kMaskFill * kLog2e : sacc[i] * scale_log2e) - l4[e]);
                            sacc[i] = mk ? 0.f : p * (dpacc[i] + nd4[e]);  // dpacc already holds dP - delta unless DROP
                            pv[i] = p;
                        }
                    }
                }
                if constexpr (DROP) {  // dV sees the dropped, rescaled probabilities; dS above used the undropped ones
#pragma unroll
                    for (int i = 0; i < 16; ++i) pv[i] = keep[i] ? pv[i] * da.rscale : 0.f;
                }
                BWD_PRIO(1);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const frag pf = M32<T>::cvt8(pv[8 * s2], pv[8 * s2 + 1], pv[8 * s2 + 2], pv[8 * s2 + 3], pv[8 * s2 + 4],
                                                 pv[8 * s2 + 5], pv[8 * s2 + 6], pv[8 * s2 + 7]);
                    const frag dsf = M32<T>::cvt8(sacc[8 * s2], sacc[8 * s2 + 1], sacc[8 * s2 + 2], sacc[8 * s2 + 3],
                                                  sacc[8 * s2 + 4], sacc[8 * s2 + 5], sacc[8 * s2 + 6], sacc[8 * s2 + 7]);
#pragma unroll
                    for (int d = 0; d < NDB; ++d) {
                        dv[d] = M32<T>::run(tr_frag<T, HD>(os, 32 * t + 16 * s2, d, lane), pf, dv[d]);   // dV^T += dO^T . P
                        dk[d] = M32<T>::run(tr_frag<T, HD>(qs, 32 * t + 16 * s2, d, lane), dsf, dk[d]);  // dK^T += Q^T . dS
                    }
                }
                BWD_PRIO(0);
            }
        }
        if (j + 1 < nqt) {
            char* nq = smem + (cur ^ 1) * 2 * TL::BYTES;
            tile_store<HD>(nq, qreg, tid);
            tile_store<HD>(nq + TL::BYTES, oreg, tid);
            store_stats(cur ^ 1);
        }
        __syncthreads();
    }
    if (krow < s) {
        u16* krow_o = dqkv + (size_t)krow * tok + ((size_t)bi * np + hi) * 3 * HD + HD;
        store_grad_row_rt<T, HD>(krow_o, dk, scale, hh, cos_t, sin_t, rot_d, krow + rot_off);
        store_grad_row<T, HD, 0>(krow_o + HD, dv, 1.0f, hh, nullptr, nullptr);
    }
}

#undef BWD_PRIO

template <typename T, int HD>
static int launch_fwd(void* ctx, float* lse, const void* qkv, const uint8_t* mask, int mode, int mb, int s, int b, int np,
                      float scale, hipStream_t st, const DropArgs* da = nullptr) {
    // Grid: x = (batch, head) - fastest - and y = 128-row block, heaviest causal blocks first.  Dispatch is x-major, so
    // (1) every heavy block of every head starts before any light one (longest-processing-time order: the causal
    // triangle's 16:1 work spread no longer leaves a tail), and (2) workgroup id % 8 = head % 8 picks the XCD, so all
    // blocks that stream one head's K/V share one XCD's L2 instead of spreading every head over all eight.
    dim3 grid(b * np, (s + 127) / 128), block(256);
    const float sl2 = scale * kLog2e;
    if (da) {  // attention dropout (training with attention_dropout > 0): its own instantiations, the p = 0 kernels stay as they are
        if (mode == 0)
            attn_fwd_k<T, HD, 0, true><<<grid, block, 0, st>>>((u16*)ctx, lse, (const u16*)qkv, nullptr, 1, s, b, np, sl2, *da);
        else
            attn_fwd_k<T, HD, 1, true><<<grid, block, 0, st>>>((u16*)ctx, lse, (const u16*)qkv, mask, mb, s, b, np, sl2, *da);
    } else if (mode == 0)
        attn_fwd_k<T, HD, 0><<<grid, block, 0, st>>>((u16*)ctx, lse, (const u16*)qkv, nullptr, 1, s, b, np, sl2);
    else
        attn_fwd_k<T, HD, 1><<<grid, block, 0, st>>>((u16*)ctx, lse, (const u16*)qkv, mask, mb, s, b, np, sl2);
    MM_CHECK_LAUNCH();
    return MM_OK;
}

template <typename T, int HD>
static int launch_bwd(void* dqkv, float* delta, const void* dctx, const void* ctx, const float* lse, const void* qkv,
                      const uint8_t* mask, int mode, int mb, int s, int b, int np, float scale, const u16* cs, const u16* sn,
                      int rd, int ro, hipStream_t st, const DropArgs* da = nullptr) {
    dim3 grid(b * np, (s + 127) / 128), block(256);  // head-fast, heavy blocks first (see launch_fwd)
    const float sl2 = scale * kLog2e;
    const uint8_t* mk = mode == 0 ? nullptr : mask;
    const int mbb = mode == 0 ? 1 : mb;
    // dQ first: it also leaves delta[b, np, s] for the dK/dV kernel behind it on the stream
#define MM_ATTN_BWD(MODE, DROP, ...)                                                                                                \
    do {                                                                                                                            \
        attn_bwd_dq_k<T, HD, MODE, DROP><<<grid, block, 0, st>>>((u16*)dqkv, (const u16*)dctx, (const u16*)ctx, lse, delta,           \
                                                                 (const u16*)qkv, mk, mbb, s, b, np, sl2, scale, cs, sn, rd, ro __VA_ARGS__); \
        MM_CHECK_LAUNCH();                                                                                                          \
        attn_bwd_dkv_k<T, HD, MODE, DROP><<<grid, block, 0, st>>>((u16*)dqkv, (const u16*)dctx, lse, delta, (const u16*)qkv, mk, mbb, \
                                                                  s, b, np, sl2, scale, cs, sn, rd, ro __VA_ARGS__);                 \
    } while (0)
    if (da) {
        if (mode == 0) MM_ATTN_BWD(0, true, , *da);
        else MM_ATTN_BWD(1, true, , *da);
    } else {
        if (mode == 0) MM_ATTN_BWD(0, false);
        else MM_ATTN_BWD(1, false);
    }
#undef MM_ATTN_BWD
    MM_CHECK_LAUNCH();
    return MM_OK;
}

// the keep mask of the attention dropout as bytes [b, np, s, s] (1 = kept): the definition the kernels above implement
__global__ __launch_bounds__(256) void attn_dropout_mask_k(uint8_t* __restrict__ keep, int s, DropArgs da) {
    const DropKey dkey = drop_key(da, (uint32_t)blockIdx.y);
    const uint32_t half_s = (uint32_t)((s + 1) >> 1);
    const size_t n = (size_t)s * s;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i / s), k = (int)(i % s);
        keep[(size_t)blockIdx.y * n + i] = drop_keep(dkey, (uint32_t)q * half_s, k, da.thr) ? 1 : 0;
    }
}

static int make_drop(const char* fn, DropArgs& da, float p, const void* rng_state, unsigned long long pos) {
    MM_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout probability %g outside [0, 1)", fn, (double)p);
    MM_CHECK_ARG(rng_state != nullptr, "%s: rng_state (device uint64[2] {seed, offset}) missing", fn);
    long thr = lrintf(p * 65536.0f);
    thr = thr < 0 ? 0 : (thr > 65535 ? 65535 : thr);
    da.st = (const unsigned long long*)rng_state;
    da.pos = pos;
    da.thr = (uint32_t)thr;
    da.rscale = 65536.0f / (float)(65536 - thr);
    return MM_OK;
}

static int check_common(const char* fn, int mask_mode, const uint8_t* mask, int mask_batches, int s, int b, int np, int hn) {
    MM_CHECK_ARG(s > 0 && b > 0 && np > 0, "%s: s=%d b=%d np=%d must be positive", fn, s, b, np);
    MM_CHECK_ARG(hn == 64 || hn == 128, "%s: head size %d not supported (64 or 128)", fn, hn);
    MM_CHECK_ARG(mask_mode == 0 || mask_mode == 1, "%s: mask_mode must be 0 (causal) or 1 (byte mask)", fn);
    if (mask_mode == 1) {
        MM_CHECK_ARG(mask != nullptr, "%s: mask_mode 1 needs a mask", fn);
        MM_CHECK_ARG(mask_batches == 1 || mask_batches == b, "%s: mask batch %d must be 1 or %d", fn, mask_batches, b);
    }
    MM_CHECK_ARG((s + 127) / 128 <= 65535, "%s: s=%d exceeds the grid limit", fn, s);
    return MM_OK;
}

}  // namespace mm

using namespace mm;

extern "C" int mm_attn_fwd(void* ctx, float* lse, const void* qkv, const uint8_t* mask, int mask_mode, int mask_batches, int s,
                           int b, int np, int hn, float softmax_scale, int dtype, void* stream) {
    MM_CHECK_ARG(ctx && lse && qkv, "mm_attn_fwd: null pointer");
    if (int rc = check_common("mm_attn_fwd", mask_mode, mask, mask_batches, s, b, np, hn)) return rc;
    MM_CHECK_ARG(aligned16(qkv) && aligned16(ctx), "mm_attn_fwd: qkv/ctx must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        if (hn == 64) return launch_fwd<T, 64>(ctx, lse, qkv, mask, mask_mode, mask_batches, s, b, np, softmax_scale, st);
        return launch_fwd<T, 128>(ctx, lse, qkv, mask, mask_mode, mask_batches, s, b, np, softmax_scale, st);
    });
}

extern "C" int mm_attn_fwd_dropout(void* ctx, float* lse, const void* qkv, const uint8_t* mask, int mask_mode, int mask_batches,
                                   int s, int b, int np, int hn, float softmax_scale, float p, const void* rng_state,
                                   unsigned long long pos, int dtype, void* stream) {
    MM_CHECK_ARG(ctx && lse && qkv, "mm_attn_fwd_dropout: null pointer");
    if (int rc = check_common("mm_attn_fwd_dropout", mask_mode, mask, mask_batches, s, b, np, hn)) return rc;
    MM_CHECK_ARG(aligned16(qkv) && aligned16(ctx), "mm_attn_fwd_dropout: qkv/ctx must be 16-byte aligned");
    DropArgs da;
    if (int rc = make_drop("mm_attn_fwd_dropout", da, p, rng_state, pos)) return rc;
    hipStream_t st = (hipStream_t)stream;
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        if (hn == 64) return launch_fwd<T, 64>(ctx, lse, qkv, mask, mask_mode, mask_batches, s, b, np, softmax_scale, st, &da);
        return launch_fwd<T, 128>(ctx, lse, qkv, mask, mask_mode, mask_batches, s, b, np, softmax_scale, st, &da);
    });
}

static int attn_bwd_impl(void* dqkv, float* delta, const void* dctx, const void* ctx, const float* lse, const void* qkv,
                         const uint8_t* mask, int mask_mode, int mask_batches, int s, int b, int np, int hn, float softmax_scale,
                         const void* cos, const void* sin, int rot_d, int rot_offset, int dtype, void* stream, const DropArgs* da);

extern "C" int mm_attn_bwd_dropout(void* dqkv, float* delta, const void* dctx, const void* ctx, const float* lse, const void* qkv,
                                   const uint8_t* mask, int mask_mode, int mask_batches, int s, int b, int np, int hn,
                                   float softmax_scale, const void* cos, const void* sin, int rot_d, int rot_offset, float p,
                                   const void* rng_state, unsigned long long pos, int dtype, void* stream) {
    DropArgs da;
    if (int rc = make_drop("mm_attn_bwd_dropout", da, p, rng_state, pos)) return rc;
    return attn_bwd_impl(dqkv, delta, dctx, ctx, lse, qkv, mask, mask_mode, mask_batches, s, b, np, hn, softmax_scale, cos, sin, rot_d,
                         rot_offset, dtype, stream, &da);
}

extern "C" int mm_attn_dropout_mask(uint8_t* keep, int s, int b, int np, float p, const void* rng_state, unsigned long long pos,
                                    void* stream) {
    MM_CHECK_ARG(keep && s > 0 && b > 0 && np > 0 && (long)b * np <= 65535, "mm_attn_dropout_mask: bad arguments");
    DropArgs da;
    if (int rc = make_drop("mm_attn_dropout_mask", da, p, rng_state, pos)) return rc;
    const size_t n = (size_t)s * s;
    dim3 grid((unsigned)std::min<size_t>((n + 255) / 256, 1024), b * np);
    attn_dropout_mask_k<<<grid, 256, 0, (hipStream_t)stream>>>(keep, s, da);
    MM_CHECK_LAUNCH();
    return MM_OK;
}

extern "C" int mm_attn_bwd_rotary(void* dqkv, float* delta, const void* dctx, const void* ctx, const float* lse, const void* qkv,
                                  const uint8_t* mask, int mask_mode, int mask_batches, int s, int b, int np, int hn,
                                  float softmax_scale, const void* cos, const void* sin, int rot_d, int rot_offset, int dtype,
                                  void* stream) {
    return attn_bwd_impl(dqkv, delta, dctx, ctx, lse, qkv, mask, mask_mode, mask_batches, s, b, np, hn, softmax_scale, cos, sin, rot_d,
                         rot_offset, dtype, stream, nullptr);
}

static int attn_bwd_impl(void* dqkv, float* delta, const void* dctx, const void* ctx, const float* lse, const void* qkv,
                         const uint8_t* mask, int mask_mode, int mask_batches, int s, int b, int np, int hn, float softmax_scale,
                         const void* cos, const void* sin, int rot_d, int rot_offset, int dtype, void* stream, const DropArgs* da) {
    MM_CHECK_ARG(dqkv && delta && dctx && ctx && lse && qkv, "mm_attn_bwd: null pointer");
    if (int rc = check_common("mm_attn_bwd", mask_mode, mask, mask_batches, s, b, np, hn)) return rc;
    MM_CHECK_ARG(aligned16(qkv) && aligned16(ctx) && aligned16(dctx) && aligned16(dqkv), "mm_attn_bwd: tensors must be 16-byte aligned");
    if (rot_d != 0) {
        MM_CHECK_ARG(cos && sin && aligned16(cos) && aligned16(sin), "mm_attn_bwd_rotary: cos/sin tables missing or not 16-byte aligned");
        MM_CHECK_ARG((rot_d == 16 || rot_d == 32 || rot_d == 64 || rot_d == 128) && rot_d <= hn && rot_offset >= 0,
                     "mm_attn_bwd_rotary: rotary dims %d (offset %d) not supported for head size %d", rot_d, rot_offset, hn);
    }
    hipStream_t st = (hipStream_t)stream;
    return dispatch_dtype(dtype, [&](auto t) -> int {
        using T = decltype(t);
        if (hn == 64)
            return launch_bwd<T, 64>(dqkv, delta, dctx, ctx, lse, qkv, mask, mask_mode, mask_batches, s, b, np, softmax_scale,
                                     (const u16*)cos, (const u16*)sin, rot_d, rot_offset, st, da);
        return launch_bwd<T, 128>(dqkv, delta, dctx, ctx, lse, qkv, mask, mask_mode, mask_batches, s, b, np, softmax_scale,
                                  (const u16*)cos, (const u16*)sin, rot_d, rot_offset, st, da);
    });
}

extern "C" int mm_attn_bwd(void* dqkv, float* delta, const void* dctx, const void* ctx, const float* lse, const void* qkv,
                           const uint8_t* mask, int mask_mode, int mask_batches, int s, int b, int np, int hn,
                           float softmax_scale, int dtype, void* stream) {
    return mm_attn_bwd_rotary(dqkv, delta, dctx, ctx, lse, qkv, mask, mask_mode, mask_batches, s, b, np, hn, softmax_scale, nullptr,
                              nullptr, 0, 0, dtype, stream);
}
