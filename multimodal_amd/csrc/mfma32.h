// mfma32.h - MFMA 32x32x16 fragment helpers shared by the attention and the weight-gradient (TN) kernels:
// the padded + XOR-swizzled LDS image of a streamed [64 rows][HD] 16-bit tile, read by rows (ds_read_b128) or
// transposed (ds_read_b64_tr_b16), both conflict-free.
#pragma once
#include "common.h"

namespace mm {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <typename T> struct M32;
template <> struct M32<BF16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ frag cvt8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
        return frag{(__bf16)a0, (__bf16)a1, (__bf16)a2, (__bf16)a3, (__bf16)a4, (__bf16)a5, (__bf16)a6, (__bf16)a7};
    }
};
template <> struct M32<F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ frag cvt8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
        return frag{(_Float16)a0, (_Float16)a1, (_Float16)a2, (_Float16)a3, (_Float16)a4, (_Float16)a5, (_Float16)a6, (_Float16)a7};
    }
};

// ---- LDS image of a streamed [64 rows][HD] tile
// HD = 128 (and the TN kernel's 128-column tiles): 64-byte row pad, so that the 4 rows of a transposed-read pass start 48
// banks apart and cover all 64 banks, + a 4-way XOR of the 16-byte chunk index for the row reads.
// HD = 64: NO pad (128-byte rows, 8 KiB per tile instead of 12: a workgroup's K/V ring is 32 KiB and four workgroups fit a
// CU).  A row is two 64-byte halves; the XOR value of a row, bits (2,1,0) = (row>>1 &1, row>>2 &1, row>>3 &1), does both jobs:
//   * transposed reads: a pass reads 64 bytes of each of rows 4a..4a+3; row parity picks banks 0-31 / 32-63 and bit 2 of the XOR
//     (row>>1 &1) flips the 64-byte half -> the four rows land on the four bank quarters;
//   * row reads / stores (16 lanes = 16 consecutive rows at one chunk index, or 2 rows x 8 chunks): within one parity the XOR
//     value runs through all 8 values over 16 consecutive rows -> 16 distinct 4-bank groups.
template <int HD> struct Tile {
    static constexpr int STRIDE = HD == 64 ? 128 : HD * 2 + 64;  // bytes per row
    static constexpr int BYTES = 64 * STRIDE;
    static constexpr int CPR = HD / 8;                // 16-byte chunks per row
    static constexpr int NCH = 64 * CPR / 256;        // chunks per thread per tile
    static __device__ __forceinline__ int swz(int row) {
        if constexpr (HD == 64) return (((row >> 1) & 1) << 2) | ((row >> 1) & 2) | ((row >> 3) & 1);
        else return (row >> 2) & 3;
    }
    static __device__ __forceinline__ int off(int row, int chunk) { return row * STRIDE + ((chunk ^ swz(row)) << 4); }
};

template <int HD>
__device__ __forceinline__ void tile_load(u32x4 (&regs)[Tile<HD>::NCH], const u16* __restrict__ g, size_t row_stride, int row0,
                                          int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < Tile<HD>::NCH; ++i) {
        const int c = tid + 256 * i;
        const int row = c / Tile<HD>::CPR, ch = c % Tile<HD>::CPR;
        int gr = row0 + row;
        gr = gr < nrows ? gr : nrows - 1;
        regs[i] = *reinterpret_cast<const u32x4*>(g + (size_t)gr * row_stride + ch * 8);
    }
}
template <int HD>
__device__ __forceinline__ void tile_store(char* lds, const u32x4 (&regs)[Tile<HD>::NCH], int tid) {
#pragma unroll
    for (int i = 0; i < Tile<HD>::NCH; ++i) {
        const int c = tid + 256 * i;
        const int row = c / Tile<HD>::CPR, ch = c % Tile<HD>::CPR;
        *reinterpret_cast<u32x4*>(lds + Tile<HD>::off(row, ch)) = regs[i];
    }
}
// A operand, rows = tile rows: lane (r = lane&31, hh = lane>>5) gets tile[row0 + r][16*ds + 8*hh .. +7]
template <typename T, int HD>
__device__ __forceinline__ typename M32<T>::frag row_frag(const char* lds, int row0, int ds, int lane) {
    const int row = row0 + (lane & 31);
    return *reinterpret_cast<const typename M32<T>::frag*>(lds + Tile<HD>::off(row, 2 * ds + (lane >> 5)));
}
// A operand, rows = tile COLUMNS (d = 32*dblk + (lane&31)), k = tile rows in the order the accumulator-as-B-operand
// uses: element j of lane half hh <-> tile row kb + 8*(j>>2) + 4*hh + (j&3).
template <typename T, int HD>
__device__ __forceinline__ typename M32<T>::frag tr_frag(const char* lds, int kb, int dblk, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, hh = lane >> 5;
    const int col = 32 * dblk + 16 * (g & 1) + 4 * p;
    const int ra = kb + 4 * hh + q;
    const int rb = ra + 8;
    const char* pa = lds + Tile<HD>::off(ra, col >> 3) + (col & 7) * 2;
    const char* pb = lds + Tile<HD>::off(rb, col >> 3) + (col & 7) * 2;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
    const s16x8 v = s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    typename M32<T>::frag f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}

// row index (within a 32-row score sub-tile) held by accumulator register `reg` of lane half hh
__device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

}  // namespace mm
