// common.h - shared device helpers for the gfx950 kernels (wave64, 16-byte vector access).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <type_traits>
#include "../../include/mm_hip.h"

namespace mm {

void set_error(const char* fmt, ...);
int cu_share();  // this thread's mm_set_cu_share value (0: none)

#define MM_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            mm::set_error(__VA_ARGS__);    \
            return (int)MM_EINVAL;             \
        }                                  \
    } while (0)

#define MM_CHECK_LAUNCH()                                                  \
    do {                                                                   \
        hipError_t e__ = hipGetLastError();                                \
        if (e__ != hipSuccess) {                                           \
            mm::set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return (int)MM_ELAUNCH;                                            \
        }                                                                  \
    } while (0)

constexpr int kWave = 64;

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E) - indices usable as template arguments
template <int B, int E, typename F> __host__ __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

typedef uint16_t u16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// ---- 16-bit <-> fp32 conversions.  T tag: 0 = bf16, 1 = f16.
struct BF16 {};
struct F16 {};

template <typename T> __device__ __forceinline__ float to_f32(u16 v);
template <> __device__ __forceinline__ float to_f32<BF16>(u16 v) { return __uint_as_float(((uint32_t)v) << 16); }
template <> __device__ __forceinline__ float to_f32<F16>(u16 v) {
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return (float)h;
}
template <typename T> __device__ __forceinline__ u16 from_f32(float f);
template <> __device__ __forceinline__ u16 from_f32<BF16>(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
    __bf16 h = (__bf16)f;
    u16 r;
    __builtin_memcpy(&r, &h, 2);
    return r;
}
template <> __device__ __forceinline__ u16 from_f32<F16>(float f) {
    _Float16 h = (_Float16)f;
    u16 r;
    __builtin_memcpy(&r, &h, 2);
    return r;
}
// round-trip through the 16-bit type (mimics an eager elementwise op's output rounding)
template <typename T> __device__ __forceinline__ float rnd(float f) { return to_f32<T>(from_f32<T>(f)); }

template <typename T> __device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = to_f32<T>((u16)(v[i] & 0xffffu));
        f[2 * i + 1] = to_f32<T>((u16)(v[i] >> 16));
    }
}
template <typename T> __device__ __forceinline__ void unpack4(const u32x2& v, float (&f)[4]) {
    f[0] = to_f32<T>((u16)(v[0] & 0xffffu)); f[1] = to_f32<T>((u16)(v[0] >> 16));
    f[2] = to_f32<T>((u16)(v[1] & 0xffffu)); f[3] = to_f32<T>((u16)(v[1] >> 16));
}
// two fp32 -> one dword of two 16-bit values (lo = a, hi = b), round-to-nearest-even: ONE v_cvt_pk_{bf16,f16}_f32
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b);
template <> __device__ __forceinline__ uint32_t pack2<BF16>(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
template <> __device__ __forceinline__ uint32_t pack2<F16>(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, f16x2));
}
template <typename T> __device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2<T>(f[2 * i], f[2 * i + 1]);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// tanh(u) = 1 - 2 / (exp(2u) + 1): one v_exp_f32 + one v_rcp_f32; |error| < 2e-7 absolute over the whole range
// (exp overflow -> inf -> 1 - 0 = 1; underflow -> 1 - 2 = -1), far below the 16-bit output rounding.
__device__ __forceinline__ float fast_tanh(float u) {
    const float e = __expf(2.0f * u);
    return 1.0f - 2.0f * __frcp_rn(e + 1.0f);
}

// tanh-GeLU (reference megatron/model/activations.py:62-70) in its sigmoid form: with u = 0.79788456 x (1 + 0.044715 x^2),
// 0.5 (1 + tanh u) = 1 / (1 + exp(-2u)) =: s, so gelu(x) = x s and, for bias_gelu_back (activations.py:75-86),
// gelu'(x) = 0.5 x (1 - t^2)(0.79788456 + 0.1070322243 x^2) + 0.5 (1 + t) = s (1 + x (1.59576912 + 0.2140644486 x^2)(1 - s)).
// 5 (forward) / 10 (derivative) plain VALU ops + one v_exp_f32 + one v_rcp_f32 per value; the constants fold -2 log2(e).
// exp2 overflow (x << 0) gives s = 0, underflow s = 1: both limits are exact.
__device__ __forceinline__ float gelu_sigmoid(float x, float x2) {
    const float w = x * __builtin_fmaf(-0.1029432396f, x2, -2.302208198f);  // -2u log2(e)
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));
}
__device__ __forceinline__ float gelu_tanh_fast(float x) { return x * gelu_sigmoid(x, x * x); }
__device__ __forceinline__ float gelu_tanh_grad_fast(float x) {
    const float x2 = x * x;
    const float s = gelu_sigmoid(x, x2);
    const float xc = x * __builtin_fmaf(0.2140644486f, x2, 1.59576912f);
    return s * __builtin_fmaf(xc, 1.0f - s, 1.0f);
}

template <typename F> inline int dispatch_dtype(int dtype, F&& f) {
    if (dtype == MM_BF16) return f(BF16{});
    if (dtype == MM_F16) return f(F16{});
    set_error("unsupported dtype code %d (only MM_BF16, MM_F16)", dtype);
    return MM_EUNSUPPORTED;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace mm
