// Developer microbenchmark: do the matrix pipe and the vector ALU of one SIMD overlap across two waves?
// 512-thread workgroups (one per CU), waves 0-3 = group A, waves 4-7 = group B (waves w and w+4 share a SIMD).  Per segment group A
// issues NM MFMAs (32x32x16 bf16) and group B issues NV vector instructions of a given kind, then everybody meets at a barrier.
// Prints cycles per segment for A only, B only and both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// CH = independent accumulator chains of group A's MFMAs (1: every MFMA waits for the one before it, as a 4-deep S = K.Q^T chain does)
template <int KIND, bool DO_A, bool DO_B, bool SWAP, int CH = 2>
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int nseg, int nm, int nv) {
    const int wave = threadIdx.x >> 6;
    const bool is_a = SWAP ? (wave >= 4) : (wave < 4);
    f32x16 acc0, acc1, acc2, acc3;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + 3 * i);
    f32x2 p[4];
    for (int i = 0; i < 4; ++i) p[i] = f32x2{0.001f * threadIdx.x, 0.002f * i};
    const long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < nseg; ++s) {
        if (is_a) {
            if (DO_A)
                for (int i = 0; i < nm; i += 4) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                    if (CH == 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc0, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
                    if (CH == 4) {
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
                        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc3, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                        if (CH == 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc0, 0, 0, 0);
                        else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
                    }
                }
        } else if (DO_B) {
            for (int i = 0; i < nv; i += 8) {
                if (KIND == 0) {  // v_fma_f32, 8 independent chains
#pragma unroll
                    for (int j = 0; j < 8; ++j) { v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f); asm volatile("" : "+v"(v[j])); }
                } else if (KIND == 1) {  // v_exp_f32
#pragma unroll
                    for (int j = 0; j < 8; ++j) { v[j] = __builtin_amdgcn_exp2f(v[j]); asm volatile("" : "+v"(v[j])); }
                } else if (KIND == 3) {  // v_cvt_pk_bf16_f32
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
                        bf2 c0 = bf2{(__bf16)v[j], (__bf16)v[j + 1]}, c1 = bf2{(__bf16)v[j + 1], (__bf16)v[j]};
                        unsigned u0, u1; __builtin_memcpy(&u0, &c0, 4); __builtin_memcpy(&u1, &c1, 4);
                        asm volatile("" : "+v"(u0)); asm volatile("" : "+v"(u1));
                        v[j] = __uint_as_float(u0 & 0xffff0000u); v[j + 1] = __uint_as_float(u1 << 16);
                        asm volatile("" : "+v"(v[j])); asm volatile("" : "+v"(v[j + 1]));
                    }
                } else if (KIND == 4) {  // v_max3_f32
#pragma unroll
                    for (int j = 0; j < 8; ++j) { v[j] = fmaxf(fmaxf(v[j], v[(j + 1) & 7]), v[(j + 3) & 7]); asm volatile("" : "+v"(v[j])); }
                } else if (KIND == 5) {  // v_cmp + v_cndmask
#pragma unroll
                    for (int j = 0; j < 8; j += 2) { v[j] = (v[j + 1] > (float)i) ? v[j] : -1.0f; asm volatile("" : "+v"(v[j])); v[j + 1] = (v[j] > (float)s) ? v[j + 1] : 2.0f; asm volatile("" : "+v"(v[j + 1])); }
                } else if (KIND == 6) {  // v_permlane32_swap
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j]), false, false);
                        v[j] = __uint_as_float(r[0]) + 0.0f; asm volatile("" : "+v"(v[j]));
                    }
                } else if (KIND == 7) {  // v_mov_b32 (register copies)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { float t = v[(j + 1) & 7]; asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(t)); }
                } else {  // v_pk_fma_f32 (4 per 8 "instructions": counts as 4 issues)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { p[j] = p[j] * f32x2{1.0001f, 0.9999f} + f32x2{0.5f, 0.25f}; asm volatile("" : "+v"(p[j])); }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { p[j] = p[j] * f32x2{0.9999f, 1.0001f} + f32x2{0.25f, 0.5f}; asm volatile("" : "+v"(p[j])); }
                }
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = acc0[0] + acc1[5] + acc2[3] + acc3[7];
    for (int i = 0; i < 8; ++i) r += v[i];
    for (int i = 0; i < 4; ++i) r += p[i][0] + p[i][1];
    if (r == 1234.5678f) out[threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, bool A, bool B, bool SWAP, int CH = 2> double run(int nseg, int nm, int nv, float* out, long long* cyc) {
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<KIND, A, B, SWAP, CH>), dim3(256), dim3(512), 0, 0, out, cyc, nseg, nm, nv);
        hipDeviceSynchronize();
    }
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    return (double)h / nseg;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    const int nseg = 2000, nm = 16;
    const char* names[8] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "v_cvt_pk_bf16", "v_max3_f32", "v_cmp+cndmask", "permlane32_swap", "v_mov_b32"};
    printf("cycles are s_memrealtime-independent shader clocks (readcyclecounter) per segment; 16 MFMA 32x32x16 per segment in group A\n");
    for (int nv = 128; nv <= 128; nv += 64) {
        double a = run<0, true, false, false>(nseg, nm, nv, out, cyc);
        printf("nv=%d  A only %.0f\n", nv, a);
#define ROW(K)                                                                                                        \
        printf("   %-14s B only %.0f   A+B %.0f   A+B (groups swapped: B on the older waves) %.0f\n", names[K],        \
               run<K, false, true, false>(nseg, nm, nv, out, cyc), run<K, true, true, false>(nseg, nm, nv, out, cyc), \
               run<K, true, true, true>(nseg, nm, nv, out, cyc));
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7)
        // the same with ONE dependent accumulator chain and with four independent ones in group A (v_fma_f32 and v_exp_f32 in group B)
#define ROWCH(K, C)                                                                                                         \
        printf("   chains=%d %-14s A only %.0f   B only %.0f   A+B %.0f   swapped %.0f\n", C, names[K],                       \
               run<K, true, false, false, C>(nseg, nm, nv, out, cyc), run<K, false, true, false, C>(nseg, nm, nv, out, cyc), \
               run<K, true, true, false, C>(nseg, nm, nv, out, cyc), run<K, true, true, true, C>(nseg, nm, nv, out, cyc));
        ROWCH(0, 1) ROWCH(1, 1) ROWCH(0, 4) ROWCH(1, 4)
    }
    return 0;
}
