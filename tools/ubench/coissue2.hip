// Developer microbenchmark (round 3): how much of a SIMD's vector-issue time does an MFMA stream of ANOTHER wave take away?
// coissue.hip's vector loops carried ~60 % loop overhead (8 instructions per trip), so "A+B = max(A, B)" there did not show that
// the port was shared without loss.  Here group B's segment is straight-line code: NV independent vector instructions over 16
// registers (no loop inside a segment), group A issues 16 MFMAs 32x32x16 in CH accumulator chains, and every segment ends at a
// workgroup barrier.  512-thread workgroups, one per CU: waves w and w + 4 share a SIMD.  MODE 2 runs both streams in the SAME
// wave (256-thread workgroups), the MFMAs placed one per NV/16 vector instructions by the source order.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int KIND> __device__ __forceinline__ void vop(float (&v)[16], int j) {
    float& x = v[j & 15];
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(v[(j + 5) & 15]), "v"(v[(j + 9) & 15]));
    else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    else if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(v[(j + 5) & 15]), "v"(v[(j + 9) & 15]));
    else if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(v[(j + 5) & 15]));
    else if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<f32x2*>(&v[(2 * j) & 14])) : "v"(*reinterpret_cast<f32x2*>(&v[(2 * j + 6) & 14])));
    else if (KIND == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(v[(j + 5) & 15]));
    else if (KIND == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(v[(j + 5) & 15]));
}

template <int KIND, int NV, int CH, int MODE, bool DO_A, bool DO_B>
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int nseg) {
    const int wave = threadIdx.x >> 6;
    const bool is_a = MODE == 2 ? true : wave < 4, is_b = MODE == 2 ? true : wave >= 4;
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.5f + 0.001f * (threadIdx.x + 3 * i);
    const long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < nseg; ++s) {
        if (MODE == 2) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (DO_A) acc[m % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % CH], 0, 0, 0);
                if (DO_B) {
#pragma unroll
                    for (int j = 0; j < NV / 16; ++j) vop<KIND>(v, m * (NV / 16) + j);
                }
            }
        } else {
            if (is_a && DO_A) {
#pragma unroll
                for (int m = 0; m < 16; ++m) acc[m % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % CH], 0, 0, 0);
            }
            if (is_b && DO_B) {
#pragma unroll
                for (int j = 0; j < NV; ++j) vop<KIND>(v, j);
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int c = 0; c < 4; ++c) r += acc[c][c];
    for (int i = 0; i < 16; ++i) r += v[i];
    if (r == 1234.5678f) out[threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NV, int CH, int MODE, bool A, bool B> double run(float* out, long long* cyc) {
    const int nseg = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<KIND, NV, CH, MODE, A, B>), dim3(256), dim3(MODE == 2 ? 256 : 512), 0, 0, out, cyc, nseg);
        (void)hipDeviceSynchronize();
    }
    long long h;
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    return (double)h / nseg;
}

template <int KIND, int NV> void row(const char* name, float* out, long long* cyc) {
    printf("%-16s NV=%3d | two waves: B only %5.0f  A(1 chain)+B %5.0f  A(2)+B %5.0f  A(4)+B %5.0f | one wave, interleaved: B only %5.0f  A(1)+B %5.0f  A(2)+B %5.0f  A(4)+B %5.0f\n",
           name, NV, run<KIND, NV, 1, 0, false, true>(out, cyc), run<KIND, NV, 1, 0, true, true>(out, cyc),
           run<KIND, NV, 2, 0, true, true>(out, cyc), run<KIND, NV, 4, 0, true, true>(out, cyc),
           run<KIND, NV, 1, 2, false, true>(out, cyc), run<KIND, NV, 1, 2, true, true>(out, cyc),
           run<KIND, NV, 2, 2, true, true>(out, cyc), run<KIND, NV, 4, 2, true, true>(out, cyc));
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 8);
    printf("cycles per segment (16 MFMA 32x32x16 bf16 in stream A = 512 matrix-pipe cycles; barrier per segment)\n");
    printf("A only: 1 chain %.0f  2 chains %.0f  4 chains %.0f (two-wave layout) | one-wave layout 1 chain %.0f  4 chains %.0f\n",
           run<0, 64, 1, 0, true, false>(out, cyc), run<0, 64, 2, 0, true, false>(out, cyc), run<0, 64, 4, 0, true, false>(out, cyc),
           run<0, 64, 1, 2, true, false>(out, cyc), run<0, 64, 4, 2, true, false>(out, cyc));
    row<0, 64>("v_fma_f32", out, cyc);
    row<0, 128>("v_fma_f32", out, cyc);
    row<0, 192>("v_fma_f32", out, cyc);
    row<1, 32>("v_exp_f32", out, cyc);
    row<1, 64>("v_exp_f32", out, cyc);
    row<2, 128>("v_max3_f32", out, cyc);
    row<3, 128>("v_cvt_pk_bf16", out, cyc);
    row<4, 64>("v_pk_fma_f32", out, cyc);
    row<5, 128>("v_add_f32", out, cyc);
    row<6, 128>("v_mul_f32", out, cyc);
    return 0;
}
