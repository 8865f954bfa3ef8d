// debug.hip - hardware-semantics probes used only by tests (tests/test_hw_probes.py).
#include "common.h"
#include "mm_hip_dev.h"

namespace mm {
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// One wave.  tile: [rows=16][cols=64] of 16-bit values in LDS.  Lane (g = lane>>4, i = lane&15, q = i>>2, p = i&3)
// supplies the address of element [row0(g) + q][col0(g) + 4p] and receives 4 values.
__global__ void probe_tr_read_k(short* __restrict__ out, const short* __restrict__ in, const int* __restrict__ row0,
                                const int* __restrict__ col0) {
    __shared__ __attribute__((aligned(16))) short tile[16 * 64];
    for (int i = threadIdx.x; i < 16 * 64; i += 64) tile[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const short* addr = &tile[(row0[g] + q) * 64 + col0[g] + 4 * p];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)addr);
#pragma unroll
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
}  // namespace mm

extern "C" int mm_debug_probe_tr_read(void* out, const void* in, const int* row0, const int* col0, void* stream) {
    MM_CHECK_ARG(out && in && row0 && col0, "mm_debug_probe_tr_read: null pointer");
    mm::probe_tr_read_k<<<1, 64, 0, (hipStream_t)stream>>>((short*)out, (const short*)in, row0, col0);
    MM_CHECK_LAUNCH();
    return MM_OK;
}

namespace mm {
int gemm256_debug(void* C, void* aux, const void* A, const void* B, const void* bias, int M, int N, int K, int gelu, int dbg,
                  hipStream_t st);
int gemm_pp_debug(void* C, void* aux, const void* A, const void* B, const void* bias, int M, int N, int K, int gelu, int dbg,
                  hipStream_t st);
}
extern "C" int mm_debug_gemm256(void* C, void* aux, const void* A, const void* B, const void* bias, int M, int N, int K, int gelu,
                                int dbg, void* stream) {
    MM_CHECK_ARG(C && A && B && (!gelu || (aux && bias)), "mm_debug_gemm256: null pointer");
    // dbg 0..15: the pipelined kernel of gemm256.hip; 16 + d: the ping-pong kernel of gemm_pp.hip with decomposition bits d
    const int rc = dbg >= 16 ? mm::gemm_pp_debug(C, aux, A, B, bias, M, N, K, gelu, dbg - 16, (hipStream_t)stream)
                             : mm::gemm256_debug(C, aux, A, B, bias, M, N, K, gelu, dbg, (hipStream_t)stream);
    if (rc != MM_OK) mm::set_error("mm_debug_gemm256: unsupported shape / variant (M=%d N=%d K=%d dbg=%d)", M, N, K, dbg);
    return rc;
}
