/* mm_hip_dev.h - entry points of the DEVELOPER library only (make DEV=1 -> libmm_hip_dev.so; never in libmm_hip.so, never in
 * include/mm_hip.h): hardware probes and kernel decomposition variants used by tests/ and tools/.  The developer library also
 * honours MM_GEMMPP / MM_GEMM256 (force a GEMM kernel for same-process A/B and for tests that must reach a kernel at small
 * shapes) and carries the superseded 256x256 pipelined kernel (gemm256.hip). */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ test-only hardware probe
 * One wave reads a [16][64] 16-bit LDS tile with ds_read_b64_tr_b16; out: [64 lanes][4].  No reference counterpart. */
int mm_debug_probe_tr_read(void* out, const void* in, const int* row0, const int* col0, void* stream);
/* Developer decomposition of the 256x256 GEMM kernel (tools/decompose_gemm256.py): bf16, contiguous operands, M and N multiples
 * of 256; dbg bit 0 = no global->LDS loads in the K loop, bit 1 = no LDS fragment reads, bit 2 = no epilogue, bit 3 = no MFMA.
 * Results are meaningless unless dbg == 0.  No reference counterpart. */
int mm_debug_gemm256(void* C, void* aux, const void* A, const void* B, const void* bias, int M, int N, int K, int gelu, int dbg,
                     void* stream);

#ifdef __cplusplus
}
#endif
