"""The entry point that allocates nothing (mm_gemm_tn_static, include/mm_hip.h): its output must equal mm_gemm_tn's
(same kernels, same fixed summation order => bit-exact) on two calls in a row, the second with a smaller input, all
through the C ABI.  It has no counterpart in the reference; it serves callers that cannot allocate device memory."""
import ctypes

import pytest
import torch

from multimodal_amd import _lib

pytestmark = pytest.mark.gpu


def _call_pair(lib, M, N, Kp, seed, dtype=torch.bfloat16, with_colsum=True):
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = torch.randn(M, N, generator=g).to(dtype).cuda()
    B = torch.randn(M, Kp, generator=g).to(dtype).cuda()
    splits = lib.mm_gemm_tn_splits(M, N, Kp)
    assert splits > 1, "the test must exercise the scratch buffer"
    slab = N * Kp + (N if with_colsum else 0)
    assert splits * slab <= _lib.GEMM_TN_STATIC_MAX_FLOATS
    st = torch.cuda.current_stream().cuda_stream
    code = _lib.dtype_code(dtype)
    # existing entry point, caller-provided workspace
    C_ref = torch.empty(N, Kp, dtype=dtype, device="cuda")
    cs_ref = torch.empty(N, dtype=dtype, device="cuda") if with_colsum else None
    ws = torch.empty(splits * slab, dtype=torch.float32, device="cuda")
    _lib.check(lib.mm_gemm_tn(C_ref.data_ptr(), cs_ref.data_ptr() if with_colsum else None, ws.data_ptr(), A.data_ptr(),
                              B.data_ptr(), M, N, Kp, N, Kp, Kp, splits, code, st), "mm_gemm_tn")
    # the new one: outputs pre-filled with garbage, no workspace
    C = torch.full((N, Kp), 7.0, dtype=dtype, device="cuda")
    cs = torch.full((N,), 7.0, dtype=dtype, device="cuda") if with_colsum else None
    free_before = torch.cuda.mem_get_info()[0]
    _lib.check(lib.mm_gemm_tn_static(C.data_ptr(), cs.data_ptr() if with_colsum else None, A.data_ptr(), B.data_ptr(), M, N, Kp,
                                     N, Kp, Kp, splits, code, st), "mm_gemm_tn_static")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free_before - (1 << 20), "the call must not allocate device memory"
    assert torch.equal(C.view(torch.int16), C_ref.view(torch.int16))
    if with_colsum:
        assert torch.equal(cs.view(torch.int16), cs_ref.view(torch.int16))
    # sanity against fp32 math so that "both wrong the same way" cannot pass
    ref = A.float().t() @ B.float()
    assert (C.float() - ref).abs().max() <= 2.0 ** -7 * ref.abs().max() + 1e-3


def test_static_scratch_matches_workspace_entry_point_twice():
    lib = _lib.get_lib()
    _call_pair(lib, 8192, 128, 1024, seed=1)             # adapter W1 gradient shape: 32 splits x 131200 fp32
    _call_pair(lib, 1024, 128, 128, seed=2)              # second call, smaller: stale scratch contents must not leak in
    _call_pair(lib, 2048, 256, 128, seed=3, dtype=torch.float16, with_colsum=False)


def test_static_scratch_rejects_oversize_and_bad_arguments():
    lib = _lib.get_lib()
    M, N, Kp, splits = 8192, 1024, 1024, 8               # 8 x (1 Mi + 1 Ki) fp32 > 6 Mi fp32
    A = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    C = torch.zeros(N, Kp, dtype=torch.bfloat16, device="cuda")
    rc = lib.mm_gemm_tn_static(C.data_ptr(), None, A.data_ptr(), A.data_ptr(), M, N, Kp, N, Kp, Kp, splits, _lib.MM_BF16,
                               torch.cuda.current_stream().cuda_stream)
    assert rc == -1 and b"exceeds the built-in scratch" in lib.mm_last_error()
    rc = lib.mm_gemm_tn_static(None, None, A.data_ptr(), A.data_ptr(), M, N, Kp, N, Kp, Kp, 1, _lib.MM_BF16, None)
    assert rc == -1 and b"null" in lib.mm_last_error()
    # splits == 1 needs no scratch at all and works for any size the kernel takes
    _lib.check(lib.mm_gemm_tn_static(C.data_ptr(), None, A.data_ptr(), A.data_ptr(), M, N, Kp, N, Kp, Kp, 1, _lib.MM_BF16,
                                     torch.cuda.current_stream().cuda_stream), "splits=1")
    torch.cuda.synchronize()
    assert float(C.float().abs().max()) == 0.0
