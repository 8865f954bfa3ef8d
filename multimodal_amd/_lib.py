"""ctypes binding of ``libmm_hip.so`` (the C ABI declared in ``include/mm_hip.h``).

The library is the product: there is NO fallback.  If it is missing, or a call returns an
error code, a ``RuntimeError`` is raised (the reference's pybind modules raise ``RuntimeError``
through ``AT_ASSERTM`` the same way, megatron/fused_kernels/scaled_masked_softmax.cpp:34-37).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmm_hip.so")

MM_BF16, MM_F16 = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU_TANH, EPI_BIAS_GELU_ERF, EPI_BIAS_RELU = 0, 1, 2, 3, 4
GEMM_TN_STATIC_MAX_FLOATS = 6 << 20  # MM_GEMM_TN_STATIC_MAX_FLOATS
EPI_RELU_BWD, EPI_GELU_TANH_BWD, EPI_GELU_ERF_BWD = 5, 6, 7   # backward epilogues: saved forward tensor in the residual slot

c_int, c_float, c_void_p, c_size_t, c_u64 = (
    ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64)

# name -> (restype, argtypes); must list every symbol include/mm_hip.h declares
# (tests/test_abi.py cross-checks this table against the header).
PROTOTYPES = {
    "mm_last_error": (ctypes.c_char_p, []),
    "mm_set_cu_share": (ctypes.c_int, [ctypes.c_int]),
    "mm_target_arch": (ctypes.c_char_p, []),
    "mm_softmax_causal_fwd": (c_int, [c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p]),
    "mm_softmax_causal_bwd": (c_int, [c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p]),
    "mm_softmax_masked_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mm_softmax_masked_bwd": (c_int, [c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mm_softmax_batch_per_block": (c_int, [c_int, c_int, c_int, c_int]),
    "mm_layernorm_fwd": (c_int, [c_void_p] * 9 + [c_int, c_int, c_float, c_int, c_void_p]),
    "mm_layernorm_bwd_nblk": (c_int, [c_int]),
    "mm_layernorm_bwd": (c_int, [c_void_p] * 10 + [c_int, c_int, c_int, c_void_p]),
    "mm_layernorm_bwd_finalize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mm_layernorm_fwd_sb": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "mm_layernorm_bwd_sb": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_int, c_void_p]),
    "mm_rotary_qk_inplace": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "mm_bias_gelu_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mm_bias_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mm_bias_dropout_add": (c_int, [c_void_p] * 5 + [c_int, c_int, c_float, c_int, c_u64, c_u64, c_int, c_void_p]),
    "mm_bias_dropout_add_devrng": (c_int, [c_void_p] * 5 + [c_int, c_int, c_float, c_int, c_void_p, c_u64, c_int, c_void_p]),
    "mm_rng_advance": (c_int, [c_void_p, c_u64, c_void_p]),
    "mm_colsum_nblk": (c_int, [c_int]),
    "mm_colsum": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mm_transpose2d": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mm_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "mm_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "mm_gptj_combine": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p]),
    "mm_gemm_nt": (c_int, [c_void_p] * 6 + [c_int] * 8 + [c_void_p]),
    "mm_gemm_nt_splitk_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mm_gemm_nt_splitk": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    "mm_gemm_nt_kernel_name": (ctypes.c_char_p, [c_int, c_int, c_int, c_int]),
    "mm_gemm_nt_qkv_rotary": (c_int, [c_void_p] * 6 + [c_int] * 11 + [c_void_p]),
    "mm_gemm_tn_splits": (c_int, [c_int, c_int, c_int]),
    "mm_adapter_supported": (c_int, [c_int, c_int]),
    "mm_adapter_fwd": (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_void_p]),
    "mm_adapter_fwd_combine": (c_int, [c_void_p] * 11 + [c_int] * 4 + [c_void_p]),
    "mm_adapter_bwd_dx": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p]),
    "mm_gemm_tn_pair_splits": (c_int, [c_int] * 5),
    "mm_gemm_tn_pair": (c_int, [c_void_p] * 4 + [c_int] * 2 + [c_void_p] * 4 + [c_int] * 3 + [c_void_p] + [c_int] * 2 + [c_void_p]),
    "mm_gemm_tn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "mm_gemm_tn_static": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "mm_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "mm_attn_bwd": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_float, c_int, c_void_p]),
    "mm_attn_bwd_rotary": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mm_attn_fwd_dropout": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_float, c_float, c_void_p, ctypes.c_ulonglong, c_int, c_void_p]),
    "mm_attn_bwd_dropout": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_float, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p,
                                    ctypes.c_ulonglong, c_int, c_void_p]),
    "mm_attn_dropout_mask": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, ctypes.c_ulonglong, c_void_p]),
    "mm_cross_entropy_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mm_cross_entropy_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mm_cross_entropy_fwd_grad": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_void_p]),
    "mm_cross_entropy_fwd_grad_max_v": (c_int, []),
    "mm_loss_mask_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mm_masked_mean": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mm_scale_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "mm_embed_prefix": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "mm_prefix_tail_supported": (c_int, [c_int, c_int, c_int]),
    "mm_prefix_tail_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mm_prefix_tail_fwd": (c_int, [c_void_p] * 11 + [c_int, c_int, c_int, c_float, c_float, c_int, c_void_p, c_u64, c_int, c_void_p]),
    "mm_dropout_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_int, c_void_p]),
    "mm_adam_chunk_elems": (c_int, []),
    "mm_adam_step": (c_int, [c_void_p] * 6 + [c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p] + [c_float] * 5 + [c_int, c_int, c_void_p]),
}

# developer library only (csrc/mm_hip_dev.h, `make DEV=1`): never exported by libmm_hip.so
DEV_PROTOTYPES = {
    "mm_debug_gemm256": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "mm_debug_probe_tr_read": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
}
DEV_LIB_PATH = os.path.join(_HERE, "libmm_hip_dev.so")

_lib = None


def get_lib():
    """Load libmm_hip.so once; raise loudly if it is not built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C multimodal_amd/csrc`). multimodal_amd has no fallback path.")
        # torch bundles its own libamdhip64.so.7; import it FIRST so that libmm_hip.so binds to the very same HIP
        # runtime (one runtime per process: streams and device pointers are shared with torch)
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _bind(path, protos):
    import torch  # noqa: F401  (one HIP runtime per process: see get_lib)
    lib = ctypes.CDLL(path)
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


class use_dev_lib:
    """TESTS AND TOOLS ONLY: route every call of this package through the developer library (libmm_hip_dev.so, `make DEV=1`:
    the product's sources plus the MM_GEMMPP / MM_GEMM256 kernel-forcing switches, the superseded 256x256 GEMM and the
    mm_debug_* probes) - as a context manager, or permanently with use_dev_lib().enter().  The product never calls this."""

    def __init__(self):
        if not os.path.exists(DEV_LIB_PATH):
            raise RuntimeError(f"{DEV_LIB_PATH} is missing: build it with `make -C multimodal_amd/csrc DEV=1`")
        self._dev = _bind(DEV_LIB_PATH, {**PROTOTYPES, **DEV_PROTOTYPES})
        self._prev = None

    def enter(self):
        global _lib
        self._prev, _lib = _lib, self._dev
        return self._dev

    __enter__ = enter

    def __exit__(self, *a):
        global _lib
        _lib = self._prev
        return False


def check(rc, what=""):
    if rc != 0:
        msg = get_lib().mm_last_error().decode(errors="replace")
        raise RuntimeError(f"multimodal_amd: {what or 'call'} failed (code {rc}): {msg}")


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.bfloat16:
        return MM_BF16
    if torch_dtype == torch.float16:
        return MM_F16
    raise RuntimeError(f"multimodal_amd: only fp16 and bf16 are supported on the HIP path, got {torch_dtype}")
