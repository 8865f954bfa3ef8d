"""Drop-in for megatron/fused_kernels: the two softmax extension modules.

The reference builds two pybind11 CUDA extensions and imports them BY NAME from anywhere
(`import scaled_upper_triang_masked_softmax_cuda`, megatron/model/fused_softmax.py:34,46,66,76,203).
`load_fused_kernels()` (reference: megatron/fused_kernels/__init__.py:32-45) registers modules with those names
and the same function signatures, backed by the C ABI in libmm_hip.so (include/mm_hip.h).  Like the reference it
fails loudly if the native library is missing.
"""
import sys
import types

from .. import ops


def _upper_triang_module():
    m = types.ModuleType("scaled_upper_triang_masked_softmax_cuda")
    m.__doc__ = "HIP (gfx950) replacement of megatron/fused_kernels/scaled_upper_triang_masked_softmax.cpp:62-70"

    def forward(input, scale):
        """input [attn_batches, s, s] fp16/bf16 -> new tensor (scaled_upper_triang_masked_softmax.cpp:31-42)."""
        return ops.softmax_causal_fwd(input, float(scale))

    def backward(output_grads, softmax_results, scale):
        """In place into output_grads, which is returned (…_cuda.cu:59-88)."""
        return ops.softmax_causal_bwd(output_grads, softmax_results, float(scale))

    m.forward, m.backward = forward, backward
    return m


def _masked_module():
    m = types.ModuleType("scaled_masked_softmax_cuda")
    m.__doc__ = "HIP (gfx950) replacement of megatron/fused_kernels/scaled_masked_softmax.cpp:70-83"

    def forward(input, mask, scale):
        """input [b, np, sq, sk], mask [b|1, 1, sq, sk] (1 = masked) (scaled_masked_softmax.cpp:33-47)."""
        return ops.softmax_masked_fwd(input, mask, float(scale))

    def backward(output_grads, softmax_results, scale):
        return ops.softmax_masked_bwd(output_grads, softmax_results, float(scale))

    def get_batch_per_block(query_seq_len, key_seq_len, batches, attn_heads):
        return ops.softmax_batch_per_block(query_seq_len, key_seq_len, batches, attn_heads)

    m.forward, m.backward, m.get_batch_per_block = forward, backward, get_batch_per_block
    return m


def load_fused_kernels():
    """Make `import scaled_upper_triang_masked_softmax_cuda` / `import scaled_masked_softmax_cuda` resolve to the
    HIP implementations.  Raises RuntimeError if libmm_hip.so is not built (the reference calls exit())."""
    from .._lib import get_lib
    get_lib()
    sys.modules.setdefault("scaled_upper_triang_masked_softmax_cuda", _upper_triang_module())
    sys.modules.setdefault("scaled_masked_softmax_cuda", _masked_module())
