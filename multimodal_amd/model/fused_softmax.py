"""scale + mask + softmax over attention scores: the operator interface of megatron/model/fused_softmax.py:24-205
(`FusedScaleMaskSoftmax`, `SoftmaxFusionTypes`, the two autograd Functions) on the HIP softmax kernels.

Two drop-in levels exist:
  level 1 - the reference's OWN fused_softmax.py keeps working unchanged once `multimodal_amd.fused_kernels.
            load_fused_kernels()` has registered the modules it imports by name;
  level 2 - this module: same class / method names and the same eligibility rule, calling the C ABI through
            `multimodal_amd.ops` directly (no import-by-name, the scale stays a Python float).
Both are exercised by tests/test_fused_kernels_shim_gpu.py.
"""
import enum

import torch
import torch.nn as nn

from .. import ops

MAX_FUSED_KEYS = 2048   # the kernels hold one row in registers (scaled_masked_softmax_cuda.cu:44, fused_softmax.py:155)


class _InPlaceSoftmaxBackward:
    """Shared backward contract of both kernels (…_cuda.cu:77-105): the gradient is written over `output_grads`
    (a contiguous copy of it when it was strided) and that tensor is what autograd receives."""

    @staticmethod
    def run(kernel, ctx, output_grads):
        (probs,) = ctx.saved_tensors
        return kernel(output_grads, probs, ctx.softmax_scale)


class ScaledUpperTriangMaskedSoftmax(torch.autograd.Function):
    """Causal: row r of every [s, s] matrix is softmax(scale * x[r, :r+1]), zeros above the diagonal (fused_softmax.py:24-53)."""

    @staticmethod
    def forward(ctx, inputs, scale):
        ctx.softmax_scale = float(scale)
        probs = ops.softmax_causal_fwd(inputs, ctx.softmax_scale)
        ctx.save_for_backward(probs)
        return probs

    @staticmethod
    def backward(ctx, output_grads):
        return _InPlaceSoftmaxBackward.run(ops.softmax_causal_bwd, ctx, output_grads), None


class ScaledMaskedSoftmax(torch.autograd.Function):
    """Explicit byte mask [1|b, 1, sq, sk], 1 = masked (-10000 fill) (fused_softmax.py:56-83)."""

    @staticmethod
    def forward(ctx, inputs, mask, scale):
        ctx.softmax_scale = float(scale)
        probs = ops.softmax_masked_fwd(inputs, mask, ctx.softmax_scale)
        ctx.save_for_backward(probs)
        return probs

    @staticmethod
    def backward(ctx, output_grads):
        return _InPlaceSoftmaxBackward.run(ops.softmax_masked_bwd, ctx, output_grads), None, None


class SoftmaxFusionTypes(enum.Enum):
    upper_triang = 1  # causal mask
    general = 2  # general mask
    none = 3  # no fusion


def fused_kernel_eligible(fusion_type, sixteen_bit, has_mask, b, np, sq, sk):
    """The reference's availability rule (fused_softmax.py:148-168) as a pure function of the configuration and shape.
    `batch_per_block` comes from the library (wave64 value, scaled_masked_softmax.h:338-352 with C10_WARP_SIZE = 64)."""
    if fusion_type == SoftmaxFusionTypes.none or not sixteen_bit or not has_mask:
        return False
    if not (16 < sk <= MAX_FUSED_KEYS) or sq % 4 or (b * np) % 4:
        return False
    per_block = ops.softmax_batch_per_block(sq, sk, b, np)
    rows_or_batches = b * np if fusion_type == SoftmaxFusionTypes.upper_triang else sq
    return rows_or_batches % per_block == 0


class FusedScaleMaskSoftmax(nn.Module):
    """Constructor and methods as the reference's (fused_softmax.py:92-205).  Shapes the reference's eligibility rule sends
    to its eager sequence (`forward_torch_softmax`) run on the HIP masked kernel too; what that kernel cannot take raises -
    the module computes nothing outside libmm_hip.so."""

    def __init__(self, input_in_fp16, input_in_bf16, fusion_type, mask_func, softmax_in_fp32, scale):
        super().__init__()
        if not isinstance(fusion_type, SoftmaxFusionTypes):
            raise AssertionError(f"Invalid fusion type {fusion_type}")
        if scale is not None and not softmax_in_fp32:
            raise AssertionError("softmax should be in fp32 when scaled")
        self.input_in_fp16, self.input_in_bf16 = input_in_fp16, input_in_bf16
        self.input_in_float16 = input_in_fp16 or input_in_bf16
        self.fusion_type = fusion_type
        self.fusion = fusion_type != SoftmaxFusionTypes.none
        self.upper_triang_mask_fusion = fusion_type == SoftmaxFusionTypes.upper_triang
        self.general_mask_fusion = fusion_type == SoftmaxFusionTypes.general
        self.mask_func, self.softmax_in_fp32, self.scale = mask_func, softmax_in_fp32, scale
        if self.fusion:
            from ..fused_kernels import load_fused_kernels
            load_fused_kernels()

    def forward(self, input, mask):
        if input.dim() != 4:
            raise AssertionError("expected [b, np, sq, sk] attention scores")
        fused = self.is_kernel_available(mask, *input.size())
        return (self.forward_fused_softmax if fused else self.forward_torch_softmax)(input, mask)

    def is_kernel_available(self, mask, b, np, sq, sk):
        return fused_kernel_eligible(self.fusion_type, self.input_in_float16, mask is not None, b, np, sq, sk)

    def forward_fused_softmax(self, input, mask):
        b, np, sq, sk = input.size()
        scale = 1.0 if self.scale is None else self.scale
        if self.upper_triang_mask_fusion:
            if sq != sk:
                raise AssertionError("causal mask is only for self attention")
            # the causal kernel masks by position and never reads `mask` (fused_softmax.py:173-179)
            return ScaledUpperTriangMaskedSoftmax.apply(input.view(b * np, sq, sk), scale).view(b, np, sq, sk)
        return ScaledMaskedSoftmax.apply(input, mask, scale)

    def forward_torch_softmax(self, input, mask):
        """The reference's eager branch (fused_softmax.py:184-199: optional fp32 up-cast, scale, `mask_func` = fill with
        -10000, Softmax(dim=-1), cast back) for shapes its fused-kernel rule rejects.  Here it runs on the SAME HIP masked
        kernel, which has no divisibility requirement of its own: fp32 statistics on the 16-bit scores, scale applied
        before the -10000 fill - the eager sequence's arithmetic with softmax_in_fp32 (and what a GPU's 16-bit softmax
        computes internally without it).  Nothing in this package computes outside libmm_hip.so: inputs the kernel cannot
        take (fp32 scores, more than 2048 keys, a single query row - the kv-cache decode shape, out of scope) raise.
        Unlike the reference's in-place masked_fill_ the scores tensor is left untouched."""
        b, np, sq, sk = input.size()
        if not self.input_in_float16 or input.dtype not in (torch.float16, torch.bfloat16):
            raise RuntimeError("FusedScaleMaskSoftmax: only 16-bit attention scores are on the HIP path (no eager fallback)")
        if sk > MAX_FUSED_KEYS or sq < 2:
            raise RuntimeError(f"FusedScaleMaskSoftmax: [sq={sq}, sk={sk}] is outside the HIP softmax kernel's range "
                               f"(2 <= sq, sk <= {MAX_FUSED_KEYS}); there is no eager fallback")
        if mask is None:
            mask = torch.zeros(1, 1, sq, sk, dtype=torch.uint8, device=input.device)  # nothing masked
        return ScaledMaskedSoftmax.apply(input, mask, 1.0 if self.scale is None else self.scale)

    @staticmethod
    def get_batch_per_block(sq, sk, b, np):
        return ops.softmax_batch_per_block(sq, sk, b, np)
