"""Activations on the path (megatron/model/activations.py:27-96): tanh bias-GeLU with the reference's custom backward,
and exact GeLU.  Inside ParallelMLP they are fused into the GEMM epilogue; these standalone forms keep the
reference's callable surface."""
import torch

from .. import ops


class GeLUFunction(torch.autograd.Function):
    """megatron/model/activations.py:82-93: returns the same gradient for input and bias."""

    @staticmethod
    def forward(ctx, input, bias):
        ctx.save_for_backward(input, bias)
        return ops.bias_gelu_fwd(input, bias, 0)

    @staticmethod
    def backward(ctx, grad_output):
        input, bias = ctx.saved_tensors
        tmp = ops.bias_gelu_bwd(grad_output, input, bias, 0)
        db = ops.colsum(tmp.reshape(-1, tmp.size(-1)))
        return tmp, db


bias_gelu_impl = GeLUFunction.apply


class _ErfGelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        zero = torch.zeros(x.size(-1), dtype=x.dtype, device=x.device)
        ctx.save_for_backward(x, zero)
        return ops.bias_gelu_fwd(x, zero, 1)

    @staticmethod
    def backward(ctx, g):
        x, zero = ctx.saved_tensors
        return ops.bias_gelu_bwd(g, x, zero, 1)


def gelu(x):
    return _ErfGelu.apply(x)


def get_activation(neox_args):
    """megatron/model/activations.py:27-50 for the activations on the accelerated path."""
    if neox_args.activation == "gelu":
        if getattr(neox_args, "onnx_safe", False):
            raise NotImplementedError("onnx_safe is outside this path's scope")
        return bias_gelu_impl if neox_args.bias_gelu_fusion else gelu
    raise NotImplementedError(f"activation {neox_args.activation}: only gelu is on the accelerated path")
