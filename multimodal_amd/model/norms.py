"""LayerNorm of the layer (megatron/model/norms.py:16-31) on the HIP kernel; RMSNorm/ScaleNorm are out of scope
(no listed config uses them)."""
import torch

from .. import functional as Fn


class LayerNorm(torch.nn.Module):
    """Same parameters (`weight`, `bias`) and semantics as torch.nn.LayerNorm over the last dimension."""

    def __init__(self, normalized_shape, eps=1e-5):
        super().__init__()
        if not isinstance(normalized_shape, int):
            (normalized_shape,) = tuple(normalized_shape)
        self.normalized_shape = (normalized_shape,)
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.ones(normalized_shape))
        self.bias = torch.nn.Parameter(torch.zeros(normalized_shape))

    def forward(self, x):
        return Fn.layer_norm(x, self.weight, self.bias, self.eps)


def get_norm(neox_args):
    """megatron/model/norms.py:19-31."""
    if neox_args.norm == "layernorm":
        return LayerNorm, neox_args.layernorm_epsilon
    raise NotImplementedError(f"norm {neox_args.norm}: only layernorm is on the accelerated path")
