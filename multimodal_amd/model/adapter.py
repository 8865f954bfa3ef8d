"""MAGMA bottleneck adapters around `.attention` / `.mlp` (megatron/model/adapter.py:6-151).

Same wrapping protocol and state-dict keys as the reference (`<name>.attn_block.*`, `<name>.adapter.{0,2}.*`,
adapter.py:106-107,25): the wrapped module returns `(output, bias)`; the adapter acts on the pre-bias,
pre-all-reduce `output`.  The two small GEMMs run on the MFMA kernel with fused bias+ReLU and bias+residual epilogues.
"""
import torch
import torch.nn as nn

from .. import functional as Fn


class Adapter(nn.Module):
    def __init__(self, dim, downsample_factor=4, activation=nn.ReLU, add_layernorm=False):
        super().__init__()
        if add_layernorm or activation is not nn.ReLU:
            raise NotImplementedError("only the Linear-ReLU-Linear adapter the reference's configs use is accelerated")
        self.adapter = nn.Sequential(nn.Linear(dim, dim // downsample_factor), activation(),
                                     nn.Linear(dim // downsample_factor, dim))
        self.adapter.apply(self.init_weights)

    def init_weights(self, m, std=1e-3):
        """adapter.py:28-36."""
        if isinstance(m, nn.Linear):
            torch.nn.init.normal_(m.weight, std=std)
            torch.nn.init.normal_(m.bias, std=std)
            m.weight.data = torch.clamp(m.weight.data, min=-2 * std, max=2 * std)
            m.bias.data = torch.clamp(m.bias.data, min=-2 * std, max=2 * std)

    def forward(self, x):
        a0, a2 = self.adapter[0], self.adapter[2]
        return Fn.adapter(x, a0.weight, a0.bias, a2.weight, a2.bias)


class AdapterWrapper(Adapter):
    """adapter.py:95-117."""

    def __init__(self, attn_block, dim, downsample_factor=4, activation=nn.ReLU, add_layernorm=False):
        super().__init__(dim, downsample_factor, activation, add_layernorm)
        self.attn_block = attn_block

    def forward(self, x, *attn_args, **attn_kwargs):
        attn_outputs = self.attn_block(x, *attn_args, **attn_kwargs)
        attn_output, outputs = attn_outputs[0], attn_outputs[1:]
        return (Adapter.forward(self, attn_output),) + tuple(outputs)


def add_adapters(neox_args, model, downsample_factor=4, location="mlp", ff_attr="mlp", attn_attr="attention",
                 **adapter_kwargs):
    """Swap `.mlp` / `.attention` of every layer for an AdapterWrapper (adapter.py:119-151)."""
    for names, module in list(model.named_modules()):
        if "image_prefix" in names:
            continue
        child_names = [name for name, _ in module.named_children()]
        attr = ff_attr if location == ff_attr else attn_attr
        if location in child_names and not isinstance(getattr(module, attr), AdapterWrapper):
            inner = getattr(module, attr)
            p = next(inner.parameters())
            wrapped = AdapterWrapper(attn_block=inner, dim=neox_args.hidden_size, downsample_factor=downsample_factor,
                                     **adapter_kwargs)
            wrapped.adapter.to(device=p.device, dtype=p.dtype)
            setattr(module, attr, wrapped)
    return model
