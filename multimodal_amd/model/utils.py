"""Small helpers of the model code that sit on the path (megatron/model/utils.py:326-332, gpt2_model.py:47-49,
megatron/utils.py:66-76)."""
import torch

from .fused_softmax import SoftmaxFusionTypes


def exists(x):
    return x is not None


def get_fusion_type(neox_args):
    """megatron/model/utils.py:326-332."""
    fusion_type = SoftmaxFusionTypes.none
    if neox_args.scaled_upper_triang_masked_softmax_fusion:
        fusion_type = SoftmaxFusionTypes.upper_triang
    elif neox_args.scaled_masked_softmax_fusion:
        fusion_type = SoftmaxFusionTypes.general
    return fusion_type


def gpt2_attention_mask_func(attention_scores, ltor_mask):
    """megatron/model/gpt2_model.py:47-49."""
    attention_scores.masked_fill_(ltor_mask, -10000.0)
    return attention_scores


def get_attn_mask(seq_length, device):
    """megatron/utils.py:66-76: [1, 1, s, s] bool, True = masked (plain causal)."""
    mask = torch.tril(torch.ones((1, seq_length, seq_length), device=device)).view(1, 1, seq_length, seq_length)
    return mask < 0.5
