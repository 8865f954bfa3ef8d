from .adapter import Adapter, AdapterWrapper, add_adapters  # noqa: F401
from .fused_softmax import FusedScaleMaskSoftmax, SoftmaxFusionTypes  # noqa: F401
from .norms import LayerNorm, get_norm  # noqa: F401
from .positional_embeddings import RotaryEmbedding  # noqa: F401
from .transformer import (ParallelMLP, ParallelSelfAttention, ParallelTransformerLayer,  # noqa: F401
                          ParallelTransformerLayerPipe)
from .utils import get_attn_mask, get_fusion_type, gpt2_attention_mask_func  # noqa: F401
