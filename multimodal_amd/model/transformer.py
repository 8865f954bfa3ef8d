"""ParallelTransformerLayer and its sub-modules with the reference's constructor / forward signatures,
sub-module attribute names and parameter names (state-dict compatible with
megatron/model/transformer.py:71-134, 179-641, 644-813), computing with hand-written gfx950 kernels:

  LayerNorm (one pass, dual output for the GPT-J variant)  ->  QKV GEMM (MFMA, bias epilogue)
  ->  rotary in place on the fused QKV activation  ->  fused causal / masked attention (no [s, s] tensor)
  ->  dense GEMM  ->  [adapter: two GEMMs with fused ReLU / residual]  ->  h->4h GEMM with the bias-GeLU epilogue
  ->  4h->h GEMM  ->  bias-dropout-add / GPT-J combine.

Tensor parallelism follows the reference exactly, including the GPT-J single all-reduce that counts the Row-linear
biases once per rank (transformer.py:690,703,747-766).
"""
import math

import os

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import mpu
from .fused_bias_dropout import (bias_dropout_add_fused_inference, bias_dropout_add_fused_train,
                                 get_bias_dropout_add)
from .norms import LayerNorm, get_norm
from .positional_embeddings import RotaryEmbedding
from .utils import exists

def _is_plain_causal(mask):
    """True if `mask` is the lower-triangular [1,1,s,s] mask get_attn_mask builds (megatron/utils.py:66-76) - the only
    mask the reference's training path ever passes.  Checked once per mask tensor (one device sync); the verdict is
    cached ON the tensor object together with its version counter and storage address."""
    if mask is None:
        return True
    tag = (mask.data_ptr(), mask._version)
    hit = getattr(mask, "_mm_causal", None)
    if hit is not None and hit[0] == tag:
        return hit[1]
    ok = mask.dim() == 4 and mask.size(0) == 1 and mask.size(1) == 1 and mask.size(2) == mask.size(3)
    if ok:
        s = mask.size(-1)
        ref = torch.triu(torch.ones(s, s, dtype=torch.bool, device=mask.device), 1)
        ok = bool(torch.equal(mask.view(s, s).to(torch.bool), ref))
    mask._mm_causal = (tag, ok)
    return ok


class ParallelMLP(nn.Module):
    """h -> 4h -> h MLP (megatron/model/transformer.py:71-134); returns (output, output_bias)."""

    def __init__(self, neox_args, init_method, output_layer_init_method, parallel_output=False):
        super().__init__()
        if neox_args.activation != "gelu":
            raise NotImplementedError(f"activation {neox_args.activation}: only gelu is on the accelerated path")
        self.activation_type = neox_args.activation
        self.bias_gelu_fusion = neox_args.bias_gelu_fusion
        ff_dim = 4 * neox_args.hidden_size
        self.dense_h_to_4h = mpu.ColumnParallelLinear(
            neox_args=neox_args, input_size=neox_args.hidden_size, output_size=ff_dim, gather_output=False,
            init_method=init_method, skip_bias_add=True)
        self.dense_4h_to_h = mpu.RowParallelLinear(
            neox_args=neox_args, input_size=ff_dim, output_size=neox_args.hidden_size, input_is_parallel=True,
            init_method=output_layer_init_method, skip_bias_add=True, parallel_output=parallel_output)

    def forward(self, hidden_states):
        # dense_h_to_4h (skip_bias_add) + bias-GeLU in ONE kernel: the activation is the GEMM's epilogue
        # and, in the backward, its derivative is the epilogue of the data-gradient GEMM through dense_4h_to_h.
        # TP mappings as in Column/RowParallelLinear (mpu/layers.py:549-565,735-757): copy in, all-reduce out.
        x = mpu.copy_to_model_parallel_region(hidden_states)
        row = self.dense_4h_to_h
        out = Fn.mlp(x, self.dense_h_to_4h.weight, self.dense_h_to_4h.bias, row.weight, 0 if self.bias_gelu_fusion else 1)
        if not row.parallel_output:
            out = mpu.reduce_from_model_parallel_region(out)
        return out, row.bias


class ParallelSelfAttention(nn.Module):
    """megatron/model/transformer.py:179-641 for pos_emb in {rotary, none}, global attention, no kv-cache."""

    def __init__(self, neox_args, attention_mask_func, init_method, output_layer_init_method, layer_number,
                 rpe=None, rotary=False, use_cache=False, parallel_output=False):
        super().__init__()
        self.fp16 = neox_args.precision == "fp16"
        self.bf16 = neox_args.precision == "bfloat16"
        self.attention_mask_func = attention_mask_func
        self.apply_query_key_layer_scaling = neox_args.apply_query_key_layer_scaling
        self.use_cache = use_cache
        self.layer_number = layer_number
        if rpe is not None or neox_args.pos_emb == "alibi":
            raise NotImplementedError("rpe / alibi position biases are outside this path's scope")
        if neox_args.attention_config[layer_number] != "global":
            raise NotImplementedError("only 'global' attention is on the accelerated path (no flash package, no sparse)")
        world_size = mpu.get_model_parallel_world_size()
        self.hidden_size_per_partition = mpu.divide(neox_args.hidden_size, world_size)
        self.hidden_size_per_attention_head = mpu.divide(neox_args.hidden_size, neox_args.num_attention_heads)
        self.num_attention_heads_per_partition = mpu.divide(neox_args.num_attention_heads, world_size)
        self.pos_emb = neox_args.pos_emb
        self.query_key_value = mpu.ColumnParallelLinear(
            neox_args=neox_args, input_size=neox_args.hidden_size, output_size=3 * neox_args.hidden_size,
            gather_output=False, init_method=init_method)
        self.norm_factor = math.sqrt(self.hidden_size_per_attention_head)
        if self.apply_query_key_layer_scaling:
            # scores / (sqrt(hn) * coeff), then softmax(coeff * .) in fp32 (transformer.py:229-233, fused_softmax.py:172):
            # the two factors cancel in exact arithmetic; the fused kernel computes scores in fp32 anyway.
            pass
        if rotary:
            if neox_args.rotary_pct == 1:
                self.rotary_ndims = None
            else:
                assert neox_args.rotary_pct < 1
                self.rotary_ndims = int(self.hidden_size_per_attention_head * neox_args.rotary_pct)
            dim = self.rotary_ndims if self.rotary_ndims is not None else self.hidden_size_per_attention_head
            self.rotary_emb = RotaryEmbedding(dim, base=neox_args.rotary_emb_base, precision=neox_args.params_dtype)
        else:
            self.rotary_emb = None
        self.upper_triang_fusion = bool(neox_args.scaled_upper_triang_masked_softmax_fusion)
        # torch.nn.Dropout(attention_dropout) on the softmax output (transformer.py:301-302, 383-384): inside the attention
        # kernels, mask regenerated from the model-parallel rng stream in the backward (include/mm_hip.h, mm_attn_fwd_dropout)
        self.dropout_p = float(neox_args.attention_dropout)
        if not 0.0 <= self.dropout_p < 1.0:
            raise ValueError("attention_dropout must be in [0, 1)")
        self.dense = mpu.RowParallelLinear(
            neox_args=neox_args, input_size=neox_args.hidden_size, output_size=neox_args.hidden_size,
            input_is_parallel=True, init_method=output_layer_init_method, skip_bias_add=True,
            parallel_output=parallel_output)

    def forward(self, hidden_states, attention_mask, layer_past=None):
        if self.use_cache or (exists(layer_past) and layer_past.numel() > 0):
            raise NotImplementedError("kv-cache inference is outside this path's scope (training hot path only)")
        s, b = hidden_states.size(0), hidden_states.size(1)
        np_, hn = self.num_attention_heads_per_partition, self.hidden_size_per_attention_head
        cos = sin = None
        d = 0
        if exists(self.rotary_emb):
            d = self.rotary_ndims if exists(self.rotary_ndims) else hn
            cos, sin = self.rotary_emb.tables(hidden_states, s)
        # the fused causal kernel ignores the mask tensor (fused_softmax.py:173-179); otherwise a mask that IS the
        # plain causal one takes the same fast path (identical result: exp(-10000 - max) == 0 in fp32)
        mask = None if (self.upper_triang_fusion or _is_plain_causal(attention_mask)) else attention_mask
        # query_key_value (ColumnParallelLinear, gather_output=False: mpu/layers.py:549-564) + rotary + attention core as one
        # node: the fused QKV activation [s, b, np*3*hn] (per-head interleaved q, k, v) is produced already rotated
        qkv_lin = self.query_key_value
        x = mpu.copy_to_model_parallel_region(hidden_states)
        dropout = None
        if self.dropout_p > 0.0 and self.training:
            with mpu.get_cuda_rng_tracker().fork():  # the reference's fork: a different stream per TP rank (its heads differ)
                state, pos = mpu.get_cuda_rng_tracker().draw_device(hidden_states.device)
            dropout = (self.dropout_p, state, pos)
        context = Fn.qkv_attention(x, qkv_lin.weight, qkv_lin.bias, cos, sin, mask, np_, hn, d,
                                   1.0 / self.norm_factor, dropout)  # [s, b, np*hn]
        return self.dense(context)


_SIDE_STREAMS = {}


def _side_stream(device):
    st = _SIDE_STREAMS.get(device)
    if st is None:
        st = _SIDE_STREAMS[device] = torch.cuda.Stream(device=device)
    return st


_HALF_CUS = {}


def _branch_shares(device):
    """Workgroup limits (attention branch, MLP branch) inside the two-stream region: half the CUs each.
    MM_CU_SHARE=<n> or <attn>,<mlp> overrides (0 = no limit; A/B)."""
    env = os.environ.get("MM_CU_SHARE")
    if env is not None:
        v = [int(x) for x in env.split(",")]
        return (v[0], v[-1])
    n = _HALF_CUS.get(device)
    if n is None:
        n = _HALF_CUS[device] = (torch.cuda.get_device_properties(device).multi_processor_count // 2) & ~7
    return (n, n)


def _branch_streams():
    """Attention branch and MLP GEMMs of a GPT-J layer on two HIP streams (MM_BRANCH_STREAMS=0 puts them back on one).
    Measured on one box: eager step 24.7 -> 23.9 ms (the second stream fills launch gaps), graph-replayed step 22.27 ->
    22.17 ms (the kernels overlap in time - rocprofv3 trace: 203 ms of kernel time in 144 ms - but share the CUs)."""
    return os.environ.get("MM_BRANCH_STREAMS", "1") != "0"


class ParallelTransformerLayer(nn.Module):
    """A single transformer layer, [s, b, h] -> [s, b, h] (megatron/model/transformer.py:644-801)."""

    def __init__(self, neox_args, attention_mask_func, init_method, output_layer_init_method, layer_number,
                 rpe=None, rotary=False, use_cache=False):
        super().__init__()
        self.layer_number = layer_number
        norm, eps = get_norm(neox_args)
        self.input_layernorm = norm(neox_args.hidden_size, eps=eps)
        self.use_cache = use_cache
        self.hidden_dropout = neox_args.hidden_dropout
        self.bias_dropout_fusion = neox_args.bias_dropout_fusion
        self.gpt_j_residual = neox_args.gpt_j_residual
        self.gpt_j_tied = neox_args.gpt_j_tied
        if self.gpt_j_residual:
            self.reduce = mpu.mappings.reduce_from_model_parallel_region
        self.attention = ParallelSelfAttention(
            neox_args=neox_args, attention_mask_func=attention_mask_func, init_method=init_method,
            output_layer_init_method=output_layer_init_method, layer_number=layer_number, rpe=rpe,
            use_cache=self.use_cache, rotary=rotary, parallel_output=self.gpt_j_residual)
        self.post_attention_layernorm = norm(neox_args.hidden_size, eps=eps)
        self.mlp = ParallelMLP(neox_args=neox_args, init_method=init_method,
                               output_layer_init_method=output_layer_init_method, parallel_output=self.gpt_j_residual)
        self.layer_past = None
        # module dtype follows the engine's cast (`layer.to(dtype)`), like the reference's LayerNorm / inv_freq
        self.to(neox_args.params_dtype)

    def _fused_adapter_tail(self):
        """(adapter Linear 0, Linear 2) when `.mlp` is an AdapterWrapper of a shape the fused adapter kernel takes
        (training.py:421-432 wraps `.mlp` / `.attention`), else None."""
        from .adapter import AdapterWrapper
        from .. import ops
        m = self.mlp
        if type(m) is not AdapterWrapper:
            return None
        a0, a2 = m.adapter[0], m.adapter[2]
        if not ops.adapter_supported(a0.weight.size(1), a0.weight.size(0)):
            return None
        return a0, a2

    def _get_bias_dropout(self):
        if self.bias_dropout_fusion:
            return bias_dropout_add_fused_train if self.training else bias_dropout_add_fused_inference
        return get_bias_dropout_add(self.training)

    def forward(self, x, attention_mask, layer_past=None):
        layer_past = layer_past if layer_past is not None else self.layer_past
        bias_dropout_fn = self._get_bias_dropout()
        if self.gpt_j_residual:
            residual = x
            ln1, ln2 = self.input_layernorm, self.post_attention_layernorm
            if self.gpt_j_tied:
                x1 = x2 = ln1(x)
            elif isinstance(ln1, LayerNorm) and isinstance(ln2, LayerNorm) and ln1.eps == ln2.eps:
                # the residual is routed through the same node: its gradient is added inside the LayerNorm backward kernel
                x1, x2, residual = Fn.dual_layer_norm_res(x, ln1.weight, ln1.bias, ln2.weight, ln2.bias, ln1.eps)
            else:
                x1, x2 = ln1(x), ln2(x)
            no_drop = self.hidden_dropout == 0 or not self.training
            tail = self._fused_adapter_tail() if no_drop else None
            if tail is None:
                attention_output, attention_bias = self.attention(x1, attention_mask, layer_past=layer_past)
            if tail is not None:
                # MAGMA layer: the MLP's adapter is the last kernel, so the residual tail rides on its output store
                a0, a2 = tail
                if _branch_streams() and x2.is_cuda:
                    # GPT-J parallel residual: the MLP GEMMs and the attention branch only meet in the tail, so they go to two
                    # HIP streams (the attention kernels are vector-bound, the GEMMs matrix-bound: they share a CU well)
                    main, side = torch.cuda.current_stream(), _side_stream(x2.device)
                    side.wait_stream(main)
                    # each branch's big GEMMs take half the CUs, so that the pair runs side by side instead of one persistent
                    # launch after the other (ops.cu_share; step 20.34 -> 19.40 ms on one box)
                    from .. import ops
                    share_attn, share_mlp = _branch_shares(x2.device)
                    with ops.cu_share(share_mlp), torch.cuda.stream(side):
                        mlp_raw, mlp_bias = self.mlp.attn_block(x2)
                    with ops.cu_share(share_attn):
                        attention_output, attention_bias = self.attention(x1, attention_mask, layer_past=layer_past)
                    main.wait_stream(side)
                    x2.record_stream(side)       # allocated on one stream, read on the other: the allocator must not
                    mlp_raw.record_stream(main)  # hand the block out again before that stream has passed the reader
                else:
                    attention_output, attention_bias = self.attention(x1, attention_mask, layer_past=layer_past)
                    mlp_raw, mlp_bias = self.mlp.attn_block(x2)
                tp1 = mpu.get_model_parallel_world_size() == 1
                output = Fn.adapter_combine(mlp_raw, a0.weight, a0.bias, a2.weight, a2.bias, residual if tp1 else None,
                                            attention_output, attention_bias, mlp_bias)
                if tp1:
                    return output
                return Fn.bias_dropout_add(self.reduce(output), None, residual, 0.0, False)
            mlp_output, mlp_bias = self.mlp(x2)
            if no_drop:
                # bias_dropout_add x2 (+ the residual add when TP == 1) in one pass
                if mpu.get_model_parallel_world_size() == 1:
                    return Fn.gptj_combine(residual, attention_output, attention_bias, mlp_output, mlp_bias)
                output = Fn.gptj_combine(None, attention_output, attention_bias, mlp_output, mlp_bias)
            else:
                attention_output = bias_dropout_fn(attention_output, bias=attention_bias.expand_as(attention_output),
                                                   residual=None, prob=self.hidden_dropout)
                output = bias_dropout_fn(mlp_output, bias=mlp_bias.expand_as(mlp_output), residual=attention_output,
                                         prob=self.hidden_dropout)
            # ONE all-reduce of (attn_partial + b_attn + mlp_partial + b_mlp): transformer.py:766
            return Fn.bias_dropout_add(self.reduce(output), None, residual, 0.0, False)
        residual = x
        attention_output, attention_bias = self.attention(self.input_layernorm(x), attention_mask, layer_past=layer_past)
        attention_output = bias_dropout_fn(attention_output, bias=attention_bias.expand_as(residual), residual=residual,
                                           prob=self.hidden_dropout)
        mlp_output, mlp_bias = self.mlp(self.post_attention_layernorm(attention_output))
        return bias_dropout_fn(mlp_output, bias=mlp_bias.expand_as(attention_output), residual=attention_output,
                               prob=self.hidden_dropout)


class ParallelTransformerLayerPipe(ParallelTransformerLayer):
    """Forwards the attention mask through a pipeline of layers (megatron/model/transformer.py:804-813)."""

    def forward(self, args):
        assert len(args) == 2, "ParallelTransformerLayerPipe expects 2 arguments - hidden_states and attention_mask"
        hidden_states, attention_mask = args
        return super().forward(hidden_states, attention_mask), attention_mask
