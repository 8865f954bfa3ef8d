"""CPU ORACLE - test infrastructure, NOT the product.

A plain-PyTorch (CPU, eager) restatement of the reference's transformer-layer hot path
(AGI-Collective/multimodal, a GPT-NeoX 2.0 fork).  Every function cites the reference file:line it
follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module - the product path (``multimodal_amd``) never does and fails loudly without its
HIP library.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against fixtures in
``tests/golden/`` that were produced by running the reference's own modules on CPU
(``tests/golden/make_golden.py``, shims per SURVEY.md 8c): bit-exact for the eager ops in
fp32/bf16/fp16, and to float round-off for the whole layer.  The two fused-softmax restatements
(``softmax_causal_fused`` / ``softmax_masked_fused``) follow the CUDA kernels, which cannot run
without a GPU; they are pinned against the reference's eager fallback at the reference's own
tolerance (mean |diff| <= 1e-3, tests/model/test_fused_kernels.py:122-142).
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------- norms / activations
def layer_norm(x, w, b, eps=1e-5):
    """torch.nn.LayerNorm via get_norm (megatron/model/norms.py:16-31)."""
    return F.layer_norm(x, (x.size(-1),), w, b, eps)


def bias_gelu_tanh(bias, y):
    """bias_gelu (megatron/model/activations.py:62-65)."""
    x = bias + y
    return x * 0.5 * (1.0 + torch.tanh(0.79788456 * x * (1 + 0.044715 * x * x)))


def bias_gelu_tanh_back(g, bias, y):
    """bias_gelu_back (megatron/model/activations.py:71-79)."""
    x = bias + y
    tanh_out = torch.tanh(0.79788456 * x * (1 + 0.044715 * x * x))
    ff = 0.5 * x * ((1 - tanh_out * tanh_out) * (0.79788456 + 0.1070322243 * x * x)) + 0.5 * (1 + tanh_out)
    return ff * g


class _GeLUFunction(torch.autograd.Function):
    """GeLUFunction (megatron/model/activations.py:82-93): same grad for input and bias."""

    @staticmethod
    def forward(ctx, inp, bias):
        ctx.save_for_backward(inp, bias)
        return bias_gelu_tanh(bias, inp)

    @staticmethod
    def backward(ctx, g):
        inp, bias = ctx.saved_tensors
        tmp = bias_gelu_tanh_back(g, bias, inp)
        return tmp, tmp


def bias_gelu_impl(y, bias):
    return _GeLUFunction.apply(y, bias)


def gelu_erf_bias(y, bias):
    """F.gelu(y + bias) (megatron/model/transformer.py:127-130)."""
    return F.gelu(y + bias)


def bias_dropout_add(x, bias, residual, prob, training):
    """bias_dropout_add (megatron/model/fused_bias_dropout.py:28-34)."""
    out = F.dropout(x + bias, p=prob, training=training)
    if residual is not None:
        out = residual + out
    return out


# --------------------------------------------------------------------------------- rotary
def rotary_tables(seq_len, dim, base=10000, dtype=torch.float32):
    """RotaryEmbedding.forward (megatron/model/positional_embeddings.py:38-63) AFTER the engine has cast
    the module to `dtype` (so inv_freq and the positions are in `dtype`: bf16 positions > 256 round)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    inv_freq = inv_freq.to(dtype)
    t = torch.arange(seq_len).type_as(inv_freq)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    if dtype == torch.bfloat16:
        emb = emb.float()
    cos, sin = emb.cos(), emb.sin()
    if dtype == torch.bfloat16:
        cos, sin = cos.bfloat16(), sin.bfloat16()
    return cos, sin  # [seq_len, dim]


def rotate_half(x):
    """positional_embeddings.py:69-73."""
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=x1.ndim - 1)


def apply_rotary(q, k, cos, sin, offset=0):
    """apply_rotary_pos_emb{,_torch} (positional_embeddings.py:76-92); cos/sin: [S, d] tables."""
    cos = cos[offset: q.shape[0] + offset][:, None, None, :]
    sin = sin[offset: q.shape[0] + offset][:, None, None, :]
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


# --------------------------------------------------------------------------------- softmax
def attention_mask_func(scores, ltor_mask):
    """gpt2_attention_mask_func (megatron/model/gpt2_model.py:47-49) - in place, -10000 fill."""
    scores.masked_fill_(ltor_mask, -10000.0)
    return scores


def causal_mask(s, device="cpu"):
    """get_attn_mask (megatron/utils.py:66-76): [1,1,s,s] bool, True = masked."""
    return torch.tril(torch.ones((1, s, s), device=device)).view(1, 1, s, s) < 0.5


def softmax_eager(x, mask, scale=None, softmax_in_fp32=False):
    """FusedScaleMaskSoftmax.forward_torch_softmax (megatron/model/fused_softmax.py:184-199)."""
    in16 = x.dtype in (torch.float16, torch.bfloat16)
    dt = x.dtype
    if in16 and softmax_in_fp32:
        x = x.float()
    if scale is not None:
        x = x * scale
    mo = attention_mask_func(x, mask) if mask is not None else x
    probs = torch.nn.Softmax(dim=-1)(mo)
    if in16 and softmax_in_fp32:
        probs = probs.to(dt)
    return probs


def softmax_causal_fused_fwd(x, scale):
    """scaled_upper_triang_masked_softmax_warp_forward
    (megatron/fused_kernels/scaled_upper_triang_masked_softmax.h:143-260): x [attn_batches, s, s];
    row r: fp32 softmax(scale * x[r, :r+1]); zeros above the diagonal; output in x.dtype."""
    s = x.size(-1)
    xf = x.float() * scale
    dead = torch.triu(torch.ones(s, s, dtype=torch.bool), diagonal=1)
    xf = xf.masked_fill(dead, float("-inf"))
    e = torch.exp(xf - xf.max(dim=-1, keepdim=True).values)
    p = e / e.sum(dim=-1, keepdim=True)
    return p.masked_fill(dead, 0.0).to(x.dtype)


def softmax_fused_bwd(grad, y, scale, causal):
    """scaled_*_softmax_warp_backward (scaled_upper_triang_masked_softmax.h:262-361,
    scaled_masked_softmax.h:240-335): scale * (g*y - y * sum(g*y)) in fp32, cast to the input dtype."""
    gf, yf = grad.float(), y.float()
    if causal:
        s = y.size(-1)
        live = torch.tril(torch.ones(s, s, dtype=torch.bool))
        yf = yf * live
    gy = gf * yf
    return (scale * (gy - yf * gy.sum(dim=-1, keepdim=True))).to(grad.dtype)


def softmax_masked_fused_fwd(x, mask, scale):
    """scaled_masked_softmax_warp_forward (megatron/fused_kernels/scaled_masked_softmax.h:117-238):
    x [b, np, sq, sk]; mask uint8/bool [1|b, 1, sq, sk]; masked -> -10000 (not scaled); fp32 math."""
    xf = x.float() * scale
    xf = torch.where(mask.to(torch.uint8) == 1, torch.full_like(xf, -10000.0), xf)
    e = torch.exp(xf - xf.max(dim=-1, keepdim=True).values)
    return (e / e.sum(dim=-1, keepdim=True)).to(x.dtype)


def softmax_batch_per_block(sq, sk, b, np_, warp_size=64):
    """get_batch_per_block (scaled_masked_softmax.h:338-352) with C10_WARP_SIZE = 64 (ROCm)."""
    pow2 = 1 << max(0, math.ceil(math.log2(sk)))
    warp = min(pow2, warp_size)
    per_warp = 2 if pow2 <= 128 else 1
    return (128 // warp) * per_warp


def fused_softmax_available(fusion, in16, mask_present, b, np_, sq, sk, upper_triang):
    """FusedScaleMaskSoftmax.is_kernel_available (megatron/model/fused_softmax.py:148-168)."""
    attn_batches = b * np_
    if fusion and in16 and mask_present and 16 < sk <= 2048 and sq % 4 == 0 and attn_batches % 4 == 0:
        bpb = softmax_batch_per_block(sq, sk, b, np_)
        if upper_triang:
            return attn_batches % bpb == 0
        return sq % bpb == 0
    return False


# --------------------------------------------------------------------------------- the layer
class LayerConfig:
    """The neox_args keys the path reads (key list)."""

    def __init__(self, hidden_size, num_attention_heads, rotary_pct=1.0, rotary_emb_base=10000,
                 gpt_j_residual=False, gpt_j_tied=False, bias_gelu_fusion=False, layernorm_epsilon=1e-5,
                 hidden_dropout=0.0, attention_dropout=0.0, softmax_fusion="none", tp=1):
        self.hidden_size = hidden_size
        self.num_attention_heads = num_attention_heads
        self.rotary_pct = rotary_pct
        self.rotary_emb_base = rotary_emb_base
        self.gpt_j_residual = gpt_j_residual
        self.gpt_j_tied = gpt_j_tied
        self.bias_gelu_fusion = bias_gelu_fusion
        self.layernorm_epsilon = layernorm_epsilon
        self.hidden_dropout = hidden_dropout
        self.attention_dropout = attention_dropout
        self.softmax_fusion = softmax_fusion  # "none" | "upper_triang" | "general"
        self.tp = tp

    @property
    def hn(self):
        return self.hidden_size // self.num_attention_heads

    @property
    def rotary_ndims(self):
        """transformer.py:248-260."""
        return self.hn if self.rotary_pct == 1 else int(self.hn * self.rotary_pct)


def self_attention(cfg, p, x, mask, prefix="attention."):
    """ParallelSelfAttention.forward for ONE tensor-parallel rank (megatron/model/transformer.py:543-641,
    attention(): 315-416).  `p` holds this rank's shards; returns (partial output, dense bias)."""
    s, b, _ = x.shape
    np_ = cfg.num_attention_heads // cfg.tp
    hn = cfg.hn
    mixed = F.linear(x, p[prefix + "query_key_value.weight"], p[prefix + "query_key_value.bias"])  # layers.py:557
    mixed = mixed.view(s, b, np_, 3 * hn)
    q, k, v = torch.split(mixed, hn, dim=-1)  # mpu/utils.py:36-53
    d = cfg.rotary_ndims
    cos, sin = rotary_tables(s, d, cfg.rotary_emb_base, x.dtype)
    if d < hn:
        qr, kr = apply_rotary(q[..., :d], k[..., :d], cos, sin)
        q = torch.cat((qr, q[..., d:]), dim=-1)
        k = torch.cat((kr, k[..., d:]), dim=-1)
    else:
        q, k = apply_rotary(q, k, cos, sin)
    norm_factor = math.sqrt(hn)
    # scores (transformer.py:330-355): baddbmm(beta=0, alpha=1/norm_factor)
    qv = q.reshape(s, b * np_, hn)
    kv = k.reshape(s, b * np_, hn)
    scores = torch.baddbmm(torch.empty(b * np_, s, s, dtype=q.dtype), qv.transpose(0, 1),
                           kv.transpose(0, 1).transpose(1, 2), beta=0.0, alpha=1.0 / norm_factor)
    scores = scores.view(b, np_, s, s)
    in16 = x.dtype in (torch.float16, torch.bfloat16)
    fused = fused_softmax_available(cfg.softmax_fusion != "none", in16, mask is not None, b, np_, s, s,
                                    cfg.softmax_fusion == "upper_triang")
    if fused and cfg.softmax_fusion == "upper_triang":
        probs = _FusedCausalSoftmax.apply(scores.view(-1, s, s), 1.0).view(b, np_, s, s)  # fused_softmax.py:173-179
    elif fused:
        probs = _FusedMaskedSoftmax.apply(scores, mask, 1.0)
    else:
        probs = softmax_eager(scores, mask, None, False)  # fused_softmax.py:184-199
    probs = F.dropout(probs, p=cfg.attention_dropout, training=cfg.attention_dropout > 0)
    vv = v.reshape(s, b * np_, hn)
    ctx = torch.bmm(probs.view(b * np_, s, s), vv.transpose(0, 1))  # transformer.py:412
    ctx = ctx.view(b, np_, s, hn).permute(2, 0, 1, 3).contiguous().view(s, b, np_ * hn)  # :624-630
    out = F.linear(ctx, p[prefix + "dense.weight"])  # RowParallelLinear, skip_bias_add (layers.py:745-757)
    return out, p[prefix + "dense.bias"]


class _FusedCausalSoftmax(torch.autograd.Function):
    """ScaledUpperTriangMaskedSoftmax (megatron/model/fused_softmax.py:24-53)."""

    @staticmethod
    def forward(ctx, x, scale):
        y = softmax_causal_fused_fwd(x, scale)
        ctx.save_for_backward(y)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return softmax_fused_bwd(g, y, ctx.scale, causal=True), None


class _FusedMaskedSoftmax(torch.autograd.Function):
    """ScaledMaskedSoftmax (megatron/model/fused_softmax.py:56-83)."""

    @staticmethod
    def forward(ctx, x, mask, scale):
        y = softmax_masked_fused_fwd(x, mask, scale)
        ctx.save_for_backward(y)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return softmax_fused_bwd(g, y, ctx.scale, causal=False), None, None


def mlp(cfg, p, x, prefix="mlp."):
    """ParallelMLP.forward for one TP rank (megatron/model/transformer.py:116-134)."""
    inter = F.linear(x, p[prefix + "dense_h_to_4h.weight"])  # skip_bias_add
    b1 = p[prefix + "dense_h_to_4h.bias"]
    if cfg.bias_gelu_fusion:
        inter = bias_gelu_impl(inter, b1)
    else:
        inter = F.gelu(inter + b1)
    out = F.linear(inter, p[prefix + "dense_4h_to_h.weight"])
    return out, p[prefix + "dense_4h_to_h.bias"]


def adapter(p, prefix, o):
    """AdapterWrapper.forward's adapter(o) + o (megatron/model/adapter.py:109-117, 18-25)."""
    hdn = F.relu(F.linear(o, p[prefix + "adapter.0.weight"], p[prefix + "adapter.0.bias"]))
    return F.linear(hdn, p[prefix + "adapter.2.weight"], p[prefix + "adapter.2.bias"]) + o


def layer_forward(cfg, shards, x, mask, training=True):
    """ParallelTransformerLayer.forward (megatron/model/transformer.py:719-801) with the TP group
    simulated in-process: `shards` is a list (one dict per TP rank) of that rank's parameters; the
    all-reduce (mpu/mappings.py:29-48) is a sum over the list.  Reproduces the reference's behaviour
    that with gpt_j_residual both Row-linear biases are added on every rank BEFORE the single reduce
    (transformer.py:747-766), i.e. counted tp times.  Adapter weights are used when present
    (keys attention.adapter.* / mlp.adapter.*): with gpt_j_residual the wrapped Row linears return their PARTIAL sums
    (parallel_output=True, transformer.py:690,703), so each rank's adapter - ReLU included - acts on that rank's partial sum
    before the single reduce (adapter.py:109-117)."""
    tp = len(shards)
    assert tp == cfg.tp
    eps = cfg.layernorm_epsilon
    pd = cfg.hidden_dropout

    def attn_of(p, h_in):
        o, bias = self_attention(cfg, p, h_in, mask)
        if "attention.adapter.0.weight" in p:
            o = adapter(p, "attention.", o)
        return o, bias

    def mlp_of(p, h_in):
        o, bias = mlp(cfg, p, h_in)
        if "mlp.adapter.0.weight" in p:
            o = adapter(p, "mlp.", o)
        return o, bias

    if cfg.gpt_j_residual:
        partial = []
        for p in shards:
            if cfg.gpt_j_tied:
                x1 = x2 = layer_norm(x, p["input_layernorm.weight"], p["input_layernorm.bias"], eps)
            else:
                x1 = layer_norm(x, p["input_layernorm.weight"], p["input_layernorm.bias"], eps)
                x2 = layer_norm(x, p["post_attention_layernorm.weight"], p["post_attention_layernorm.bias"], eps)
            ao, ab = attn_of(p, x1)
            ao = bias_dropout_add(ao, ab.expand_as(ao), None, pd, training)
            mo, mb = mlp_of(p, x2)
            partial.append(bias_dropout_add(mo, mb.expand_as(mo), ao, pd, training))
        red = partial[0] if tp == 1 else torch.stack(partial).sum(0)
        return x + red
    # sequential residual (transformer.py:767-799): each Row linear all-reduces INSIDE the wrapped module
    # (parallel_output=False, layers.py:747-748), so an AdapterWrapper sees the reduced output and every rank computes the
    # same adapter result from its own (identical) copy of the adapter weights - rank 0's copy stands for all of them here;
    # the Row bias is added once, after
    p0 = shards[0]
    ln1 = [layer_norm(x, p["input_layernorm.weight"], p["input_layernorm.bias"], eps) for p in shards]
    outs = [self_attention(cfg, p, h_in, mask) for p, h_in in zip(shards, ln1)]
    ao = outs[0][0] if tp == 1 else torch.stack([o for o, _ in outs]).sum(0)
    if "attention.adapter.0.weight" in p0:
        ao = adapter(p0, "attention.", ao)
    a = bias_dropout_add(ao, outs[0][1].expand_as(x), x, pd, training)
    ln2 = [layer_norm(a, p["post_attention_layernorm.weight"], p["post_attention_layernorm.bias"], eps) for p in shards]
    outs = [mlp(cfg, p, h_in) for p, h_in in zip(shards, ln2)]
    mo = outs[0][0] if tp == 1 else torch.stack([o for o, _ in outs]).sum(0)
    if "mlp.adapter.0.weight" in p0:
        mo = adapter(p0, "mlp.", mo)
    return bias_dropout_add(mo, outs[0][1].expand_as(a), a, pd, training)


def column_parallel_linear(x, w, b, skip_bias_add=False):
    """ColumnParallelLinear.forward, gather_output=False (megatron/mpu/layers.py:549-564)."""
    y = F.linear(x, w, None if skip_bias_add else b)
    return y, (b if skip_bias_add else None)


def row_parallel_linear(x_shards, w_shards, b, skip_bias_add=False, parallel_output=False):
    """RowParallelLinear.forward with input_is_parallel (megatron/mpu/layers.py:736-757); the reduce is a sum."""
    parts = [F.linear(xs, ws) for xs, ws in zip(x_shards, w_shards)]
    if parallel_output:
        outs = parts
    else:
        tot = parts[0] if len(parts) == 1 else torch.stack(parts).sum(0)
        outs = [tot] * len(parts)
    if not skip_bias_add:
        outs = [o + b if b is not None else o for o in outs]
        return outs, None
    return outs, b


def cross_entropy_loss(logits, labels, loss_mask):
    """cross_entropy (megatron/model/gpt2_model.py:52-70) at TP=1: fp32 logits, masked mean."""
    losses = F.cross_entropy(logits.float().view(-1, logits.size(-1)), labels.reshape(-1), reduction="none")
    lm = loss_mask.reshape(-1)
    return torch.sum(losses * lm) / lm.sum()


def vocab_parallel_cross_entropy(logit_shards, target):
    """_VocabParallelCrossEntropy (megatron/mpu/cross_entropy.py:27-112) with the TP group simulated in-process:
    `logit_shards` is a list (one per TP rank) of fp32 [..., V/p] partitions; the three all-reduces (MAX :33, SUM :68,
    SUM :78) are reductions over the list.  Returns the per-token loss (target's shape); autograd through it gives the
    reference's hand-written backward, (softmax - onehot) * grad_output (:93-112) - the shift by the (detached) maximum
    has zero derivative."""
    vp = logit_shards[0].size(-1)
    mx = torch.stack([x.detach().max(dim=-1)[0] for x in logit_shards]).max(dim=0)[0]
    predicted = 0.0
    sum_exp = 0.0
    for r, x in enumerate(logit_shards):
        sh = x - mx.unsqueeze(-1)
        lo, hi = r * vp, (r + 1) * vp                       # VocabUtility.vocab_range_from_per_partition_vocab_size
        outside = (target < lo) | (target >= hi)
        local = (target - lo).masked_fill(outside, 0)
        pr = torch.gather(sh, -1, local.unsqueeze(-1)).squeeze(-1)
        predicted = predicted + pr.masked_fill(outside, 0.0)
        sum_exp = sum_exp + torch.exp(sh).sum(dim=-1)
    return torch.log(sum_exp) - predicted


# --------------------------------------------------------------------------------- image prefix + embedding (row f3)
def image_prefix_tail(enc_out, proj_w, proj_b, ln_w, ln_b, eps=1e-5, dropout_p=0.0, training=False):
    """ImagePrefix.forward behind its encoder (megatron/model/image_prefix.py:224-252): `enc_out` is the encoder's output,
    [b, enc] (pooled: openclip-H as coded, :106) or [b, P, enc]; proj (:232) -> dropout (:243) -> LayerNorm (:245-246) ->
    unsqueeze(1) for the 2-D case (:249-250).  Returns [b, P, h]."""
    y = F.linear(enc_out, proj_w, proj_b)
    y = F.dropout(y, p=dropout_p, training=training)
    y = F.layer_norm(y, (y.size(-1),), ln_w, ln_b, eps)
    return y.unsqueeze(1) if y.dim() == 2 else y


def embed_with_prefix(prefix, table, tokens):
    """EmbeddingPipe.forward (megatron/model/word_embeddings.py:182-205: word embeddings, rotary => no position embedding
    :138-149, embedding dropout 0) + cat((image_embeddings, word_embeddings[:, :-P]), dim=1) (:197-203), then
    _pre_transformer_block's [b, s, h] -> [s, b, h] (megatron/model/gpt2_model.py:73-77).  prefix: [b, P, h] or None."""
    emb = F.embedding(tokens, table)
    if prefix is not None:
        emb = torch.cat((prefix, emb[:, : -prefix.shape[1], :]), dim=1)
    return emb.transpose(0, 1).contiguous()

