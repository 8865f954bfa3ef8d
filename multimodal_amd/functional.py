"""Autograd wiring of the HIP kernels: each Function's forward AND backward are C-ABI kernel launches
(``multimodal_amd.ops``); torch.autograd only sequences them.

Memory policy (MI355X, 288 GB HBM3E): the transposed copy W^T that the data-gradient GEMM wants as its
K-contiguous operand is cached per FROZEN weight (refreshed when the weight's version counter moves), so a frozen
language model (reference `freeze_lm`, megatron/training.py:454-459) pays for the transposes once.
"""
import contextlib

import torch

from . import _lib, _streams, ops


def weight_t(w):
    """Contiguous W^T ([in, out]) of a [out, in] weight.  Cached ON the weight object (so it dies with it) for FROZEN
    weights only, and refreshed when the storage or the version counter changes (load_state_dict, .to(device)).
    Trainable weights are transposed on every call: fused optimizers (torch.optim.Adam(fused=True)) update parameters
    without bumping the version counter, so a cache keyed on it would go stale after the first step."""
    if w.requires_grad:
        return ops.transpose2d(w.detach())
    ent = getattr(w, "_mm_wt", None)
    if ent is not None and ent[0] == w.data_ptr() and ent[1] == w._version and ent[2].device == w.device:
        _streams.wait(ent[3], w.device)  # filled on another stream (branch / sub-batch streams): order behind that kernel
        return ent[2]
    wt = ops.transpose2d(w.detach())
    w._mm_wt = (w.data_ptr(), w._version, wt, _streams.stamp(w.device) if w.is_cuda else None)
    return wt


def _flat2d(t):
    return t.reshape(-1, t.size(-1))


def _wgrad(dy2d, x2d):
    """dW[N, K] = dY^T[N, M] . X[M, K]  (contraction over the token dimension)."""
    return ops.gemm_tn(dy2d, x2d)


def _wgrad_bias(dy2d, x2d, need_w, need_b):
    """(dW, db) of y = x W^T + b: one launch pair gives both when both are wanted (db = column sums of dY, taken from
    the tiles the weight-gradient GEMM stages anyway)."""
    if need_w and need_b:
        return ops.gemm_tn(dy2d, x2d, with_colsum=True)
    if need_w:
        return ops.gemm_tn(dy2d, x2d), None
    return None, (ops.colsum(dy2d) if need_b else None)


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ bias) (+ residual): F.linear of the reference's Column/Row linears (mpu/layers.py:557,745)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        return ops.gemm_nt(x, weight, bias, _lib.EPI_BIAS if bias is not None else _lib.EPI_NONE, residual=residual)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dy2 = _flat2d(dy)
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dy2, weight_t(weight)).view(x.shape)
        dw, db = _wgrad_bias(dy2, _flat2d(x), ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dw, db, dres


def linear(x, weight, bias=None, residual=None):
    return LinearFn.apply(x, weight, bias, residual)


class LinearBiasGeluFn(torch.autograd.Function):
    """act = gelu(x W^T + bias) with the activation fused into the GEMM epilogue.
    mode 0: tanh approximation = bias_gelu_impl (megatron/model/activations.py:62-96);
    mode 1: exact erf = F.gelu(y + bias) (megatron/model/transformer.py:127-130).
    The rounded pre-bias product is kept for the backward, as the reference's GeLUFunction saves it."""

    @staticmethod
    def forward(ctx, x, weight, bias, mode):
        act, pre = ops.gemm_nt(x, weight, bias, _lib.EPI_BIAS_GELU_TANH if mode == 0 else _lib.EPI_BIAS_GELU_ERF)
        ctx.save_for_backward(x, weight, bias, pre)
        ctx.mode = mode
        return act

    @staticmethod
    def backward(ctx, g):
        x, weight, bias, pre = ctx.saved_tensors
        dpre = ops.bias_gelu_bwd(g.contiguous(), pre, bias, ctx.mode)
        d2 = _flat2d(dpre)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(d2, weight_t(weight)).view(x.shape)
        dw, db = _wgrad_bias(d2, _flat2d(x), ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dw, db, None


def linear_bias_gelu(x, weight, bias, mode):
    return LinearBiasGeluFn.apply(x, weight, bias, mode)


class MlpFn(torch.autograd.Function):
    """out = gelu(x W1^T + b1) W2^T - both GEMMs of ParallelMLP (megatron/model/transformer.py:118-134; the output bias is
    returned separately by the module, skip_bias_add).  Forward: bias-GeLU is the first GEMM's epilogue.  Backward: the
    GeLU derivative is the epilogue of the data-gradient GEMM through W2, so neither direction has an elementwise pass."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, mode):
        act, pre = ops.gemm_nt(x, w1, b1, _lib.EPI_BIAS_GELU_TANH if mode == 0 else _lib.EPI_BIAS_GELU_ERF)
        out = ops.gemm_nt(act, w2)
        ctx.save_for_backward(x, w1, b1, w2, pre, act)
        ctx.mode = mode
        ctx.cu_share = ops.current_cu_share() if x.is_cuda else 0  # the layer's two-stream region (model/transformer.py)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2, pre, act = ctx.saved_tensors
        dy2 = _flat2d(dy.contiguous())
        with ops.cu_share(ctx.cu_share) if ctx.cu_share else contextlib.nullcontext():
            dpre = ops.gemm_nt(dy2, weight_t(w2), b1, _lib.EPI_GELU_TANH_BWD if ctx.mode == 0 else _lib.EPI_GELU_ERF_BWD,
                               residual=_flat2d(pre))
        dx = dw1 = db1 = dw2 = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dpre, weight_t(w1)).view(x.shape)
        dw1, db1 = _wgrad_bias(dpre, _flat2d(x), ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        if ctx.needs_input_grad[3]:
            dw2 = _wgrad(dy2, _flat2d(act))
        return dx, dw1, db1, dw2, None


def mlp(x, w1, b1, w2, mode):
    return MlpFn.apply(x, w1, b1, w2, mode)


class AdapterFn(torch.autograd.Function):
    """o' = o + W2 relu(W1 o + b1) + b2 - the bottleneck adapter of AdapterWrapper
    (megatron/model/adapter.py:18-25,109-117) as two GEMMs with fused bias/ReLU/residual epilogues."""

    @staticmethod
    def forward(ctx, o, w1, b1, w2, b2):
        if ops.adapter_supported(w1.size(1), w1.size(0)):
            out, h1 = ops.adapter_fwd(o, w1, b1, w2, b2)           # one kernel (csrc/adapter.hip)
        else:
            h1 = ops.gemm_nt(o, w1, b1, _lib.EPI_BIAS_RELU)
            out = ops.gemm_nt(h1, w2, b2, _lib.EPI_BIAS, residual=o)
        ctx.save_for_backward(o, w1, w2, h1)
        return out

    @staticmethod
    def backward(ctx, g):
        o, w1, w2, h1 = ctx.saved_tensors
        g = g.contiguous()
        g2 = _flat2d(g)
        do = None
        if ops.adapter_supported(w1.size(1), w1.size(0)):
            do, dpre = ops.adapter_bwd_dx(g2, w2.detach(), w1.detach(), _flat2d(h1))  # weights as stored: no W^T copies
            do = do.view(o.shape)
        else:
            # [M, r] data gradient through W2 and the ReLU in one kernel (the ReLU mask is the GEMM's epilogue)
            dpre = ops.gemm_nt(g2, weight_t(w2), None, _lib.EPI_RELU_BWD, residual=_flat2d(h1))
            if ctx.needs_input_grad[0]:
                do = ops.gemm_nt(dpre, weight_t(w1), residual=g2).view(o.shape)  # + identity branch
        if all(ctx.needs_input_grad[1:5]):  # both weight gradients and both bias gradients: one launch pair
            dw1, db1, dw2, db2 = ops.gemm_tn_pair(dpre, _flat2d(o), g2, _flat2d(h1))
        else:
            dw1, db1 = _wgrad_bias(dpre, _flat2d(o), ctx.needs_input_grad[1], ctx.needs_input_grad[2])
            dw2, db2 = _wgrad_bias(g2, _flat2d(h1), ctx.needs_input_grad[3], ctx.needs_input_grad[4])
        return do, dw1, db1, dw2, db2


def adapter(o, w1, b1, w2, b2):
    return AdapterFn.apply(o, w1, b1, w2, b2)


class AdapterCombineFn(torch.autograd.Function):
    """out = [x +] ((attn + b_a) + (adapter(o) + b_m)): the MLP's adapter (the last kernel of a MAGMA layer) with the GPT-J
    residual tail (GptjCombineFn) on its output store.  Backward = AdapterFn.backward on g, and g itself for x and attn."""

    @staticmethod
    def forward(ctx, o, w1, b1, w2, b2, x, attn, ba, bm):
        out, h1 = ops.adapter_fwd_combine(o, w1, b1, w2, b2, x, attn, ba, bm)
        ctx.save_for_backward(o, w1, w2, h1)
        ctx.has_x = x is not None
        return out

    @staticmethod
    def backward(ctx, g):
        o, w1, w2, h1 = ctx.saved_tensors
        g = g.contiguous()
        g2 = _flat2d(g)
        do, dpre = ops.adapter_bwd_dx(g2, w2.detach(), w1.detach(), _flat2d(h1))
        do = do.view(o.shape)
        if all(ctx.needs_input_grad[1:5]):
            dw1, db1, dw2, db2 = ops.gemm_tn_pair(dpre, _flat2d(o), g2, _flat2d(h1))
        else:
            dw1, db1 = _wgrad_bias(dpre, _flat2d(o), ctx.needs_input_grad[1], ctx.needs_input_grad[2])
            dw2, db2 = _wgrad_bias(g2, _flat2d(h1), ctx.needs_input_grad[3], ctx.needs_input_grad[4])
        db = ops.colsum(g2) if (ctx.needs_input_grad[7] or ctx.needs_input_grad[8]) else None
        return (do, dw1, db1, dw2, db2, (g if ctx.has_x else None), g, (db if ctx.needs_input_grad[7] else None),
                (db if ctx.needs_input_grad[8] else None))


def adapter_combine(o, w1, b1, w2, b2, x, attn, attn_bias, mlp_bias):
    return AdapterCombineFn.apply(o, w1, b1, w2, b2, x, attn, attn_bias, mlp_bias)


class LayerNormFn(torch.autograd.Function):
    """torch.nn.LayerNorm of the layer (megatron/model/norms.py:16-31)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        y, _, mean, rstd = ops.layernorm_fwd(x, w, b, eps)
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        need_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, grads = ops.layernorm_bwd(dy, x, w, mean, rstd, param_grads=need_p)
        dw, db = (grads[0], grads[1]) if need_p else (None, None)
        return dx, dw, db, None


def layer_norm(x, w, b, eps):
    return LayerNormFn.apply(x, w, b, eps)


class FinalLayerNormSbFn(torch.autograd.Function):
    """NormPipe (transformer.py:828-839) fused with _post_transformer_block's transpose(0, 1).contiguous()
    (gpt2_model.py:80-85): x [s, b, h] -> LayerNorm(x) as [b, s, h]; neither direction copies the activation."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        y, mean, rstd = ops.layernorm_fwd_sb(x, w, b, eps)
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        need_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, grads = ops.layernorm_bwd_sb(dy, x, w, mean, rstd, param_grads=need_p)
        dw, db = (grads[0], grads[1]) if need_p else (None, None)
        return dx, dw, db, None


def final_layer_norm_sb(x, w, b, eps):
    return FinalLayerNormSbFn.apply(x.contiguous(), w, b, eps)


class DualLayerNormFn(torch.autograd.Function):
    """(LN(x; w1, b1), LN(x; w2, b2)) from one read of x and shared statistics: the GPT-J residual variant
    normalises the same input twice (megatron/model/transformer.py:737)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, eps):
        y1, y2, mean, rstd = ops.layernorm_fwd(x, w1, b1, eps, w2, b2)
        ctx.save_for_backward(x, w1, w2, mean, rstd)
        return y1, y2

    @staticmethod
    def backward(ctx, dy1, dy2):
        x, w1, w2, mean, rstd = ctx.saved_tensors
        need_p = any(ctx.needs_input_grad[1:5])
        dx, grads = ops.layernorm_bwd(dy1, x, w1, mean, rstd, dy2=dy2, w2=w2, param_grads=need_p)
        if need_p:
            return dx, grads[0], grads[1], grads[2], grads[3], None
        return dx, None, None, None, None, None


class DualLayerNormResFn(torch.autograd.Function):
    """DualLayerNormFn that also hands x back for the residual branch, so that the residual's gradient arrives HERE and is
    added inside the LayerNorm backward kernel (fp32, one rounding) instead of by a separate autograd accumulation pass
    over [T, h]."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, eps):
        y1, y2, mean, rstd = ops.layernorm_fwd(x, w1, b1, eps, w2, b2)
        ctx.save_for_backward(x, w1, w2, mean, rstd)
        return y1, y2, x.view_as(x)

    @staticmethod
    def backward(ctx, dy1, dy2, dres):
        x, w1, w2, mean, rstd = ctx.saved_tensors
        need_p = any(ctx.needs_input_grad[1:5])
        dx, grads = ops.layernorm_bwd(dy1, x, w1, mean, rstd, dy2=dy2, w2=w2, dresidual=dres, param_grads=need_p)
        if need_p:
            return dx, grads[0], grads[1], grads[2], grads[3], None
        return dx, None, None, None, None, None


def dual_layer_norm_res(x, w1, b1, w2, b2, eps):
    """(LN1(x), LN2(x), x): use the returned x for the residual add."""
    return DualLayerNormResFn.apply(x, w1, b1, w2, b2, eps)


def dual_layer_norm(x, w1, b1, w2, b2, eps):
    return DualLayerNormFn.apply(x, w1, b1, w2, b2, eps)


class AttentionCoreFn(torch.autograd.Function):
    """rotary + scores + mask + softmax + P.V + permute of ParallelSelfAttention
    (megatron/model/transformer.py:555-630) on the fused QKV activation [s, b, np, 3*hn].
    The rotary is applied in place on `mixed` (the QKV GEMM's output, not needed again)."""

    @staticmethod
    def forward(ctx, mixed, cos, sin, mask, np_, hn, rotary_ndims, softmax_scale, dropout=None):
        s, b = mixed.size(0), mixed.size(1)
        qkv = mixed.view(s, b, np_, 3 * hn)
        if rotary_ndims > 0:
            ops.rotary_qk_inplace(qkv, cos, sin, s, b, np_, hn, rotary_ndims)
        out, lse = ops.attn_fwd(qkv, s, b, np_, hn, softmax_scale, mask, dropout=dropout)
        ctx.save_for_backward(qkv, out, lse, cos, sin, mask)
        ctx.dims = (s, b, np_, hn, rotary_ndims, softmax_scale)
        ctx.dropout = dropout  # (p, rng_state, pos): the backward regenerates the forward's mask from it
        return out

    @staticmethod
    def backward(ctx, dctx):
        qkv, out, lse, cos, sin, mask = ctx.saved_tensors
        s, b, np_, hn, d, scale = ctx.dims
        # the rotary backward rides in the dq / dk stores of the attention backward kernels
        dqkv = ops.attn_bwd(dctx, out, lse, qkv, s, b, np_, hn, scale, mask, rotary=(cos, sin, d, 0) if d > 0 else None,
                            dropout=ctx.dropout)
        return dqkv.view(s, b, np_ * 3 * hn), None, None, None, None, None, None, None, None


def attention_core(mixed, cos, sin, mask, np_, hn, rotary_ndims, softmax_scale, dropout=None):
    """dropout = (p, rng_state, pos) or None: attention dropout on the softmax output (ops.attn_fwd)."""
    return AttentionCoreFn.apply(mixed, cos, sin, mask, np_, hn, rotary_ndims, softmax_scale, dropout)


def bias_needed(ctx):
    return ctx.has_bias and ctx.needs_input_grad[2]


class QkvAttentionFn(torch.autograd.Function):
    """ctx = attention(rotary(x Wqkv^T + b)) - query_key_value + rotary + the attention core of ParallelSelfAttention
    (megatron/model/transformer.py:552-630) as one autograd node, so that the rotary embedding is the QKV GEMM's epilogue
    and its backward sits in the attention backward's stores: no pass over the [s, b, 3h] activation in either direction.
    Falls back to GEMM + in-place rotary for shapes the fused epilogue does not take (head size 128)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cos, sin, mask, np_, hn, rotary_ndims, softmax_scale, dropout=None):
        s, b = x.size(0), x.size(1)
        d = rotary_ndims
        if d > 0 and bias is not None and ops.qkv_rotary_supported(hn, d, weight.size(0), weight.size(1), s * b, b):
            mixed = ops.gemm_nt_qkv_rotary(x, weight, bias, cos, sin, b, hn, d)
        else:
            mixed = ops.gemm_nt(x, weight, bias, _lib.EPI_BIAS if bias is not None else _lib.EPI_NONE)
            if d > 0:
                ops.rotary_qk_inplace(mixed.view(s, b, np_, 3 * hn), cos, sin, s, b, np_, hn, d)
        qkv = mixed.view(s, b, np_, 3 * hn)
        out, lse = ops.attn_fwd(qkv, s, b, np_, hn, softmax_scale, mask, dropout=dropout)
        ctx.save_for_backward(x, weight, qkv, out, lse, cos, sin, mask)
        ctx.dims = (s, b, np_, hn, d, softmax_scale)
        ctx.has_bias = bias is not None
        ctx.dropout = dropout
        return out

    @staticmethod
    def backward(ctx, dctx):
        x, weight, qkv, out, lse, cos, sin, mask = ctx.saved_tensors
        s, b, np_, hn, d, scale = ctx.dims
        dmixed = ops.attn_bwd(dctx, out, lse, qkv, s, b, np_, hn, scale, mask, rotary=(cos, sin, d, 0) if d > 0 else None,
                              dropout=ctx.dropout)
        d2 = dmixed.view(s * b, np_ * 3 * hn)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(d2, weight_t(weight)).view(x.shape)
        dw, db = _wgrad_bias(d2, _flat2d(x), ctx.needs_input_grad[1], bias_needed(ctx))
        return dx, dw, db, None, None, None, None, None, None, None, None


def qkv_attention(x, weight, bias, cos, sin, mask, np_, hn, rotary_ndims, softmax_scale, dropout=None):
    return QkvAttentionFn.apply(x, weight, bias, cos, sin, mask, np_, hn, rotary_ndims, softmax_scale, dropout)


class BiasDropoutAddFn(torch.autograd.Function):
    """bias_dropout_add (megatron/model/fused_bias_dropout.py:28-34)."""

    @staticmethod
    def forward(ctx, x, bias, residual, p, training, seed, offset, rng_state=None):
        drop = training and p > 0
        if drop:
            out, keep = ops.bias_dropout_add(x, bias, residual, p, True, seed, offset, want_mask=True, rng_state=rng_state)
            ctx.save_for_backward(keep)
        else:
            out = ops.bias_dropout_add(x, bias, residual, 0.0, training)
        ctx.drop, ctx.p = drop, p
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dx = g
        if ctx.drop:
            (keep,) = ctx.saved_tensors
            dx = ops.dropout_bwd(g, keep, ctx.p)
        db = ops.colsum(_flat2d(dx)) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
        dres = g if (ctx.has_res and ctx.needs_input_grad[2]) else None
        return dx, db, dres, None, None, None, None, None


def bias_dropout_add(x, bias, residual, prob, training, seed=0, offset=0, rng_state=None):
    """bias may be a [cols] vector or its expand_as view (the reference passes `bias.expand_as(x)`)."""
    if bias is not None and bias.dim() > 1:
        # an expanded view repeats one row: every leading dimension has stride 0 (or size 1, whose stride is arbitrary)
        lead_ok = all(st == 0 or n == 1 for st, n in zip(bias.stride()[:-1], bias.shape[:-1]))
        bias = bias.reshape(-1, bias.size(-1))[0] if bias.stride(-1) == 1 and lead_ok else bias
        if bias.dim() > 1:
            raise RuntimeError("bias_dropout_add: bias must be a vector (or an expanded view of one)")
    return BiasDropoutAddFn.apply(x, bias, residual, float(prob), bool(training), int(seed), int(offset), rng_state)


class GptjCombineFn(torch.autograd.Function):
    """out = [x +] ((attn + b_a) + (mlp + b_m)): the GPT-J tail at hidden_dropout = 0 in one kernel."""

    @staticmethod
    def forward(ctx, x, attn, ba, mlp, bm):
        ctx.has_x = x is not None
        return ops.gptj_combine(x, attn, ba, mlp, bm)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        db = ops.colsum(_flat2d(g)) if (ctx.needs_input_grad[2] or ctx.needs_input_grad[4]) else None
        return (g if ctx.has_x else None), g, (db if ctx.needs_input_grad[2] else None), g, (db if ctx.needs_input_grad[4] else None)


def gptj_combine(x, attn, attn_bias, mlp, mlp_bias):
    return GptjCombineFn.apply(x, attn, attn_bias, mlp, mlp_bias)


class CrossEntropyFn(torch.autograd.Function):
    """Per-token cross entropy on 16-bit logits with fp32 math (megatron/mpu/cross_entropy.py:27-112 at TP = 1)."""

    @staticmethod
    def forward(ctx, logits2d, labels):
        loss, lse = ops.cross_entropy_fwd(logits2d, labels)
        ctx.save_for_backward(logits2d, labels, lse)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits2d, labels, lse = ctx.saved_tensors
        # the logits are dead after this: write the gradient over them (saves an 800 MB allocation at V = 50304)
        return ops.cross_entropy_bwd(logits2d, labels, lse, g.contiguous().float(), out=logits2d), None


def masked_cross_entropy(logits, labels, loss_mask):
    """cross_entropy (megatron/model/gpt2_model.py:52-70): masked mean of the per-token losses."""
    V = logits.size(-1)
    losses = CrossEntropyFn.apply(logits.reshape(-1, V), labels.reshape(-1).contiguous())
    lm = loss_mask.reshape(-1).float()
    return torch.sum(losses * lm) / lm.sum()


class LmHeadLossFn(torch.autograd.Function):
    """Bias-free LM head (transformer.py:816-839) + masked-mean cross entropy (gpt2_model.py:52-70) as ONE node, so that the
    [T, V] logits are read once: the loss kernel leaves (softmax - onehot) * mask / sum(mask) in the logits buffer, and the
    backward feeds that buffer to the data- / weight-gradient GEMMs and scales their [T, h] / [V, h] results by the loss's
    upstream gradient (a device scalar: no host sync, graph-capturable).  Same loss bits as `linear` +
    `masked_cross_entropy` up to the last place of logsumexp; bf16 only - with fp16 the unscaled gradient would lose what
    the reference's loss scaling protects - and tensor-parallel size 1 (the vocab-parallel loss needs its three exchanges).
    The mask sum, the per-row weights and the final masked mean are two one-block HIP launches (ops.loss_mask_weights /
    ops.masked_mean): no framework kernel runs in this node."""

    @staticmethod
    def forward(ctx, x, weight, labels, loss_mask):
        x2 = _flat2d(x)
        logits = ops.gemm_nt(x2, weight, None, _lib.EPI_NONE)
        row_w, den, lm = ops.loss_mask_weights(loss_mask)
        losses, _ = ops.cross_entropy_fwd_grad(logits, labels.reshape(-1).contiguous(), row_w)
        ctx.save_for_backward(x2, weight, logits)  # `logits` now holds the gradient for an upstream gradient of 1
        ctx.xshape = x.shape
        return ops.masked_mean(losses, lm, den)

    @staticmethod
    def backward(ctx, gup):
        x2, weight, g = ctx.saved_tensors
        gup = gup.reshape(1).float()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(g, weight_t(weight))
            dx = ops.scale_dev(dx, gup, out=dx).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dw = _wgrad(g, x2)
            dw = ops.scale_dev(dw, gup, out=dw)
        return dx, dw, None, None


def lm_head_loss_supported(x, weight):
    return x.dtype == torch.bfloat16 and ops.cross_entropy_fwd_grad_supported(weight.size(0))


def lm_head_loss(x, weight, labels, loss_mask):
    """x [b, s, h] (after the final norm), weight [V, h], labels / loss_mask [b, s] -> scalar loss."""
    return LmHeadLossFn.apply(x, weight, labels, loss_mask)


class FinalNormHeadLossFn(torch.autograd.Function):
    """_post_transformer_block's transpose (gpt2_model.py:80-85) + NormPipe (transformer.py:828-839) + the bias-free LM head
    (transformer.py:816-825) + the masked-mean cross entropy (gpt2_model.py:52-70) as ONE node on the layers' [s, b, h]
    output.  The LayerNorm reads [s, b, h] by stride and writes [b, s, h] (no transposed copy, forward or backward); the
    loss's upstream gradient (a device scalar) rides into the LayerNorm backward kernel, which is linear in dy, instead of
    costing a pass over the [T, h] gradient.  Numerics = LayerNormFn + LmHeadLossFn (same kernels, same roundings)."""

    @staticmethod
    def forward(ctx, x_sbh, ln_w, ln_b, eps, weight, labels, loss_mask):
        xn, mean, rstd = ops.layernorm_fwd_sb(x_sbh, ln_w, ln_b, eps)   # [b, s, h]
        x2 = _flat2d(xn)
        logits = ops.gemm_nt(x2, weight, None, _lib.EPI_NONE)
        row_w, den, lm = ops.loss_mask_weights(loss_mask)
        losses, _ = ops.cross_entropy_fwd_grad(logits, labels.reshape(-1).contiguous(), row_w)
        ctx.save_for_backward(x_sbh, ln_w, mean, rstd, x2, weight, logits)
        return ops.masked_mean(losses, lm, den)

    @staticmethod
    def backward(ctx, gup):
        x_sbh, ln_w, mean, rstd, x2, weight, g = ctx.saved_tensors
        gup = gup.reshape(1).float()
        dx = dlw = dlb = dw = None
        need_ln_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        if ctx.needs_input_grad[0] or need_ln_p:
            dxn = ops.gemm_nt(g, weight_t(weight))  # [T, h] in [b, s] row order, unscaled
            dx, grads = ops.layernorm_bwd_sb(dxn, x_sbh, ln_w, mean, rstd, dy_scale=gup, param_grads=need_ln_p)
            if need_ln_p:
                dlw, dlb = grads
        if ctx.needs_input_grad[4]:
            dw = _wgrad(g, x2)
            dw = ops.scale_dev(dw, gup, out=dw)
        return dx, dlw, dlb, None, dw, None, None


def final_norm_head_loss(x_sbh, ln_w, ln_b, eps, weight, labels, loss_mask):
    """x_sbh: the last layer's [s, b, h] output; labels / loss_mask [b, s] -> scalar loss."""
    return FinalNormHeadLossFn.apply(x_sbh, ln_w, ln_b, eps, weight, labels, loss_mask)


class ImagePrefixTailFn(torch.autograd.Function):
    """ImagePrefix.forward behind its encoder (image_prefix.py:232-252): LayerNorm(dropout(x proj^T + b)) for the few rows of a
    step in one launch (csrc/prefix.hip); the backward is LayerNorm backward -> dropout backward -> the linear's gradients,
    all HIP launches."""

    @staticmethod
    def forward(ctx, x2, w, b, ln_w, ln_b, eps, p, training, rng_state, offset):
        out, ln_in, keep, mean, rstd = ops.prefix_tail_fwd(x2, w, b, ln_w, ln_b, eps, p, training, rng_state, offset)
        ctx.save_for_backward(x2, w, ln_w, ln_in, mean, rstd, keep)
        ctx.p = p
        return out

    @staticmethod
    def backward(ctx, dy):
        x2, w, ln_w, ln_in, mean, rstd, keep = ctx.saved_tensors
        need_ln_p = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        dyl, grads = ops.layernorm_bwd(dy.contiguous(), ln_in, ln_w, mean, rstd, param_grads=need_ln_p)
        dlw, dlb = (grads[0], grads[1]) if need_ln_p else (None, None)
        if keep is not None:
            dyl = ops.dropout_bwd(dyl, keep, ctx.p)
        dx = ops.gemm_nt(dyl, weight_t(w)) if ctx.needs_input_grad[0] else None
        dw, db = _wgrad_bias(dyl, x2, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dw, db, dlw, dlb, None, None, None, None, None


def image_prefix_tail(enc_out, w, b, ln_w, ln_b, eps, p=0.0, training=False, rng_state=None, offset=0):
    """enc_out [b, enc] (pooled encoder output -> one prefix token, image_prefix.py:249-250) or [b, P, enc] -> [b, P, h].
    rng_state / offset: the rng tracker's device Philox state and this draw's position (mpu/random.py draw_device), needed
    when training with p > 0.  Row counts beyond the fused kernel's range take the GEMM + dropout + LayerNorm launches."""
    if enc_out.dim() == 2:
        enc_out = enc_out.unsqueeze(1)
    bsz, P, enc = enc_out.shape
    h = w.size(0)
    x2 = enc_out.reshape(bsz * P, enc)
    if ops.prefix_tail_supported(bsz * P, enc, h):
        return ImagePrefixTailFn.apply(x2, w, b, ln_w, ln_b, float(eps), float(p), bool(training), rng_state, int(offset)).view(bsz, P, h)
    y = linear(x2, w, b)
    if training and p > 0:
        y = bias_dropout_add(y, None, None, p, True, 0, offset, rng_state=rng_state)
    return layer_norm(y, ln_w, ln_b, eps).view(bsz, P, h)


class EmbedPrefixFn(torch.autograd.Function):
    """Word embeddings + image-prefix concat + transpose to [s, b, h] (megatron/model/word_embeddings.py:135-205,
    gpt2_model.py:73-77) for a FROZEN embedding table: only the prefix rows carry a gradient back."""

    @staticmethod
    def forward(ctx, prefix, table, tokens):
        ctx.P = 0 if prefix is None else prefix.size(1)
        return ops.embed_prefix(table.detach(), tokens, None if prefix is None else prefix.detach())

    @staticmethod
    def backward(ctx, g):
        dprefix = None
        if ctx.P > 0 and ctx.needs_input_grad[0]:
            dprefix = g[:ctx.P].transpose(0, 1).contiguous()  # [b, P, h]: P rows of the gradient
        return dprefix, None, None


def embed_prefix(prefix, table, tokens):
    """First-layer input [s, b, h].  A trainable table takes the unfused path (its gradient is a scatter-add)."""
    if table.requires_grad:
        emb = torch.nn.functional.embedding(tokens, table)
        if prefix is not None:
            emb = torch.cat((prefix, emb[:, : tokens.size(1) - prefix.size(1), :]), dim=1)
        return emb.transpose(0, 1).contiguous()
    return EmbedPrefixFn.apply(prefix, table, tokens)
