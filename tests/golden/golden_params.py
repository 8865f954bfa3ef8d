"""Seeded parameter/input builders shared by make_golden.py and the tests (numpy PCG64 streams
are stable across numpy versions, so the big-layer fixture can regenerate its weights from a seed)."""
import numpy as np
import torch


def rng_t(rng, shape, scale=1.0, shift=0.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale + shift).astype(np.float32))


PARAM_ORDER = [
    "input_layernorm.weight", "input_layernorm.bias",
    "attention.query_key_value.weight", "attention.query_key_value.bias",
    "attention.dense.weight", "attention.dense.bias",
    "post_attention_layernorm.weight", "post_attention_layernorm.bias",
    "mlp.dense_h_to_4h.weight", "mlp.dense_h_to_4h.bias",
    "mlp.dense_4h_to_h.weight", "mlp.dense_4h_to_h.bias",
]


def master_params(rng, h):
    """Full (un-sharded) fp32 master parameters with NON-zero biases."""
    std = (2 / (5 * h)) ** 0.5
    return {
        "input_layernorm.weight": rng_t(rng, (h,), 0.1, 1.0),
        "input_layernorm.bias": rng_t(rng, (h,), 0.1),
        "attention.query_key_value.weight": rng_t(rng, (3 * h, h), std),
        "attention.query_key_value.bias": rng_t(rng, (3 * h,), 0.1),
        "attention.dense.weight": rng_t(rng, (h, h), std),
        "attention.dense.bias": rng_t(rng, (h,), 0.1),
        "post_attention_layernorm.weight": rng_t(rng, (h,), 0.1, 1.0),
        "post_attention_layernorm.bias": rng_t(rng, (h,), 0.1),
        "mlp.dense_h_to_4h.weight": rng_t(rng, (4 * h, h), std),
        "mlp.dense_h_to_4h.bias": rng_t(rng, (4 * h,), 0.1),
        "mlp.dense_4h_to_h.weight": rng_t(rng, (h, 4 * h), std),
        "mlp.dense_4h_to_h.bias": rng_t(rng, (h,), 0.1),
    }


def shard(name, w, rank, p):
    """The reference's TP sharding (mpu/layers.py:53-93): Column -> dim 0, Row -> dim 1,
    Row biases and LayerNorm replicated."""
    if p == 1:
        return w
    if name in ("attention.query_key_value.weight", "attention.query_key_value.bias",
                "mlp.dense_h_to_4h.weight", "mlp.dense_h_to_4h.bias"):
        return w.chunk(p, dim=0)[rank].contiguous()
    if name in ("attention.dense.weight", "mlp.dense_4h_to_h.weight"):
        return w.chunk(p, dim=1)[rank].contiguous()
    return w


def causal_mask(s):
    return torch.tril(torch.ones((1, s, s))).view(1, 1, s, s) < 0.5


def stack_init(rng, V, h, L):
    """Initial weights of the 2-layer loss-curve stack, in the generator's draw order."""
    init = {"emb": rng_t(rng, (V, h), 0.05)}
    for i in range(L):
        m = master_params(rng, h)
        for k in PARAM_ORDER:
            init[f"l{i}.{k}"] = m[k]
    init["fln.w"] = rng_t(rng, (h,), 0.1, 1.0)
    init["fln.b"] = rng_t(rng, (h,), 0.1)
    init["head"] = rng_t(rng, (V, h), 0.05)
    return init


ADAPTER_KEYS = [f"{loc}.adapter.{i}.{wb}" for loc in ("attention", "mlp") for i in (0, 2) for wb in ("weight", "bias")]


def adapter_params(rng, h, r):
    """Adapter weights (megatron/model/adapter.py:18-25 shapes) for both wrappers, in ADAPTER_KEYS order.  The first Linear's
    bias is pushed to +-1 and its weight kept small (0.004, the order of the reference's own init std 1e-3, adapter.py:28-33)
    so that the ReLU inputs, 1 +- ~0.15, stay away from the kink: a kink crossing differs between ANY two precisions, costs a
    whole W1 row in dx, and says nothing about parity."""
    out = {}
    for loc in ("attention", "mlp"):
        for i, shape in ((0, (r, h)), (2, (h, r))):
            w = rng_t(rng, shape, 0.004 if i == 0 else 0.05)
            b = rng_t(rng, (shape[0],), 0.05)
            if i == 0:
                b = torch.where(torch.arange(shape[0]) % 2 == 0, 1.0, -1.0) + b
            out[f"{loc}.adapter.{i}.weight"], out[f"{loc}.adapter.{i}.bias"] = w, b
    return out
