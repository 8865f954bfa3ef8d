"""CPU ORACLE - test infrastructure, NOT the product.

BASELINE.md section 3 "workload A" / BASELINE.json configs[0]: the reference's `19M.yml + cpu_mock_config.yml` text-only
GPT-NeoX train step (configs/19M.yml:6-17: L=6, h=512, 8 heads, s=2048, rotary_pct 1.0, sequential residual, exact-erf
GeLU, no fusions; micro-batch 4, Adam) restated on the oracle's layer (oracle/neox_layer_oracle.py) in fp32 on the host:
word embedding -> 6 x ParallelTransformerLayer -> final LayerNorm -> untied LM head -> masked-mean cross entropy
(megatron/model/gpt2_model.py:52-70) -> backward -> torch.optim.Adam step.  The reference has no CPU training mode
(initialize_megatron asserts CUDA, megatron/initialize.py:42-44) and its Python cannot travel to the GPU box, so this
restatement - pinned layer by layer against reference fixtures in tests/test_oracle_golden.py - is what bench.py times
as `cpu_baseline.workload_a`.  Only tests/ and bench.py's cpu_baseline leg may import this module.
"""
import math
import time

import torch

from . import neox_layer_oracle as O

CONFIG_19M = dict(hidden_size=512, num_attention_heads=8, num_layers=6, seq_length=2048, micro_batch=4, vocab=50304,
                  rotary_pct=1.0, gpt_j_residual=False, bias_gelu_fusion=False)

LAYER_KEYS = ["input_layernorm.weight", "input_layernorm.bias", "attention.query_key_value.weight",
              "attention.query_key_value.bias", "attention.dense.weight", "attention.dense.bias",
              "post_attention_layernorm.weight", "post_attention_layernorm.bias", "mlp.dense_h_to_4h.weight",
              "mlp.dense_h_to_4h.bias", "mlp.dense_4h_to_h.weight", "mlp.dense_4h_to_h.bias"]


def init_params(cfg, seed=1234):
    """small_init / wang_init (megatron/model/init_functions.py:138-167), zero biases, unit LayerNorm weights."""
    g = torch.Generator().manual_seed(seed)
    h, L, V = cfg["hidden_size"], cfg["num_layers"], cfg["vocab"]
    small, wang = (2 / (5 * h)) ** 0.5, 2 / L / h ** 0.5
    nrm = lambda shape, std: (torch.randn(*shape, generator=g) * std).requires_grad_(True)
    p = {"emb": nrm((V, h), small), "fln.w": torch.ones(h, requires_grad=True), "fln.b": torch.zeros(h, requires_grad=True),
         "head": nrm((V, h), small)}
    shapes = {"attention.query_key_value.weight": ((3 * h, h), small), "attention.dense.weight": ((h, h), wang),
              "mlp.dense_h_to_4h.weight": ((4 * h, h), small), "mlp.dense_4h_to_h.weight": ((h, 4 * h), wang)}
    for i in range(L):
        for k in LAYER_KEYS:
            if k in shapes:
                p[f"l{i}.{k}"] = nrm(*shapes[k])
            elif k.endswith("layernorm.weight"):
                p[f"l{i}.{k}"] = torch.ones(h, requires_grad=True)
            else:
                n = {"attention.query_key_value.bias": 3 * h, "mlp.dense_h_to_4h.bias": 4 * h}.get(k, h)
                p[f"l{i}.{k}"] = torch.zeros(n, requires_grad=True)
    return p


def synthetic_batch(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    b, s = cfg["micro_batch"], cfg["seq_length"]
    tokens = torch.randint(0, min(cfg["vocab"], 50277), (b, s + 1), generator=g)
    loss_mask = torch.ones(b, s)
    loss_mask[:, -17:] = 0.0  # padded caption tail (megatron/utils.py:96-101)
    return tokens, loss_mask


def loss_of(cfg, p, tokens, loss_mask):
    lc = O.LayerConfig(cfg["hidden_size"], cfg["num_attention_heads"], rotary_pct=cfg["rotary_pct"],
                       gpt_j_residual=cfg["gpt_j_residual"], bias_gelu_fusion=cfg["bias_gelu_fusion"])
    s = tokens.size(1) - 1
    mask = O.causal_mask(s)
    x = torch.nn.functional.embedding(tokens[:, :-1], p["emb"]).transpose(0, 1).contiguous()   # [b,s,h] -> [s,b,h]
    for i in range(cfg["num_layers"]):
        x = O.layer_forward(lc, [{k: p[f"l{i}.{k}"] for k in LAYER_KEYS}], x, mask)
    x = x.transpose(0, 1).contiguous()
    logits = torch.nn.functional.linear(O.layer_norm(x, p["fln.w"], p["fln.b"]), p["head"])
    return O.cross_entropy_loss(logits, tokens[:, 1:], loss_mask)


def train_steps(cfg, steps, lr=1e-3, seed=1234, budget_s=None, fixed_batch=False):
    """Runs up to `steps` Adam steps (stops early once `budget_s` seconds have passed, after at least one step).
    Returns (losses, seconds per step list)."""
    p = init_params(cfg, seed)
    opt = torch.optim.Adam(list(p.values()), lr=lr, betas=(0.9, 0.95), eps=1e-8)   # configs/19M.yml:23-30
    losses, times = [], []
    t_start = time.time()
    for it in range(steps):
        tokens, lm = synthetic_batch(cfg, seed=0 if fixed_batch else it)
        t0 = time.time()
        opt.zero_grad()
        loss = loss_of(cfg, p, tokens, lm)
        loss.backward()
        opt.step()
        times.append(time.time() - t0)
        losses.append(float(loss.detach()))
        if budget_s is not None and time.time() - t_start > budget_s:
            break
    return losses, times


def flops_per_step(cfg):
    h, L, s, b, V = cfg["hidden_size"], cfg["num_layers"], cfg["seq_length"], cfg["micro_batch"], cfg["vocab"]
    T = s * b
    return 3 * (L * (24 * T * h * h + 4 * T * s * h) + 2 * T * h * V)


def expected_initial_loss(cfg):
    return math.log(cfg["vocab"])
