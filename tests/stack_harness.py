"""2-layer loss-curve harness shared by the oracle test (CPU) and the HIP parity test (GPU).
Mirrors the stack tests/golden/make_golden.py:gen_stack built from the reference's layers."""
import numpy as np
import torch

import golden_params
from oracle import neox_layer_oracle as O


def build_params(meta, dtype, device="cpu"):
    init = golden_params.stack_init(np.random.default_rng(meta["seed"]), meta["V"], meta["h"], meta["L"])
    return {k: v.to(dtype).to(device).requires_grad_(True) for k, v in init.items()}


def run_oracle_stack(meta, tokens, loss_mask, dtype):
    p = build_params(meta, dtype)
    cfg = O.LayerConfig(meta["h"], meta["n"], rotary_pct=meta["rotary_pct"], gpt_j_residual=meta["gpt_j_residual"],
                        bias_gelu_fusion=meta["bias_gelu_fusion"])
    opt = torch.optim.Adam(list(p.values()), lr=meta["lr"], betas=tuple(meta["betas"]), eps=meta["eps"])
    mask = O.causal_mask(meta["s"])
    losses = []
    for _ in range(meta["steps"]):
        opt.zero_grad()
        x = torch.nn.functional.embedding(tokens[:, :-1], p["emb"]).transpose(0, 1).contiguous()
        for i in range(meta["L"]):
            lp = {k: p[f"l{i}.{k}"] for k in golden_params.PARAM_ORDER}
            x = O.layer_forward(cfg, [lp], x, mask)
        x = x.transpose(0, 1).contiguous()
        logits = torch.nn.functional.linear(O.layer_norm(x, p["fln.w"], p["fln.b"]), p["head"])
        loss = O.cross_entropy_loss(logits, tokens[:, 1:], loss_mask)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses
