"""Builds the HIP layer from seeded master parameters and runs it against the CPU oracle / reference fixtures.
Used by tests/test_layer_gpu.py and __graft_entry__.smoke() (both may use the oracle; the product never does)."""
from types import SimpleNamespace

import numpy as np
import torch

import golden_params
from oracle import neox_layer_oracle as O


def make_neox_args(h, n, dtype, rotary_pct, gpt_j, gelu_fusion, tp=1, softmax_fusion=False, L=2, s=2048):
    return SimpleNamespace(
        hidden_size=h, num_attention_heads=n, num_layers=L, seq_length=s,
        precision={torch.bfloat16: "bfloat16", torch.float16: "fp16"}[dtype], params_dtype=dtype, pos_emb="rotary",
        rotary_pct=rotary_pct, rotary_emb_base=10000, norm="layernorm", layernorm_epsilon=1e-5, activation="gelu",
        bias_gelu_fusion=gelu_fusion, bias_dropout_fusion=False,
        scaled_upper_triang_masked_softmax_fusion=softmax_fusion, scaled_masked_softmax_fusion=False,
        apply_query_key_layer_scaling=False, attention_softmax_in_fp32=False, attention_dropout=0.0, hidden_dropout=0.0,
        gpt_j_residual=gpt_j, gpt_j_tied=False, attention_config=["global"] * L, use_mup=False,
        use_cpu_initialization=True, model_parallel_size=tp, fp32_allreduce=False, onnx_safe=False)


def build_hip_layer(args, master, device, rank=0, tp=1):
    from multimodal_amd.model import ParallelTransformerLayer, gpt2_attention_mask_func
    layer = ParallelTransformerLayer(neox_args=args, attention_mask_func=gpt2_attention_mask_func,
                                     init_method=torch.nn.init.zeros_, output_layer_init_method=torch.nn.init.zeros_,
                                     layer_number=0, rotary=True)
    with torch.no_grad():
        sd = dict(layer.named_parameters())
        for k in golden_params.PARAM_ORDER:
            sd[k].copy_(golden_params.shard(k, master[k], rank, tp).to(args.params_dtype))
    return layer.to(device)


def oracle_layer_fp32(meta_or_cfg, master, x16, dy16, dtype, tp=1, adapters=None):
    """fp32 oracle on the SAME 16-bit-rounded parameters and inputs (the yardstick for 16-bit error).  `adapters`
    (golden_params.adapter_params): every simulated rank gets its own copy, as every reference rank holds one."""
    cfg = meta_or_cfg
    shared = {k: master[k].to(dtype).float().requires_grad_(True) for k in golden_params.PARAM_ORDER
              if golden_params.shard(k, master[k], 0, 2) is master[k]}
    shards = [{k: shared[k] if k in shared else golden_params.shard(k, master[k], r, tp).to(dtype).float().requires_grad_(True)
               for k in golden_params.PARAM_ORDER} for r in range(tp)]
    if adapters is not None:
        for sh in shards:
            sh.update({k: v.to(dtype).float().requires_grad_(True) for k, v in adapters.items()})
    xin = x16.float().clone().requires_grad_(True)
    y = O.layer_forward(cfg, shards, xin, O.causal_mask(x16.size(0)))
    y.backward(dy16.float())
    return y.detach(), xin.grad, shards


def run_hip_layer(layer, x16, dy16, device, mask=None):
    xin = x16.to(device).clone().requires_grad_(True)
    s = x16.size(0)
    if mask is None:
        mask = O.causal_mask(s).to(device)
    layer.train()
    y = layer(xin, mask)
    y.backward(dy16.to(device))
    torch.cuda.synchronize()
    grads = {k: (p.grad.detach().cpu() if p.grad is not None else None) for k, p in layer.named_parameters()}
    return y.detach().cpu(), xin.grad.detach().cpu(), grads


def smoke_check(device="cuda:0"):
    """One small 410M-style layer (gpt-j residual, tanh bias-GeLU, partial rotary) fwd+bwd vs the fp32 oracle."""
    h, n, s, b = 256, 4, 128, 2
    dtype = torch.bfloat16
    rng = np.random.default_rng(11)
    master = golden_params.master_params(rng, h)
    x = golden_params.rng_t(rng, (s, b, h)).to(dtype)
    dy = golden_params.rng_t(rng, (s, b, h)).to(dtype)
    args = make_neox_args(h, n, dtype, 0.25, True, True, softmax_fusion=True)
    from multimodal_amd import mpu
    if not mpu.model_parallel_is_initialized():
        mpu.initialize_model_parallel(1)
    layer = build_hip_layer(args, master, torch.device(device))
    y, dx, _ = run_hip_layer(layer, x, dy, torch.device(device))
    cfg = O.LayerConfig(h, n, rotary_pct=0.25, gpt_j_residual=True, bias_gelu_fusion=True)
    y32, dx32, _ = oracle_layer_fp32(cfg, master, x, dy, dtype)
    ey = (y.float() - y32).abs().max().item() / y32.abs().max().item()
    ex = (dx.float() - dx32).abs().max().item() / dx32.abs().max().item()
    assert ey < 2e-2 and ex < 3e-2, (ey, ex)
    return {"y_rel_err": ey, "dx_rel_err": ex}
