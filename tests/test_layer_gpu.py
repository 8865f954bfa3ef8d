"""GPU parity of the whole ParallelTransformerLayer (HIP path, through the C ABI) against
(a) the fixtures the reference itself produced on CPU in bf16 and (b) the fp32 CPU oracle.

Stated bf16 tolerance: the reference's own bf16 run differs from the fp32 evaluation of the same
rounded weights/inputs by E_ref = max|y_ref_bf16 - y_fp32|; the HIP layer must satisfy
max|y_hip - y_fp32| <= 2 * E_ref (+ 1 bf16 ulp of the largest output), likewise for dX and the parameter grads."""
import glob
import os

import numpy as np
import pytest
import torch

import golden_io
import golden_params
import layer_harness as LH
from oracle import neox_layer_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _mpu(cuda_device):
    from multimodal_amd import mpu
    mpu.destroy_model_parallel()
    mpu.initialize_model_parallel(1)
    yield
    mpu.destroy_model_parallel()


def _files():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(golden_io.HERE, "layer_*_bf16_*_tp1.npz")))


def _check(name, got, ref16, ref32, factor=2.0):
    got, ref16 = got.float(), ref16.float()
    scale = ref32.abs().max().item()
    e_ref = (ref16 - ref32).abs().max().item()
    e_hip = (got - ref32).abs().max().item()
    tol = factor * e_ref + 2.0 ** -8 * scale
    assert e_hip <= tol, f"{name}: HIP err {e_hip:.4g} vs reference-bf16 err {e_ref:.4g} (scale {scale:.3g}, tol {tol:.4g})"
    rel = ((got - ref32).norm() / ref32.norm().clamp_min(1e-9)).item()
    rel_ref = ((ref16 - ref32).norm() / ref32.norm().clamp_min(1e-9)).item()
    assert rel <= factor * rel_ref + 2.0 ** -9, f"{name}: relative L2 {rel:.4g} vs reference-bf16 {rel_ref:.4g}"


@pytest.mark.parametrize("name", _files())
@pytest.mark.parametrize("softmax_fusion", [False, True])
def test_layer_matches_reference_bf16(cuda_device, name, softmax_fusion):
    t, meta = golden_io.load(name)
    if meta["h"] > 128 and softmax_fusion:
        pytest.skip("one pass over the big case is enough")
    dtype = torch.bfloat16
    master = golden_params.master_params(np.random.default_rng(meta["seed"]), meta["h"])
    args = LH.make_neox_args(meta["h"], meta["n"], dtype, meta["rotary_pct"], meta["gpt_j_residual"],
                             meta["bias_gelu_fusion"], softmax_fusion=softmax_fusion)
    layer = LH.build_hip_layer(args, master, cuda_device)
    y, dx, grads = LH.run_hip_layer(layer, t["x"], t["dy"], cuda_device)
    cfg = O.LayerConfig(meta["h"], meta["n"], rotary_pct=meta["rotary_pct"], gpt_j_residual=meta["gpt_j_residual"],
                        bias_gelu_fusion=meta["bias_gelu_fusion"])
    y32, dx32, shards = LH.oracle_layer_fp32(cfg, master, t["x"], t["dy"], dtype)
    _check("y", y, t["y"], y32)
    _check("dx", dx, t["dx"], dx32)
    for k in golden_params.PARAM_ORDER:
        if "g." + k in t:
            _check("grad " + k, grads[k], t["g." + k], shards[0][k].grad, factor=3.0)
    # state-dict compatibility with the reference layer (12 tensors + the rotary buffer)
    keys = set(layer.state_dict().keys())
    assert set(golden_params.PARAM_ORDER) <= keys and "attention.rotary_emb.inv_freq" in keys


def test_layer_explicit_general_mask_vs_oracle(cuda_device):
    """A non-causal byte mask goes through the explicit-mask kernel (-10000 fill), like the reference's eager path."""
    h, n, s, b = 128, 2, 64, 2
    dtype = torch.bfloat16
    rng = np.random.default_rng(3)
    master = golden_params.master_params(rng, h)
    x = golden_params.rng_t(rng, (s, b, h)).to(dtype)
    dy = golden_params.rng_t(rng, (s, b, h)).to(dtype)
    mask = torch.from_numpy(rng.random((1, 1, s, s)) < 0.3)
    mask[..., 7, :] = True
    args = LH.make_neox_args(h, n, dtype, 1.0, False, False)
    layer = LH.build_hip_layer(args, master, cuda_device)
    y, dx, _ = LH.run_hip_layer(layer, x, dy, cuda_device, mask=mask.to(cuda_device))
    cfg = O.LayerConfig(h, n, rotary_pct=1.0, gpt_j_residual=False, bias_gelu_fusion=False)
    p32 = {k: master[k].to(dtype).float() for k in golden_params.PARAM_ORDER}
    xin = x.float().requires_grad_(True)
    y32 = O.layer_forward(cfg, [p32], xin, mask)
    y32.backward(dy.float())
    assert (y.float() - y32.detach()).abs().max() <= 2 ** -6 * y32.abs().max()
    assert (dx.float() - xin.grad).abs().max() <= 2 ** -5 * xin.grad.abs().max()


def test_layer_with_adapters_vs_oracle(cuda_device):
    """MAGMA adapters wrapped around .attention and .mlp (adapter.py:119-151), frozen LM: only adapter params get grads."""
    from multimodal_amd.model import add_adapters
    h, n, s, b = 512, 8, 128, 2
    dtype = torch.bfloat16
    rng = np.random.default_rng(21)
    master = golden_params.master_params(rng, h)
    x = golden_params.rng_t(rng, (s, b, h)).to(dtype)
    dy = golden_params.rng_t(rng, (s, b, h)).to(dtype)
    args = LH.make_neox_args(h, n, dtype, 0.25, True, True, softmax_fusion=True)
    layer = LH.build_hip_layer(args, master, cuda_device)
    holder = torch.nn.ModuleList([layer])
    add_adapters(args, holder, downsample_factor=8, location="mlp")
    add_adapters(args, holder, downsample_factor=8, location="attention")
    ad = {}
    with torch.no_grad():
        for loc in ("attention", "mlp"):
            for i, shape in ((0, (h // 8, h)), (2, (h, h // 8))):
                w = golden_params.rng_t(rng, shape, 0.05)
                bb = golden_params.rng_t(rng, (shape[0],), 0.05)
                if i == 0:  # keep the ReLU inputs away from 0: a kink crossing differs between ANY two precisions
                    bb = torch.where(torch.arange(shape[0]) % 2 == 0, 1.0, -1.0) + bb
                getattr(layer, loc).adapter[i].weight.copy_(w)
                getattr(layer, loc).adapter[i].bias.copy_(bb)
                ad[f"{loc}.adapter.{i}.weight"], ad[f"{loc}.adapter.{i}.bias"] = w, bb
    for nme, p in layer.named_parameters():  # freeze_lm (megatron/training.py:454-459)
        p.requires_grad = "adapter" in nme
    keys = set(layer.state_dict().keys())
    assert "attention.attn_block.query_key_value.weight" in keys and "mlp.adapter.2.bias" in keys
    y, dx, grads = LH.run_hip_layer(layer, x, dy, cuda_device)
    cfg = O.LayerConfig(h, n, rotary_pct=0.25, gpt_j_residual=True, bias_gelu_fusion=True)

    def oracle(cast):
        p = {k: cast(master[k].to(dtype)) for k in golden_params.PARAM_ORDER}
        p.update({k: cast(v.to(dtype)).requires_grad_(True) for k, v in ad.items()})
        xin = cast(x).requires_grad_(True)
        yy = O.layer_forward(cfg, [p], xin, O.causal_mask(s))
        yy.backward(cast(dy))
        return yy.detach(), xin.grad, {k: p[k].grad for k in ad}

    y32, dx32, g32 = oracle(lambda t_: t_.float())
    y16, dx16, g16 = oracle(lambda t_: t_.clone())  # the eager bf16 evaluation the reference would do: the yardstick
    _check("y", y, y16, y32)
    _check("dx", dx, dx16, dx32)
    for k in ad:
        assert grads[k] is not None, k
        _check("grad " + k, grads[k], g16[k], g32[k], factor=3.0)
    assert grads["attention.attn_block.dense.weight"] is None and grads["input_layernorm.weight"] is None


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(golden_io.HERE, "adapter_layer_*_bf16_*_tp1.npz"))))
def test_layer_with_adapters_matches_reference_fixture(cuda_device, name):
    """The HIP layer wrapped by add_adapters against the REFERENCE layer wrapped by its own add_adapters
    (tests/golden/make_golden.py::run_adapter_case; megatron/model/adapter.py:95-151), both residual variants, LM frozen."""
    from multimodal_amd.model import add_adapters
    t, meta = golden_io.load(name)
    dtype = torch.bfloat16
    rng = np.random.default_rng(meta["seed"])
    master = golden_params.master_params(rng, meta["h"])
    golden_params.rng_t(rng, (meta["s"], meta["b"], meta["h"]))  # x, dy: the generator's draw order
    golden_params.rng_t(rng, (meta["s"], meta["b"], meta["h"]))
    ad = golden_params.adapter_params(rng, meta["h"], meta["h"] // meta["adapter_downsample_factor"])
    args = LH.make_neox_args(meta["h"], meta["n"], dtype, meta["rotary_pct"], meta["gpt_j_residual"], meta["bias_gelu_fusion"],
                             softmax_fusion=meta["gpt_j_residual"])
    layer = LH.build_hip_layer(args, master, cuda_device)
    holder = torch.nn.ModuleList([layer])
    add_adapters(args, holder, downsample_factor=meta["adapter_downsample_factor"], location="mlp")
    add_adapters(args, holder, downsample_factor=meta["adapter_downsample_factor"], location="attention")
    sd = dict(layer.named_parameters())
    with torch.no_grad():
        for k in golden_params.ADAPTER_KEYS:
            sd[k].copy_(ad[k].to(dtype))
    for nme, p in layer.named_parameters():
        p.requires_grad = "adapter" in nme
    assert set(layer.state_dict().keys()) == set(meta["state_dict_keys"])
    y, dx, grads = LH.run_hip_layer(layer, t["x"], t["dy"], cuda_device)
    cfg = O.LayerConfig(meta["h"], meta["n"], rotary_pct=meta["rotary_pct"], gpt_j_residual=meta["gpt_j_residual"],
                        bias_gelu_fusion=meta["bias_gelu_fusion"])
    y32, dx32, shards = LH.oracle_layer_fp32(cfg, master, t["x"], t["dy"], dtype, adapters=ad)
    _check("y", y, t["y"], y32)
    _check("dx", dx, t["dx"], dx32)
    for k in golden_params.ADAPTER_KEYS:
        _check("grad " + k, grads[k], t["g." + k], shards[0][k].grad, factor=3.0)
    assert all(g is None for k, g in grads.items() if "adapter" not in k)


def _full_shape_layer_vs_fp32_oracle(cuda_device, h, n, s, b, rotary_pct, gpt_j, gelu_fusion, adapter_factor, seed):
    """One layer at a config's FULL shape against the fp32 CPU oracle on the same bf16-rounded weights and inputs.
    adapter_factor > 0: MAGMA adapters on .attention and .mlp with the LM frozen (only adapter gradients exist);
    adapter_factor == 0: plain layer, every parameter trainable.  Stated tolerance: relative L2 error <= 8e-3 for y and dX
    (the small reference fixtures sit at 3e-3 for both the HIP layer and the reference's own bf16 run), max |err| <= 2^-5 of
    the largest value, parameter gradients relative L2 <= 2e-2."""
    from multimodal_amd.model import add_adapters
    dtype = torch.bfloat16
    rng = np.random.default_rng(seed)
    master = golden_params.master_params(rng, h)
    x = golden_params.rng_t(rng, (s, b, h)).to(dtype)
    dy = golden_params.rng_t(rng, (s, b, h)).to(dtype)
    ad = golden_params.adapter_params(rng, h, h // adapter_factor) if adapter_factor else {}
    args = LH.make_neox_args(h, n, dtype, rotary_pct, gpt_j, gelu_fusion, softmax_fusion=gpt_j)
    layer = LH.build_hip_layer(args, master, cuda_device)
    if adapter_factor:
        holder = torch.nn.ModuleList([layer])
        add_adapters(args, holder, downsample_factor=adapter_factor, location="mlp")
        add_adapters(args, holder, downsample_factor=adapter_factor, location="attention")
        sd = dict(layer.named_parameters())
        with torch.no_grad():
            for k in golden_params.ADAPTER_KEYS:
                sd[k].copy_(ad[k].to(dtype))
        for nme, p in layer.named_parameters():
            p.requires_grad = "adapter" in nme
    y, dx, grads = LH.run_hip_layer(layer, x, dy, cuda_device)
    torch.set_num_threads(min(16, torch.get_num_threads() if torch.get_num_threads() > 1 else 16))
    cfg = O.LayerConfig(h, n, rotary_pct=rotary_pct, gpt_j_residual=gpt_j, bias_gelu_fusion=gelu_fusion)
    p32 = {k: master[k].to(dtype).float().requires_grad_(not adapter_factor) for k in golden_params.PARAM_ORDER}
    p32.update({k: v.to(dtype).float().requires_grad_(True) for k, v in ad.items()})
    xin = x.float().requires_grad_(True)
    y32 = O.layer_forward(cfg, [p32], xin, O.causal_mask(s))
    y32.backward(dy.float())
    y32 = y32.detach()

    def rel(a, ref):
        return float((a.float() - ref).norm() / ref.norm())

    assert rel(y, y32) <= 8e-3 and rel(dx, xin.grad) <= 8e-3, (rel(y, y32), rel(dx, xin.grad))
    assert float((y.float() - y32).abs().max()) <= 2.0 ** -5 * float(y32.abs().max())
    assert float((dx.float() - xin.grad).abs().max()) <= 2.0 ** -5 * float(xin.grad.abs().max())
    if adapter_factor:
        for k in golden_params.ADAPTER_KEYS:
            assert grads[k] is not None, k
            assert rel(grads[k], p32[k].grad) <= 2e-2, (k, rel(grads[k], p32[k].grad))
        assert all(g is None for k, g in grads.items() if "adapter" not in k)
        return
    # Every parameter trainable: a bias / LayerNorm gradient is a sum over T = 8192 tokens of terms of either sign, so its
    # 16-bit error is set by the cancellation, not by the result's size.  Yardstick: the eager bf16 evaluation
    # of the same layer - what the reference computes - against the same fp32 oracle; the HIP layer may be 3x as far.
    p16 = {k: master[k].to(dtype).clone().requires_grad_(True) for k in golden_params.PARAM_ORDER}
    x16 = x.clone().requires_grad_(True)
    O.layer_forward(cfg, [p16], x16, O.causal_mask(s)).backward(dy)
    for k in golden_params.PARAM_ORDER:
        assert grads[k] is not None, k
        _check("grad " + k, grads[k], p16[k].grad, p32[k].grad, factor=3.0)


def test_full_shape_410m_layer_vs_fp32_oracle(cuda_device):
    """The assembled magma_pythia_410M layer AT THE BENCH SHAPE (s=2048, b=4, h=1024, 16 heads, GPT-J residual, tanh bias-GeLU,
    rotary 25 %, adapters r = h/8 on .attention and .mlp, LM frozen: configs/magma_pythia_410M.yml:5-33).  ~10 s of host time."""
    _full_shape_layer_vs_fp32_oracle(cuda_device, 1024, 16, 2048, 4, 0.25, True, True, 8, seed=410)


def test_full_shape_summit_70m_layer_vs_fp32_oracle(cuda_device):
    """summit-70m-openclipH at its own shape (configs/summit-70m-openclipH.yml:16-33: h=512, 8 heads, GPT-J residual, exact-erf
    GeLU - no bias_gelu_fusion -, rotary 25 %, adapters at the default downsample factor 4 => r = 128, LM frozen), s=2048, b=4:
    the h = 512 / r = 128 instantiation of the fused adapter kernels, the erf epilogues and the GPT-J tail together."""
    _full_shape_layer_vs_fp32_oracle(cuda_device, 512, 8, 2048, 4, 0.25, True, False, 4, seed=70)


def test_full_shape_19m_layer_vs_fp32_oracle(cuda_device):
    """19M.yml at its own shape (configs/19M.yml:6-17: h=512, 8 heads, sequential residual, exact-erf GeLU, full rotary, no
    adapters, every parameter trainable), s=2048, b=4: all twelve parameter gradients against the fp32 oracle."""
    _full_shape_layer_vs_fp32_oracle(cuda_device, 512, 8, 2048, 4, 1.0, False, False, 0, seed=19)


@pytest.mark.parametrize("dt", ["bf16"])
def test_stack_loss_curve_on_gpu(cuda_device, dt):
    """20 Adam steps of the 2-layer stack through the HIP layers reproduce the reference's loss curve (stated tolerance:
    0.05 absolute at every step; the reference's own bf16-vs-fp32 gap on this curve is 0.043)."""
    from multimodal_amd import functional as Fn
    t, meta = golden_io.load(f"stack_L2_{dt}")
    dtype = torch.bfloat16
    init = golden_params.stack_init(np.random.default_rng(meta["seed"]), meta["V"], meta["h"], meta["L"])
    args = LH.make_neox_args(meta["h"], meta["n"], dtype, meta["rotary_pct"], meta["gpt_j_residual"], meta["bias_gelu_fusion"],
                             softmax_fusion=True, L=meta["L"])
    layers = []
    for i in range(meta["L"]):
        master = {k: init[f"l{i}.{k}"] for k in golden_params.PARAM_ORDER}
        layers.append(LH.build_hip_layer(args, master, cuda_device))
    emb = init["emb"].to(dtype).to(cuda_device).requires_grad_(True)
    fw = init["fln.w"].to(dtype).to(cuda_device).requires_grad_(True)
    fb = init["fln.b"].to(dtype).to(cuda_device).requires_grad_(True)
    head = init["head"].to(dtype).to(cuda_device).requires_grad_(True)
    params = [emb, fw, fb, head] + [p for l in layers for p in l.parameters()]
    opt = torch.optim.Adam(params, lr=meta["lr"], betas=tuple(meta["betas"]), eps=meta["eps"])
    tokens, lm = t["tokens"].to(cuda_device), t["loss_mask"].to(cuda_device)
    mask = O.causal_mask(meta["s"]).to(cuda_device)
    losses = []
    for _ in range(meta["steps"]):
        opt.zero_grad()
        x = torch.nn.functional.embedding(tokens[:, :-1], emb).transpose(0, 1).contiguous()
        for l in layers:
            x = l(x, mask)
        x = x.transpose(0, 1).contiguous()
        logits = Fn.linear(Fn.layer_norm(x, fw, fb, 1e-5), head)
        loss = Fn.masked_cross_entropy(logits, tokens[:, 1:], lm)  # HIP cross entropy (fp32 math on the bf16 logits)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    ref = t["losses"].tolist()
    assert max(abs(a - b) for a, b in zip(losses, ref)) <= 0.05, (losses, ref)
    assert losses[-1] < 0.2 * losses[0]


@pytest.mark.gpu
def test_adapter_backward_after_fused_optimizer_step(cuda_device):
    """torch.optim.Adam(fused=True) updates parameters WITHOUT bumping their version counter: a transposed-weight cache
    keyed on the counter would feed the data-gradient GEMMs the pre-update weights.  After a large step the input gradient
    must follow the NEW weights."""
    from multimodal_amd import functional as Fn
    torch.manual_seed(3)
    M, h, r = 256, 512, 64
    mk = lambda *s, sc=1.0: torch.nn.Parameter((torch.randn(*s, device=cuda_device) * sc).bfloat16())
    w1, b1, w2, b2 = mk(r, h, sc=h ** -0.5), mk(r), mk(h, r, sc=r ** -0.5), mk(h)
    opt = torch.optim.Adam([w1, b1, w2, b2], lr=0.05, fused=True)
    g = torch.randn(M, h, device=cuda_device).bfloat16()

    def run():
        o = torch.randn(M, h, device=cuda_device, generator=torch.Generator(cuda_device).manual_seed(5)).bfloat16().requires_grad_(True)
        out = Fn.adapter(o, w1, b1, w2, b2)
        out.backward(g)
        return o, o.grad

    run()
    opt.step()          # weights move by ~lr per element
    opt.zero_grad()
    o, do = run()
    o32 = o.detach().float().requires_grad_(True)
    h1 = torch.relu((o32 @ w1.float().t() + b1.float()).bfloat16().float())
    ref_out = o32 + h1 @ w2.float().t() + b2.float()
    ref_out.backward(g.float())
    err = (do.float() - o32.grad).abs().max()
    assert float(err) <= 2.0 ** -6 * float(o32.grad.abs().max()), f"stale weights in the adapter backward? max err {float(err)}"
