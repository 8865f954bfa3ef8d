"""Attention dropout (megatron/model/transformer.py:301-302, 383-384: nn.Dropout on the softmax output inside the rng tracker's
fork) on the fused attention kernels: mm_attn_fwd_dropout / mm_attn_bwd_dropout / mm_attn_dropout_mask.

The reference's mask comes out of the device generator, so no fixture can pin individual mask bits: parity is (1) exact
against an fp32 restatement of `dropout(softmax(scores)) . V` that applies the mask the library itself publishes
(mm_attn_dropout_mask, the same function of (seed, offset, pos, batch*head, q, key) the kernels evaluate) - forward,
log-sum-exp and the gradient wrt q, k, v - and (2) distributional: keep rate, independence between heads, calls and steps."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {"bf16": torch.bfloat16, "f16": torch.float16}


def _state(dev, seed=1234, offset=0):
    return torch.tensor([seed, offset], dtype=torch.int64, device=dev)


def ref_attention_dropout(qkv, s, b, np_, hn, scale, mask, causal_inf, keep, p_eff):
    q, k, v = qkv.float().split(hn, dim=-1)
    scores = torch.einsum("sbnd,tbnd->bnst", q, k) * scale
    if causal_inf:
        scores = scores.masked_fill(torch.triu(torch.ones(s, s, dtype=torch.bool), 1), float("-inf"))
    elif mask is not None:
        scores = torch.where(mask.expand(b, np_, s, s), torch.full_like(scores, -10000.0), scores)
    lse = torch.logsumexp(scores, dim=-1)
    p = torch.softmax(scores, dim=-1) * keep.float() / (1.0 - p_eff)   # torch.nn.Dropout semantics with a given mask
    return torch.einsum("bnst,tbnd->sbnd", p, v).reshape(s, b, np_ * hn), lse


@pytest.mark.parametrize("dt,hn,s,mode,p", [("bf16", 64, 256, "causal", 0.1), ("bf16", 64, 200, "general", 0.25),
                                            ("f16", 64, 128, "causal", 0.5), ("bf16", 128, 192, "causal", 0.1),
                                            ("bf16", 128, 130, "general", 0.3), ("bf16", 64, 77, "causal", 0.2)])
def test_attention_dropout_matches_restatement_with_published_mask(cuda_device, dt, hn, s, mode, p):
    from multimodal_amd import ops
    dev = cuda_device
    torch.manual_seed(5)
    b, np_ = 2, 3
    dtype = DT[dt]
    qkv = torch.randn(s, b, np_, 3 * hn).to(dtype)
    dctx = torch.randn(s, b, np_ * hn).to(dtype)
    scale = 1.0 / math.sqrt(hn)
    mask = None
    if mode == "general":
        mask = torch.rand(b, 1, s, s) < 0.3
        mask[:, :, 5, :] = False
    drop = (p, _state(dev, 77, 1000), 3)
    keep = ops.attn_dropout_mask(s, b, np_, drop).cpu().bool()
    p_eff = round(p * 65536) / 65536.0
    x = qkv.clone().requires_grad_(True)
    ctx_ref, lse_ref = ref_attention_dropout(x, s, b, np_, hn, scale, mask, mode == "causal", keep, p_eff)
    ctx_ref.backward(dctx.float())

    g_qkv, g_mask = qkv.to(dev), (mask.to(dev) if mask is not None else None)
    ctx, lse = ops.attn_fwd(g_qkv, s, b, np_, hn, scale, g_mask, dropout=drop)
    dqkv = ops.attn_bwd(dctx.to(dev), ctx, lse, g_qkv, s, b, np_, hn, scale, g_mask, dropout=drop)
    torch.cuda.synchronize()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    rs = 1.0 / (1.0 - p_eff)
    err = (ctx.float().cpu() - ctx_ref.detach()).abs().max().item()
    assert err <= 4 * eps * rs * qkv.float().abs().max().item(), f"ctx max err {err}"
    assert (lse.cpu() - lse_ref.detach()).abs().max().item() <= 2e-3   # the log-sum-exp is over ALL keys, dropped or not
    gerr = (dqkv.float().cpu() - x.grad).abs().max().item()
    gscale = x.grad.abs().max().item()
    assert gerr <= 6 * eps * gscale + 1e-3, f"dqkv max err {gerr} (scale {gscale})"
    # and the dropped run really differs from the plain one
    ctx0, _ = ops.attn_fwd(g_qkv, s, b, np_, hn, scale, g_mask)
    assert (ctx0.float() - ctx.float()).abs().max().item() > 0.05


def test_attention_dropout_mask_statistics(cuda_device):
    from multimodal_amd import ops
    dev = cuda_device
    s, b, np_ = 512, 2, 4
    for p in (0.1, 0.5):
        st = _state(dev, 1234, 0)
        k0 = ops.attn_dropout_mask(s, b, np_, (p, st, 0)).float()
        n = k0[0, 0].numel()
        tol = 5.0 * math.sqrt(p * (1 - p) / n)
        for bi in range(b):
            for h in range(np_):
                assert abs(k0[bi, h].mean().item() - (1 - p)) < tol            # keep rate per (batch, head)
        assert abs(k0.mean(dim=(0, 1, 2)).mean().item() - (1 - p)) < tol       # no column bias
        assert (k0.mean(dim=3) - (1 - p)).abs().max().item() < 6.0 * math.sqrt(p * (1 - p) / s)   # nor row bias
        # independence: two heads / two positions in the step / two steps (base offset moved) / two seeds agree no more than chance
        k1 = ops.attn_dropout_mask(s, b, np_, (p, st, 1)).float()
        k2 = ops.attn_dropout_mask(s, b, np_, (p, _state(dev, 1234, 7), 0)).float()
        k3 = ops.attn_dropout_mask(s, b, np_, (p, _state(dev, 99, 0), 0)).float()
        chance = (1 - p) ** 2 + p ** 2
        for a, c in ((k0[0, 0], k0[0, 1]), (k0[0, 0], k0[1, 0]), (k0, k1), (k0, k2), (k0, k3)):
            agree = (a == c).float().mean().item()
            assert abs(agree - chance) < 0.01, (p, agree, chance)
        # neighbours along keys and along queries are uncorrelated too (the 16-bit halves of one hash; consecutive hashes)
        x = k0[0, 0] - (1 - p)
        for sh in ((0, 1), (1, 0), (0, 2)):
            y = torch.roll(x, shifts=sh, dims=(0, 1))
            corr = (x * y).mean().item() / (p * (1 - p))
            assert abs(corr) < 0.02, (sh, corr)
    # p == 0 keeps everything; same arguments, same mask
    assert bool(ops.attn_dropout_mask(64, 1, 1, (0.0, _state(dev), 0)).all())
    assert torch.equal(ops.attn_dropout_mask(64, 1, 2, (0.3, _state(dev), 5)), ops.attn_dropout_mask(64, 1, 2, (0.3, _state(dev), 5)))


def test_attention_dropout_in_layer_train_vs_eval(cuda_device):
    """ParallelTransformerLayer with attention_dropout = 0.2: eval equals the p = 0 layer; training differs, draws a new mask at
    the next call, replays the same mask when the tracker is rewound (what activation recompute does), and autograd runs
    through it (the backward regenerates the forward's mask from the saved stream position)."""
    import layer_harness as LH
    from multimodal_amd import mpu
    from multimodal_amd.model import ParallelTransformerLayer, gpt2_attention_mask_func
    dev = cuda_device
    mpu.initialize_model_parallel(1)
    mpu.model_parallel_cuda_manual_seed(4321)

    def build(p):
        args = LH.make_neox_args(256, 4, torch.bfloat16, 1.0, gpt_j=False, gelu_fusion=True, s=128)
        args.attention_dropout = p
        init = lambda w: torch.nn.init.normal_(w, std=0.05)
        return ParallelTransformerLayer(neox_args=args, attention_mask_func=gpt2_attention_mask_func, init_method=init,
                                        output_layer_init_method=init, layer_number=0, rotary=True).to(dev)

    torch.manual_seed(3)
    l0, l1 = build(0.0), build(0.2)
    l1.load_state_dict(l0.state_dict())
    x = torch.randn(128, 2, 256, device=dev, dtype=torch.bfloat16)
    mask = torch.triu(torch.ones(128, 128, dtype=torch.bool, device=dev), 1).view(1, 1, 128, 128)
    l0.eval(); l1.eval()
    assert torch.equal(l0(x, mask), l1(x, mask))
    l1.train(); l0.train()
    states = mpu.get_cuda_rng_tracker().get_states()
    y0 = l0(x, mask)
    y1 = l1(x, mask)
    y2 = l1(x, mask)                      # next position of the stream: another mask
    assert (y1.float() - y0.float()).abs().max().item() > 1e-3
    assert (y1.float() - y2.float()).abs().max().item() > 1e-3
    mpu.get_cuda_rng_tracker().set_states(states)
    assert torch.equal(y1, l1(x, mask))   # rewound: the same mask
    xg = x.clone().requires_grad_(True)
    l1(xg, mask).float().square().mean().backward()
    assert torch.isfinite(xg.grad.float()).all() and xg.grad.float().abs().max().item() > 0
