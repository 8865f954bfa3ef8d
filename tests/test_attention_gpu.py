"""GPU parity of the fused attention kernels (C ABI mm_attn_fwd / mm_attn_bwd) against an fp32 CPU restatement of
ParallelSelfAttention.attention (megatron/model/transformer.py:315-416) on the same 16-bit inputs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {"bf16": torch.bfloat16, "f16": torch.float16}


def ref_attention(qkv, s, b, np_, hn, scale, mask, causal_inf):
    """fp32 reference on the rounded inputs.  qkv: [s,b,np,3hn].  Returns ctx [s,b,np*hn], lse [b,np,s]."""
    q, k, v = qkv.float().split(hn, dim=-1)
    scores = torch.einsum("sbnd,tbnd->bnst", q, k) * scale
    if causal_inf:
        dead = torch.triu(torch.ones(s, s, dtype=torch.bool), 1)
        scores = scores.masked_fill(dead, float("-inf"))
    else:
        scores = torch.where(mask.expand(b, np_, s, s), torch.full_like(scores, -10000.0), scores)
    lse = torch.logsumexp(scores, dim=-1)
    p = torch.softmax(scores, dim=-1)
    ctx = torch.einsum("bnst,tbnd->sbnd", p, v).reshape(s, b, np_ * hn)
    return ctx, lse


def run_case(dev, dt, s, b, np_, hn, mode, seed=0):
    from multimodal_amd import ops
    torch.manual_seed(seed)
    dtype = DT[dt]
    qkv = torch.randn(s, b, np_, 3 * hn).to(dtype)
    dctx = torch.randn(s, b, np_ * hn).to(dtype)
    scale = 1.0 / math.sqrt(hn)
    mask = None
    if mode == "causal_bytes":
        mask = torch.triu(torch.ones(s, s, dtype=torch.bool), 1).view(1, 1, s, s)
    elif mode == "general":
        mask = torch.rand(b, 1, s, s) < 0.4
        mask[:, :, 3, :] = True      # a fully masked row -> uniform attention over all keys (reference -10000 semantics)
        mask[:, :, 5, :] = False
    x = qkv.clone().requires_grad_(True)
    ctx_ref, lse_ref = ref_attention(x, s, b, np_, hn, scale, mask, causal_inf=(mode == "causal"))
    ctx_ref.backward(dctx.float())
    dqkv_ref = x.grad

    g_qkv = qkv.to(dev)
    g_mask = mask.to(dev) if mask is not None else None
    ctx, lse = ops.attn_fwd(g_qkv, s, b, np_, hn, scale, g_mask)
    dqkv = ops.attn_bwd(dctx.to(dev), ctx, lse, g_qkv, s, b, np_, hn, scale, g_mask)
    torch.cuda.synchronize()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    # forward: P is rounded to 16 bit before P.V -> error ~ eps * sum|p v| <= eps * max|v|
    err = (ctx.float().cpu() - ctx_ref.detach()).abs().max().item()
    assert err <= 4 * eps * qkv.float().abs().max().item(), f"ctx max err {err}"
    lerr = (lse.cpu() - lse_ref.detach()).abs().max().item()
    assert lerr <= 2e-3, f"lse max err {lerr}"
    gerr = (dqkv.float().cpu() - dqkv_ref).abs().max().item()
    gscale = dqkv_ref.abs().max().item()
    assert gerr <= 6 * eps * gscale + 1e-3, f"dqkv max err {gerr} (scale {gscale})"
    # relative L2 per slice is a much sharper check for systematic (permutation / sign) errors
    for name, sl in (("dq", slice(0, hn)), ("dk", slice(hn, 2 * hn)), ("dv", slice(2 * hn, 3 * hn))):
        a, r = dqkv.float().cpu()[..., sl], dqkv_ref[..., sl]
        rel = ((a - r).norm() / r.norm().clamp_min(1e-6)).item()
        assert rel <= 3 * eps, f"{name} relative L2 error {rel}"
    rel = ((ctx.float().cpu() - ctx_ref.detach()).norm() / ctx_ref.detach().norm()).item()
    assert rel <= 2 * eps, f"ctx relative L2 error {rel}"
    return ctx, dqkv


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("hn", [64, 128])
@pytest.mark.parametrize("s", [64, 128, 200, 512])
@pytest.mark.parametrize("mode", ["causal", "causal_bytes", "general"])
def test_attention_fwd_bwd(cuda_device, dt, hn, s, mode):
    run_case(cuda_device, dt, s, 2, 3, hn, mode, seed=s + hn)


def test_attention_causal_equals_byte_mask(cuda_device):
    """Implicit causal (-inf) and the same mask given as bytes (-10000) agree: exp(-10000 - m) == 0 in fp32."""
    from multimodal_amd import ops
    torch.manual_seed(1)
    s, b, np_, hn = 256, 1, 4, 64
    qkv = torch.randn(s, b, np_, 3 * hn).bfloat16().to(cuda_device)
    mask = torch.triu(torch.ones(s, s, dtype=torch.bool), 1).view(1, 1, s, s).to(cuda_device)
    c1, l1 = ops.attn_fwd(qkv, s, b, np_, hn, 0.125, None)
    c2, l2 = ops.attn_fwd(qkv, s, b, np_, hn, 0.125, mask)
    assert torch.equal(c1, c2) and torch.allclose(l1, l2, atol=1e-6)


def test_attention_410m_shape_properties(cuda_device):
    """BASELINE size (s=2048, b=4, np=16, hn=64): too big for the CPU reference in seconds, so check size-independent
    properties: (1) V = const -> ctx = const; (2) linearity in V; (3) row 0 attends to key 0 only; (4) sum_d dV check."""
    from multimodal_amd import ops
    torch.manual_seed(2)
    s, b, np_, hn = 2048, 4, 16, 64
    qkv = torch.randn(s, b, np_, 3 * hn, device=cuda_device).bfloat16()
    ctx, lse = ops.attn_fwd(qkv, s, b, np_, hn, 0.125, None)
    v = qkv[..., 2 * hn:]
    assert torch.equal(ctx[0].view(b, np_, hn), v[0]), "query 0 sees only key 0"
    q2 = qkv.clone()
    q2[..., 2 * hn:] = 1.0
    c_const, _ = ops.attn_fwd(q2, s, b, np_, hn, 0.125, None)
    assert (c_const.float() - 1.0).abs().max() <= 2 ** -7
    q3 = qkv.clone()
    q3[..., 2 * hn:] = v * 2
    c_dbl, l_dbl = ops.attn_fwd(q3, s, b, np_, hn, 0.125, None)
    assert torch.equal(c_dbl.float(), ctx.float() * 2) and torch.equal(l_dbl, lse)
    # backward: with dctx = const c, dV[t] = c * sum_q P[q,t]; summed over keys per (b, head) that is c * s
    dctx = torch.ones(s, b, np_ * hn, device=cuda_device).bfloat16()
    dqkv = ops.attn_bwd(dctx, ctx, lse, qkv, s, b, np_, hn, 0.125, None)
    dv = dqkv[..., 2 * hn:].float()
    tot = dv.sum(dim=0)  # [b, np, hn]
    assert (tot / s - 1.0).abs().max() <= 2e-2
    assert torch.isfinite(dqkv.float()).all()


@pytest.mark.parametrize("s,b,np_,hn,mode", [(1024, 1, 2, 128, "causal"),      # summit-1-3B / 6-7B head size, 16 key tiles
                                              (2048, 1, 1, 128, "causal"),      # ... and the full 32 key tiles of s = 2048
                                              (2048, 1, 1, 128, "general"),     # byte-mask path at the full sequence length
                                              (2048, 1, 2, 64, "general"),
                                              (1000, 2, 1, 128, "causal_bytes")])  # ragged last tile, hn = 128
def test_attention_long_sequence_vs_fp32(cuda_device, s, b, np_, hn, mode):
    """Head size 128 (configs/summit-1-3B.yml:20-26, 6-7B.yml:9-15) and the byte-mask path at the sequence lengths the configs
    run: the causal tile skip, the heavy-first grid and the 1-workgroup-per-CU hn = 128 instantiation over 16-32 key tiles,
    against the fp32 CPU restatement (a few seconds of host time each)."""
    run_case(cuda_device, "bf16", s, b, np_, hn, mode, seed=7 * s + hn)


def test_attention_hn128_s2048_shape_properties(cuda_device):
    """The summit-1-3B attention shape per TP rank (s=2048, b=4, np=9, hn=128): size-independent properties, as in
    test_attention_410m_shape_properties, plus the same checks through the byte-mask kernels with the causal mask given as
    bytes (must equal the implicit-causal result bit for bit, forward and backward)."""
    from multimodal_amd import ops
    torch.manual_seed(5)
    s, b, np_, hn = 2048, 4, 9, 128
    scale = 1.0 / math.sqrt(hn)
    qkv = torch.randn(s, b, np_, 3 * hn, device=cuda_device).bfloat16()
    ctx, lse = ops.attn_fwd(qkv, s, b, np_, hn, scale, None)
    v = qkv[..., 2 * hn:]
    assert torch.equal(ctx[0].view(b, np_, hn), v[0]), "query 0 sees only key 0"
    q2 = qkv.clone()
    q2[..., 2 * hn:] = 1.0
    c_const, _ = ops.attn_fwd(q2, s, b, np_, hn, scale, None)
    assert (c_const.float() - 1.0).abs().max() <= 2 ** -7
    q3 = qkv.clone()
    q3[..., 2 * hn:] = v * 2
    c_dbl, l_dbl = ops.attn_fwd(q3, s, b, np_, hn, scale, None)
    assert torch.equal(c_dbl.float(), ctx.float() * 2) and torch.equal(l_dbl, lse)
    dctx = torch.ones(s, b, np_ * hn, device=cuda_device).bfloat16()
    dqkv = ops.attn_bwd(dctx, ctx, lse, qkv, s, b, np_, hn, scale, None)
    tot = dqkv[..., 2 * hn:].float().sum(dim=0)
    assert (tot / s - 1.0).abs().max() <= 2e-2
    assert torch.isfinite(dqkv.float()).all()
    mask = torch.triu(torch.ones(s, s, dtype=torch.bool, device=cuda_device), 1).view(1, 1, s, s)
    c_m, l_m = ops.attn_fwd(qkv, s, b, np_, hn, scale, mask)
    assert torch.equal(c_m, ctx) and torch.allclose(l_m, lse, atol=1e-6)
    rnd = torch.randn(s, b, np_ * hn, device=cuda_device).bfloat16()
    g0 = ops.attn_bwd(rnd, ctx, lse, qkv, s, b, np_, hn, scale, None)
    g1 = ops.attn_bwd(rnd, ctx, lse, qkv, s, b, np_, hn, scale, mask)
    rel = float((g0.float() - g1.float()).norm() / g0.float().norm())
    assert rel <= 2 ** -9, rel  # same arithmetic up to the lse the two forwards returned (equal to 1e-6)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("s,b,np_,hn,d,offset,mode", [(200, 2, 2, 64, 16, 0, "causal"), (128, 1, 3, 64, 64, 3, "causal"),
                                                      (96, 2, 2, 64, 32, 0, "general"), (160, 2, 2, 128, 128, 0, "causal"),
                                                      (130, 1, 2, 128, 32, 1, "causal"), (64, 2, 1, 128, 64, 0, "general")])
def test_attention_bwd_fused_rotary_equals_two_passes(cuda_device, dt, s, b, np_, hn, d, offset, mode):
    """mm_attn_bwd_rotary (rotary backward inside the dq / dk stores) is bit-identical to mm_attn_bwd followed by
    mm_rotary_qk_inplace(inverse)."""
    from multimodal_amd import ops
    dtype = DT[dt]
    torch.manual_seed(s + d)
    qkv = torch.randn(s, b, np_, 3 * hn).to(dtype).to(cuda_device)
    dctx = torch.randn(s, b, np_ * hn).to(dtype).to(cuda_device)
    mask = (torch.rand(b, 1, s, s) < 0.3).to(cuda_device) if mode == "general" else None
    pos = torch.arange(offset + s, dtype=torch.float32)
    inv = 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
    emb = torch.cat((pos[:, None] * inv[None, :],) * 2, dim=-1)
    cos, sin = emb.cos().to(dtype).to(cuda_device), emb.sin().to(dtype).to(cuda_device)
    scale = 1.0 / math.sqrt(hn)
    ctx, lse = ops.attn_fwd(qkv, s, b, np_, hn, scale, mask)
    fused = ops.attn_bwd(dctx, ctx, lse, qkv, s, b, np_, hn, scale, mask, rotary=(cos, sin, d, offset))
    two = ops.attn_bwd(dctx, ctx, lse, qkv, s, b, np_, hn, scale, mask)
    plain = two.clone()
    ops.rotary_qk_inplace(two, cos, sin, s, b, np_, hn, d, offset, inverse=True)
    torch.cuda.synchronize()
    assert torch.equal(fused.view(torch.int16), two.view(torch.int16))
    assert not torch.equal(fused[1:, ..., :d], plain[1:, ..., :d])
