"""GPU parity of the kernels either side of the layers: the ImagePrefix tail + embedding concat against
fixtures made by the REFERENCE's own ImagePrefix / EmbeddingPipe (tests/golden/make_golden.py::run_embed_case), the final
LayerNorm that reads [s, b, h] by stride, and the fused final-norm + LM-head + masked-mean-loss node against the unfused chain
and the CPU oracle."""
import glob
import os

import pytest
import torch

import golden_io
from oracle import neox_layer_oracle as O

pytestmark = pytest.mark.gpu

EMBED = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(golden_io.HERE, "op_embed_prefix_*_bf16.npz")))


@pytest.mark.parametrize("name", EMBED)
def test_image_prefix_tail_and_embedding_match_reference_fixture(cuda_device, name):
    """image_prefix.py:217-252 (encoder = identity) + word_embeddings.py:182-205 + gpt2_model.py:73-77, eval mode: the HIP
    prefix kernel + embed_prefix_k against what the reference computed in bf16 on CPU.  Tolerance: the prefix is a LayerNorm
    output of O(1) values from a K-deep bf16 dot product: 2 bf16 ulps of the largest value (the reference's own bf16 run
    differs from its fp32 run by about one); the gathered embedding rows must be bit-exact."""
    from multimodal_amd import functional as Fn
    t, meta = golden_io.load(name)
    dev = cuda_device
    leaf = {k: t[k].to(dev).requires_grad_(True) for k in ("enc", "proj_w", "proj_b", "ln_w", "ln_b")}
    prefix = Fn.image_prefix_tail(leaf["enc"], leaf["proj_w"], leaf["proj_b"], leaf["ln_w"], leaf["ln_b"], meta["eps"])
    assert prefix.shape == (meta["b"], meta["P"], meta["h"])
    ref = t["prefix"].float()
    tol = 2.0 ** -7 * float(ref.abs().max())
    assert float((prefix.float().cpu() - ref).abs().max()) <= tol
    table = t["table"].to(dev)
    x0 = Fn.embed_prefix(prefix, table, t["tokens"].to(dev))
    P = meta["P"]
    assert x0.shape == (meta["s"], meta["b"], meta["h"])
    assert torch.equal(x0[P:].cpu(), t["x0"][P:]), "gathered word embeddings are copies: bit-exact"
    assert float((x0[:P].float().cpu() - t["x0"][:P].float()).abs().max()) <= tol
    x0.backward(t["g"].to(dev))
    # gradients: fp32 oracle on the same bf16 inputs as the yardstick; the reference's bf16 gradients bound the allowed error
    l32 = {k: t[k].float().requires_grad_(True) for k in leaf}
    p32 = O.image_prefix_tail(l32["enc"], l32["proj_w"], l32["proj_b"], l32["ln_w"], l32["ln_b"], meta["eps"])
    O.embed_with_prefix(p32, t["table"].float(), t["tokens"]).backward(t["g"].float())
    for k, gk in (("enc", "d_enc"), ("proj_w", "d_proj_w"), ("proj_b", "d_proj_b"), ("ln_w", "d_ln_w"), ("ln_b", "d_ln_b")):
        g32 = l32[k].grad
        e_ref = float((t[gk].float() - g32).abs().max())
        e_hip = float((leaf[k].grad.float().cpu() - g32).abs().max())
        assert e_hip <= 3 * e_ref + 2.0 ** -7 * float(g32.abs().max()), (k, e_hip, e_ref)


@pytest.mark.parametrize("rows,enc,h", [(4, 1024, 1024), (3, 64, 256), (16, 2048, 512), (1, 64, 32)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_prefix_tail_kernel_equals_three_kernel_chain(cuda_device, rows, enc, h, p):
    """mm_prefix_tail_fwd against mm_gemm_nt + mm_bias_dropout_add_devrng + mm_layernorm_fwd (the chain it replaces): same
    dropout mask bit for bit (same Philox stream), outputs within one rounding of the projection (the dot products are summed
    in a different order), and the backward through both paths agrees likewise."""
    from multimodal_amd import functional as Fn, ops
    torch.manual_seed(rows + enc)
    dt = torch.bfloat16
    x = torch.randn(rows, enc, device=cuda_device).to(dt)
    w = (torch.randn(h, enc, device=cuda_device) * enc ** -0.5).to(dt)
    b = (torch.randn(h, device=cuda_device) * 0.1).to(dt)
    lw = (1 + 0.1 * torch.randn(h, device=cuda_device)).to(dt)
    lb = (0.1 * torch.randn(h, device=cuda_device)).to(dt)
    state = torch.tensor([1234, 77], dtype=torch.int64, device=cuda_device)
    assert ops.prefix_tail_supported(rows, enc, h)
    out, ln_in, keep, mean, rstd = ops.prefix_tail_fwd(x, w, b, lw, lb, 1e-5, p, True, state, 5)
    y = ops.gemm_nt(x, w, b)
    if p > 0:
        y2, keep2 = ops.bias_dropout_add(y, None, None, p, True, 0, 5, want_mask=True, rng_state=state)
        assert torch.equal(keep, keep2), "same Philox stream, same mask"
        assert 0.8 <= float(keep.float().mean()) <= 0.97
    else:
        y2 = y
        assert keep is None
    ref, _, mean2, rstd2 = ops.layernorm_fwd(y2, lw, lb, 1e-5)
    ulp = 2.0 ** -7
    assert float((ln_in.float() - y2.float()).abs().max()) <= ulp * float(y2.float().abs().max())
    assert float((out.float() - ref.float()).abs().max()) <= 2 * ulp * float(ref.float().abs().max())
    # repeated calls reuse nothing stale (the ticket is re-zeroed per call) and are deterministic
    out_b = ops.prefix_tail_fwd(x, w, b, lw, lb, 1e-5, p, True, state, 5)[0]
    assert torch.equal(out, out_b)
    # autograd through the fused node vs the chain
    g = torch.randn(rows, h, device=cuda_device).to(dt)

    def grads(fn):
        leaves = [t_.clone().requires_grad_(True) for t_ in (x, w, b, lw, lb)]
        fn(*leaves).backward(g)
        return [l.grad for l in leaves]

    fused = grads(lambda x_, w_, b_, lw_, lb_: Fn.image_prefix_tail(x_.view(rows, 1, enc), w_, b_, lw_, lb_, 1e-5, p, p > 0, state, 5).view(rows, h))

    def chain(x_, w_, b_, lw_, lb_):
        yy = Fn.linear(x_, w_, b_)
        if p > 0:
            yy = Fn.bias_dropout_add(yy, None, None, p, True, 0, 5, rng_state=state)
        return Fn.layer_norm(yy, lw_, lb_, 1e-5)

    for a, r in zip(fused, grads(chain)):
        scale = float(r.float().abs().max())
        assert float((a.float() - r.float()).abs().max()) <= 2.0 ** -5 * scale + 1e-6, (a.shape, scale)


def test_prefix_tail_large_row_count_takes_the_unfused_path(cuda_device):
    """An un-pooled encoder (257 prefix tokens per sample, image_prefix.py:155-159) is beyond the fused kernel's 16 rows."""
    from multimodal_amd import functional as Fn, ops
    assert not ops.prefix_tail_supported(2 * 257, 1024, 1024)
    torch.manual_seed(0)
    dt = torch.bfloat16
    enc = torch.randn(2, 257, 128, device=cuda_device).to(dt)
    w = (torch.randn(256, 128, device=cuda_device) * 0.09).to(dt)
    b = torch.zeros(256, device=cuda_device, dtype=dt)
    lw, lb = torch.ones(256, device=cuda_device, dtype=dt), torch.zeros(256, device=cuda_device, dtype=dt)
    out = Fn.image_prefix_tail(enc, w, b, lw, lb, 1e-5)
    ref = O.image_prefix_tail(enc.float().cpu(), w.float().cpu(), b.float().cpu(), lw.float().cpu(), lb.float().cpu())
    assert out.shape == (2, 257, 256)
    assert float((out.float().cpu() - ref).abs().max()) <= 2.0 ** -6 * float(ref.abs().max())


@pytest.mark.parametrize("s,b,h", [(64, 2, 256), (200, 3, 1024), (2048, 4, 1024)])
def test_final_layernorm_strided_equals_transpose_then_layernorm(cuda_device, s, b, h):
    """mm_layernorm_fwd_sb / _bwd_sb == transpose(0, 1).contiguous() (gpt2_model.py:80-85) + the plain LayerNorm kernels,
    bit for bit, forward, dx and the parameter gradients; dy_scale multiplies the gradient."""
    from multimodal_amd import ops
    torch.manual_seed(s)
    dt = torch.bfloat16
    x = torch.randn(s, b, h, device=cuda_device).to(dt)
    w = (1 + 0.1 * torch.randn(h, device=cuda_device)).to(dt)
    bb = (0.1 * torch.randn(h, device=cuda_device)).to(dt)
    dy = torch.randn(b, s, h, device=cuda_device).to(dt)
    y, mean, rstd = ops.layernorm_fwd_sb(x, w, bb, 1e-5)
    xt = x.transpose(0, 1).contiguous()
    y_ref, _, mean_ref, rstd_ref = ops.layernorm_fwd(xt, w, bb, 1e-5)
    assert torch.equal(y, y_ref) and torch.equal(mean, mean_ref) and torch.equal(rstd, rstd_ref)
    dx, grads = ops.layernorm_bwd_sb(dy, x, w, mean, rstd)
    dx_ref, grads_ref = ops.layernorm_bwd(dy, xt, w, mean_ref, rstd_ref)
    assert torch.equal(dx, dx_ref.transpose(0, 1).contiguous())
    assert torch.equal(grads[0], grads_ref[0]) and torch.equal(grads[1], grads_ref[1])
    two = torch.full((1,), 2.0, device=cuda_device)
    dx2, grads2 = ops.layernorm_bwd_sb(dy, x, w, mean, rstd, dy_scale=two)
    dx2_ref, grads2_ref = ops.layernorm_bwd((dy.float() * 2).to(dt), xt, w, mean_ref, rstd_ref)
    assert torch.equal(dx2, dx2_ref.transpose(0, 1).contiguous())   # a power of two scales exactly
    assert torch.equal(grads2[0], grads2_ref[0]) and torch.equal(grads2[1], grads2_ref[1])


@pytest.mark.parametrize("trainable", [False, True])
def test_fused_final_norm_head_loss_equals_unfused_chain_and_oracle(cuda_device, trainable):
    """FinalNormHeadLossFn (one node: strided final LayerNorm, LM head, masked-mean loss, HIP masked-mean helpers) against
    (a) transpose + LayerNormFn + LinearFn + masked_cross_entropy - the same kernels node by node - and (b) the fp32 CPU
    oracle (transformer.py:816-839, gpt2_model.py:52-70); upstream gradient 0.5 (a power of two: exact either way)."""
    from multimodal_amd import functional as Fn
    torch.manual_seed(3)
    s, b, h, V = 96, 2, 256, 1024
    dt = torch.bfloat16
    x = torch.randn(s, b, h, device=cuda_device).to(dt)
    lw = (1 + 0.1 * torch.randn(h, device=cuda_device)).to(dt)
    lb = (0.1 * torch.randn(h, device=cuda_device)).to(dt)
    W = (torch.randn(V, h, device=cuda_device) * 0.05).to(dt)
    labels = torch.randint(0, V, (b, s), device=cuda_device)
    lm = torch.ones(b, s, device=cuda_device)
    lm[:, -7:] = 0

    def run(fn):
        leaves = [x.clone().requires_grad_(True), lw.clone().requires_grad_(trainable), lb.clone().requires_grad_(trainable),
                  W.clone().requires_grad_(trainable)]
        loss = fn(*leaves)
        (loss * 0.5).backward()
        return loss.detach(), [l.grad for l in leaves]

    l_f, g_f = run(lambda x_, lw_, lb_, W_: Fn.final_norm_head_loss(x_, lw_, lb_, 1e-5, W_, labels, lm))

    def chain(x_, lw_, lb_, W_):
        xn = Fn.layer_norm(x_.transpose(0, 1).contiguous(), lw_, lb_, 1e-5)
        return Fn.masked_cross_entropy(Fn.linear(xn, W_), labels, lm)

    l_c, g_c = run(chain)
    assert abs(float(l_f) - float(l_c)) <= 1e-5
    for a, r in zip(g_f, g_c):
        if r is None:
            assert a is None
            continue
        # the chain rounds the [T, V] gradient with the upstream scale folded in, the fused node scales afterwards: 1 ulp
        assert float((a.float() - r.float()).abs().max()) <= 2.0 ** -7 * float(r.float().abs().max()) + 1e-8
    x32 = x.float().cpu().requires_grad_(True)
    xn32 = O.layer_norm(x32.transpose(0, 1), lw.float().cpu(), lb.float().cpu())
    logits32 = (xn32.to(dt).float() @ W.float().cpu().t()).to(dt)  # the reference's bf16 logits, then .float() for the loss
    l_o = O.cross_entropy_loss(logits32, labels.cpu(), lm.cpu())
    assert abs(float(l_f) - float(l_o)) <= 2e-3


def test_prefix_tail_replays_correctly_in_a_hip_graph(cuda_device):
    """The ticket that elects the normalising workgroup must be re-zeroed by a GRAPH NODE: a replayed launch that finds the
    ticket spent returns the previous batch's rows (seen once with hipMemsetAsync, which ran at capture time)."""
    from multimodal_amd import ops
    torch.manual_seed(0)
    dt = torch.bfloat16
    rows, enc, h = 2, 64, 256
    w = (torch.randn(h, enc, device=cuda_device) * 0.1).to(dt)
    b = torch.zeros(h, device=cuda_device, dtype=dt)
    lw, lb = torch.ones(h, device=cuda_device, dtype=dt), torch.zeros(h, device=cuda_device, dtype=dt)
    xs = [torch.randn(rows, enc, device=cuda_device).to(dt) for _ in range(3)]
    eager = [ops.prefix_tail_fwd(x, w, b, lw, lb, 1e-5)[0].clone() for x in xs]
    static = xs[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.prefix_tail_fwd(static, w, b, lw, lb, 1e-5)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.prefix_tail_fwd(static, w, b, lw, lb, 1e-5)[0]
    for i, x in enumerate(xs):
        static.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager[i]), f"replay {i}"
