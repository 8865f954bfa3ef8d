"""Pins the CPU oracle (oracle/neox_layer_oracle.py) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  Runs on CPU (-m "not gpu")."""
import glob
import os

import numpy as np
import pytest
import torch

import golden_io
import golden_params
from oracle import neox_layer_oracle as O

DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def eq(a, b):
    return torch.equal(a, b)


@pytest.mark.parametrize("h", [512, 1024])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_layernorm_matches_reference(h, dt):
    t, meta = golden_io.load(f"op_layernorm_h{h}_{dt}")
    x = t["x"].clone().requires_grad_(True)
    w = t["w"].clone().requires_grad_(True)
    b = t["b"].clone().requires_grad_(True)
    y = O.layer_norm(x, w, b, meta["eps"])
    y.backward(t["dy"])
    assert eq(y.detach(), t["y"]) and eq(x.grad, t["dx"]) and eq(w.grad, t["dw"]) and eq(b.grad, t["db"])


@pytest.mark.parametrize("d", [16, 64, 128])
@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_rotary_tables_bit_exact(d, dt):
    t, meta = golden_io.load(f"op_rotary_table_d{d}_{dt}")
    cos, sin = O.rotary_tables(meta["seq_len"], d, meta["base"], DT[dt])
    assert eq(cos, t["cos"]) and eq(sin, t["sin"])
    if dt == "bf16":  # the reference's quirk: bf16 positions above 256 are rounded
        cos32, _ = O.rotary_tables(meta["seq_len"], d, meta["base"], torch.float32)
        assert not torch.allclose(cos.float(), cos32, atol=1e-2)


@pytest.mark.parametrize("d", [16, 64])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_rotary_apply_bit_exact(d, dt):
    t, _ = golden_io.load(f"op_rotary_apply_d{d}_{dt}")
    q = t["q"].clone().requires_grad_(True)
    k = t["k"].clone().requires_grad_(True)
    hn = q.size(-1)
    qr, kr = O.apply_rotary(q[..., :d], k[..., :d], t["cos"], t["sin"])
    if d < hn:
        qr = torch.cat((qr, q[..., d:]), -1)
        kr = torch.cat((kr, k[..., d:]), -1)
    torch.autograd.backward([qr, kr], [t["dq_out"], t["dk_out"]])
    assert eq(qr.detach(), t["q_out"]) and eq(kr.detach(), t["k_out"])
    assert eq(q.grad, t["dq"]) and eq(k.grad, t["dk"])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_bias_gelu_matches_reference(dt):
    t, _ = golden_io.load(f"op_bias_gelu_{dt}")
    y = t["y"].clone().requires_grad_(True)
    bias = t["bias"].clone().requires_grad_(True)
    out = O.bias_gelu_impl(y, bias)
    out.backward(t["g"])
    # the reference op is TorchScript: allow 1 ulp of the storage dtype for fusion-order differences
    tol = dict(rtol=0, atol=0) if dt == "f32" else dict(rtol=2 ** -7, atol=1e-6)
    if dt == "f32":
        torch.testing.assert_close(out.detach(), t["out_tanh"], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(y.grad, t["dy_tanh"], rtol=1e-5, atol=1e-6)
    else:
        torch.testing.assert_close(out.detach().float(), t["out_tanh"].float(), **tol)
        torch.testing.assert_close(y.grad.float(), t["dy_tanh"].float(), rtol=2 ** -6, atol=1e-5)
    y2 = t["y"].clone().requires_grad_(True)
    o2 = O.gelu_erf_bias(y2, t["bias"])
    o2.backward(t["g"])
    assert eq(o2.detach(), t["out_erf"]) and eq(y2.grad, t["dy_erf"])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_bias_dropout_add_bit_exact(dt):
    t, _ = golden_io.load(f"op_bias_dropout_add_{dt}")
    x, bias, res = t["x"], t["bias"], t["residual"]
    assert eq(O.bias_dropout_add(x, bias.expand_as(x), res, 0.0, True), t["out_res"])
    assert eq(O.bias_dropout_add(x, bias.expand_as(x), None, 0.0, True), t["out_nores"])
    assert eq(O.bias_dropout_add(x, bias.expand_as(x), res, 0.1, False), t["out_eval"])


SOFTMAX_FILES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(golden_io.HERE, "op_softmax_eager_*.npz")))


@pytest.mark.parametrize("name", SOFTMAX_FILES)
def test_softmax_eager_bit_exact_and_fused_restatement_close(name):
    t, meta = golden_io.load(name)
    x = t["x"]
    sk = x.size(-1)
    xin = x.clone().requires_grad_(True)
    p = O.softmax_eager(xin * 1.0, O.causal_mask(sk), meta["scale"], meta["softmax_in_fp32"])
    p.backward(t["dy"])
    assert eq(p.detach(), t["probs_causal"]) and eq(xin.grad, t["dx_causal"])
    p2 = O.softmax_eager(x.clone(), t["mask_general"], meta["scale"], meta["softmax_in_fp32"])
    assert eq(p2, t["probs_general"])
    if x.dtype != torch.float32:
        # fused-kernel restatements vs the reference's eager path: the reference's own bar is mean|diff| <= 1e-3
        scale = meta["scale"] if meta["scale"] is not None else 1.0
        b, np_ = x.shape[:2]
        pf = O.softmax_causal_fused_fwd(x.view(-1, sk, sk), scale).view(b, np_, sk, sk)
        assert (pf.float() - t["probs_causal"].float()).abs().mean() <= 1e-3
        assert torch.all(pf[..., torch.triu(torch.ones(sk, sk, dtype=torch.bool), 1)] == 0)
        pm = O.softmax_masked_fused_fwd(x, t["mask_general"], scale)
        assert (pm.float() - t["probs_general"].float()).abs().mean() <= 1e-3
        gb = O.softmax_fused_bwd(t["dy"].view(-1, sk, sk), pf.view(-1, sk, sk), scale, causal=True).view_as(x)
        assert (gb.float() - t["dx_causal"].float()).abs().mean() <= 2e-3


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_linears_tp1_bit_exact(dt):
    t, meta = golden_io.load(f"op_linear_tp1_{dt}")
    x = t["x"].clone().requires_grad_(True)
    cw = t["col_w"].clone().requires_grad_(True)
    cb = t["col_b"].clone().requires_grad_(True)
    rw = t["row_w"].clone().requires_grad_(True)
    yc, bc = O.column_parallel_linear(x, cw, cb, skip_bias_add=False)
    (yr,), br = O.row_parallel_linear([yc], [rw], t["row_b"], skip_bias_add=True)
    yr.backward(t["dy"])
    assert bc is None and eq(br, t["row_b"])
    assert eq(yc.detach(), t["y_col"]) and eq(yr.detach(), t["y_row"])
    assert eq(x.grad, t["dx"]) and eq(cw.grad, t["d_col_w"]) and eq(cb.grad, t["d_col_b"]) and eq(rw.grad, t["d_row_w"])


def _layer_files(tp):
    pat = os.path.join(golden_io.HERE, f"layer_*_tp{tp}{'_r0' if tp > 1 else ''}.npz")
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(pat))


def _run_oracle_layer(meta, x, dy, tp):
    dt = DT[meta["dtype"]]
    master = golden_params.master_params(np.random.default_rng(meta["seed"]), meta["h"])
    # replicated parameters (LayerNorm, Row-linear biases) are ONE tensor shared by every simulated rank: their
    # gradient is then the all-reduced one each reference rank sees (mpu/mappings.py:121-123)
    shared = {k: master[k].to(dt).requires_grad_(True) for k in golden_params.PARAM_ORDER
              if golden_params.shard(k, master[k], 0, 2) is master[k]}
    shards = []
    for r in range(tp):
        shards.append({k: shared[k] if k in shared else golden_params.shard(k, master[k], r, tp).to(dt).requires_grad_(True)
                       for k in golden_params.PARAM_ORDER})
    cfg = O.LayerConfig(meta["h"], meta["n"], rotary_pct=meta["rotary_pct"], gpt_j_residual=meta["gpt_j_residual"],
                        bias_gelu_fusion=meta["bias_gelu_fusion"], tp=tp)
    xin = x.clone().requires_grad_(True)
    y = O.layer_forward(cfg, shards, xin, O.causal_mask(meta["s"]))
    y.backward(dy)
    return y.detach(), xin.grad, shards


@pytest.mark.parametrize("name", _layer_files(1))
def test_layer_tp1_matches_reference(name):
    t, meta = golden_io.load(name)
    y, dx, shards = _run_oracle_layer(meta, t["x"], t["dy"], 1)
    psum = float(sum(p.detach().double().abs().sum() for p in shards[0].values()))
    assert abs(psum - meta["param_abs_sum"]) <= 1e-9 * meta["param_abs_sum"], "seeded parameters do not regenerate"
    if meta["dtype"] == "f32":
        tol = dict(rtol=1e-5, atol=1e-5)
    else:  # bf16 eager: identical op sequence -> identical roundings, except the TorchScript gelu
        tol = dict(rtol=2 ** -6, atol=2e-2) if meta["bias_gelu_fusion"] else dict(rtol=0, atol=0)
    torch.testing.assert_close(y.float(), t["y"].float(), **tol)
    torch.testing.assert_close(dx.float(), t["dx"].float(), **(tol if tol["atol"] else dict(rtol=2 ** -6, atol=2e-2)))
    for k in golden_params.PARAM_ORDER:
        if "g." + k in t:
            g = shards[0][k].grad.float()
            ref = t["g." + k].float()
            scale = ref.abs().max().clamp_min(1e-6)
            assert (g - ref).abs().max() / scale <= (1e-4 if meta["dtype"] == "f32" else 3e-2), k


def _rank_fixtures(name_r0):
    """{rank: tensors} of every rank stored for a TP fixture set (TP = 8 stores two of its eight ranks)."""
    base = name_r0[:-1]
    out = {}
    for p in glob.glob(os.path.join(golden_io.HERE, base + "*.npz")):
        r = int(os.path.basename(p)[len(base):-4])
        out[r] = golden_io.load(base + str(r))[0]
    return out


@pytest.mark.parametrize("name", _layer_files(2) + _layer_files(8))
def test_layer_tp_matches_reference(name):
    """TP = 2 (both residual variants at head size 64, sequential at head size 128 incl. the summit-1-3B width h=2304, n=18)
    and TP = 8 at the 6-7B width (h=4096, n=32): fixtures made by the reference with 2 / 8 gloo ranks."""
    t0, meta = golden_io.load(name)
    tp = meta["tp"]
    ranks = _rank_fixtures(name)
    y, dx, shards = _run_oracle_layer(meta, t0["x"], t0["dy"], tp)
    wide = meta["h"] >= 2048   # long bf16 contractions: eager CPU summation order (blocked differently for a stacked sum) shows
    tol = dict(rtol=1e-5, atol=1e-5) if meta["dtype"] == "f32" else dict(rtol=2 ** -6, atol=6e-2 if wide else 3e-2)
    for t in ranks.values():  # every rank holds the same y / dx
        torch.testing.assert_close(y.float(), t["y"].float(), **tol)
        torch.testing.assert_close(dx.float(), t["dx"].float(), **tol)
    for r, t in ranks.items():
        for k in ("attention.query_key_value.bias", "mlp.dense_h_to_4h.bias", "attention.dense.bias", "input_layernorm.weight"):
            g = shards[r][k].grad.float()
            if k == "attention.dense.bias" and meta["gpt_j_residual"]:
                g = g / tp  # gpt-j: every rank adds its own copy of the bias, so a rank's grad is 1/tp of the shared tensor's
            ref = t["g." + k].float()
            assert (g - ref).abs().max() / ref.abs().max().clamp_min(1e-6) <= (1e-4 if meta["dtype"] == "f32" else 3e-2), (r, k)


def _adapter_files():
    pat = os.path.join(golden_io.HERE, "adapter_layer_*.npz")
    return sorted({os.path.basename(p)[:-4] for p in glob.glob(pat) if not os.path.basename(p)[:-4].endswith("_r1")})


@pytest.mark.parametrize("name", _adapter_files())
def test_layer_with_adapters_matches_reference(name):
    """oracle.adapter / layer_forward with adapters against the reference layer wrapped by its own add_adapters
    (megatron/model/adapter.py:95-151), LM frozen, TP = 1 and TP = 2.  Under TP = 2 with the GPT-J residual the adapter
    (ReLU included) acts on each rank's PARTIAL sum; with the sequential residual it acts on the reduced output."""
    t0, meta = golden_io.load(name)
    tp, dt = meta["tp"], DT[meta["dtype"]]
    ranks = {0: t0} if tp == 1 else _rank_fixtures(name)
    rng = np.random.default_rng(meta["seed"])
    master = golden_params.master_params(rng, meta["h"])
    x = golden_params.rng_t(rng, (meta["s"], meta["b"], meta["h"])).to(dt)
    dy = golden_params.rng_t(rng, (meta["s"], meta["b"], meta["h"])).to(dt)
    ad = golden_params.adapter_params(rng, meta["h"], meta["h"] // meta["adapter_downsample_factor"])
    assert torch.equal(x, t0["x"]) and torch.equal(dy, t0["dy"]), "seeded inputs do not regenerate"
    shards = []
    for r in range(tp):
        p = {k: golden_params.shard(k, master[k], r, tp).to(dt) for k in golden_params.PARAM_ORDER}
        p.update({k: v.to(dt).clone().requires_grad_(True) for k, v in ad.items()})  # one copy per rank, as in the reference
        shards.append(p)
    psum = float(sum(v.detach().double().abs().sum() for v in shards[0].values()))
    assert abs(psum - meta["param_abs_sum"]) <= 1e-9 * meta["param_abs_sum"]
    cfg = O.LayerConfig(meta["h"], meta["n"], rotary_pct=meta["rotary_pct"], gpt_j_residual=meta["gpt_j_residual"],
                        bias_gelu_fusion=meta["bias_gelu_fusion"], tp=tp)
    xin = x.clone().requires_grad_(True)
    y = O.layer_forward(cfg, shards, xin, O.causal_mask(meta["s"]))
    y.backward(dy)
    f32 = meta["dtype"] == "f32"
    tol = dict(rtol=1e-5, atol=2e-5) if f32 else dict(rtol=2 ** -6, atol=3e-2)
    for r, t in ranks.items():
        torch.testing.assert_close(y.detach().float(), t["y"].float(), **tol)
        torch.testing.assert_close(xin.grad.float(), t["dx"].float(), **tol)
        # sequential residual: every rank computes the same adapter on the reduced tensor; the simulation keeps rank 0's copy
        src = shards[r] if meta["gpt_j_residual"] else shards[0]
        for k in golden_params.ADAPTER_KEYS:
            g, ref = src[k].grad.float(), t["g." + k].float()
            assert (g - ref).abs().max() / ref.abs().max().clamp_min(1e-6) <= (1e-4 if f32 else 3e-2), (r, k)
    # the wrapped layer's state-dict keys (adapter.py:106-107,25) as the product must expose them
    assert "attention.attn_block.query_key_value.weight" in meta["state_dict_keys"] and "mlp.adapter.2.bias" in meta["state_dict_keys"]


@pytest.mark.parametrize("V", [2048, 50304])
@pytest.mark.parametrize("tp", [1, 2])
def test_vocab_parallel_cross_entropy_matches_reference(V, tp):
    """oracle.vocab_parallel_cross_entropy / cross_entropy_loss against mpu.vocab_parallel_cross_entropy run by the
    reference on 1 and 2 gloo ranks (megatron/mpu/cross_entropy.py:27-117)."""
    fx = [golden_io.load(f"op_vocab_ce_V{V}_tp{tp}" + (f"_r{r}" if tp > 1 else ""))[0] for r in range(tp)]
    logits, labels, gout = fx[0]["logits"], fx[0]["labels"], fx[0]["gout"]
    vp = V // tp
    parts = [logits[..., r * vp:(r + 1) * vp].float().contiguous().requires_grad_(True) for r in range(tp)]
    loss = O.vocab_parallel_cross_entropy(parts, labels)
    loss.backward(gout)
    for r in range(tp):
        torch.testing.assert_close(loss.detach(), fx[r]["loss"], rtol=1e-6, atol=2e-6)
        torch.testing.assert_close(parts[r].grad, fx[r]["dlogits"], rtol=1e-5, atol=1e-7)
    # the TP = 1 masked-mean loss helper (gpt2_model.py:52-70) agrees with the per-token losses
    lm = (gout > 0.3).float()
    ref = (torch.cat([f["loss"] for f in fx[:1]]).view(-1) * lm.view(-1)).sum() / lm.sum()
    torch.testing.assert_close(O.cross_entropy_loss(logits, labels, lm), ref, rtol=1e-5, atol=1e-5)


def test_gptj_tp2_counts_row_biases_twice():
    """with gpt_j_residual the Row-linear biases are added on every rank before the single reduce."""
    h, n, s, b = 128, 2, 16, 1
    master = golden_params.master_params(np.random.default_rng(5), h)
    x = golden_params.rng_t(np.random.default_rng(6), (s, b, h))
    outs = {}
    for tp in (1, 2):
        cfg = O.LayerConfig(h, n, rotary_pct=0.25, gpt_j_residual=True, bias_gelu_fusion=True, tp=tp)
        shards = [{k: golden_params.shard(k, master[k], r, tp) for k in golden_params.PARAM_ORDER} for r in range(tp)]
        outs[tp] = O.layer_forward(cfg, shards, x, O.causal_mask(s))
    extra = master["attention.dense.bias"] + master["mlp.dense_4h_to_h.bias"]
    torch.testing.assert_close(outs[2] - outs[1], extra.expand_as(x), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_stack_loss_curve_matches_reference(dt):
    from tests.stack_harness import run_oracle_stack
    t, meta = golden_io.load(f"stack_L2_{dt}")
    losses = run_oracle_stack(meta, t["tokens"], t["loss_mask"], DT[dt])
    ref = t["losses"].tolist()
    tol = 2e-4 if dt == "f32" else 3e-2
    assert max(abs(a - b) for a, b in zip(losses, ref)) <= tol, (losses, ref)
    assert losses[-1] < losses[0] * 0.2  # and it trains (the reference's own bar: loss goes down)


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(golden_io.HERE, "op_embed_prefix_*.npz"))))
def test_image_prefix_and_embedding_match_reference(name):
    """the reference's ImagePrefix tail (image_prefix.py:217-252, encoder = identity) and
    EmbeddingPipe.forward + _pre_transformer_block (word_embeddings.py:182-205, gpt2_model.py:73-77), forward bit-exact,
    gradients of the prefix parameters and of the encoder output to round-off."""
    t, meta = golden_io.load(name)
    leaf = {k: t[k].clone().requires_grad_(True) for k in ("enc", "proj_w", "proj_b", "ln_w", "ln_b")}
    prefix = O.image_prefix_tail(leaf["enc"], leaf["proj_w"], leaf["proj_b"], leaf["ln_w"], leaf["ln_b"], meta["eps"])
    assert prefix.shape == (meta["b"], meta["P"], meta["h"])
    assert torch.equal(prefix.detach(), t["prefix"])
    x0 = O.embed_with_prefix(prefix, t["table"], t["tokens"])
    assert x0.shape == (meta["s"], meta["b"], meta["h"]) and torch.equal(x0.detach(), t["x0"])
    x0.backward(t["g"])
    tol = 1e-5 if meta["dtype"] == "f32" else 2.0 ** -7
    for k, gk in (("enc", "d_enc"), ("proj_w", "d_proj_w"), ("proj_b", "d_proj_b"), ("ln_w", "d_ln_w"), ("ln_b", "d_ln_b")):
        ref = t[gk].float()
        err = (leaf[k].grad.float() - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (k, err)

