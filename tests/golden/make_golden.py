#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run only in the build container (``/root/reference`` present):

    python tests/golden/make_golden.py

It imports the reference's hot-path modules (megatron/model/transformer.py and what it
pulls in) on CPU with the environment shims listed in SURVEY.md section 8(c) - none of which
touch the path's arithmetic - feeds them seeded inputs and stores inputs + outputs as
``.npz`` fixtures.  Nothing from the reference is copied: only data it computed.

The reference never travels to the GPU box; the tests only read the ``.npz`` files.
"""
import contextlib
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_io as _io  # noqa: E402

REF = os.environ.get("MM_REFERENCE", "/root/reference")


# --------------------------------------------------------------------------- shims
def install_shims():
    ds = types.ModuleType("deepspeed")
    ck = types.ModuleType("deepspeed.checkpointing")
    ck._MODEL_PARALLEL_RNG_TRACKER_NAME = "model-parallel-rng"
    ck._CUDA_RNG_STATE_TRACKER = None
    rt = types.ModuleType("deepspeed.runtime")
    ac = types.ModuleType("deepspeed.runtime.activation_checkpointing")
    acc = types.ModuleType("deepspeed.runtime.activation_checkpointing.checkpointing")

    class _Tracker:  # dropout is 0 in every listed config -> fork() may be a null context
        def fork(self, name=None):
            return contextlib.nullcontext()

    acc._set_cuda_rng_state = lambda *a, **k: None
    acc.checkpoint = lambda f, *a: f(*a)
    acc.model_parallel_cuda_manual_seed = lambda s: None
    acc.get_cuda_rng_tracker = lambda: _Tracker()
    ds.checkpointing, ds.runtime = ck, rt
    rt.activation_checkpointing = ac
    ac.checkpointing = acc
    for n, m in [
        ("deepspeed", ds),
        ("deepspeed.checkpointing", ck),
        ("deepspeed.runtime", rt),
        ("deepspeed.runtime.activation_checkpointing", ac),
        ("deepspeed.runtime.activation_checkpointing.checkpointing", acc),
    ]:
        sys.modules[n] = m
    mg = types.ModuleType("megatron")
    mg.__path__ = [REF + "/megatron"]
    mg.print_rank_0 = lambda *a, **k: None
    mm = types.ModuleType("megatron.model")
    mm.__path__ = [REF + "/megatron/model"]
    sys.modules["megatron"] = mg
    sys.modules["megatron.model"] = mm
    torch.cuda.current_device = lambda: "cpu"  # transformer.py:342 hard-codes it


def install_torchtyping_stub():
    """megatron/model/adapter.py:3 imports torchtyping only for annotations; an environment shim that touches no arithmetic."""
    tt = types.ModuleType("torchtyping")

    class TensorType:
        def __class_getitem__(cls, item):
            return torch.Tensor

    tt.TensorType = TensorType
    tt.patch_typeguard = lambda: None  # imported by image_prefix.py:3, its call at :16 is commented out
    sys.modules["torchtyping"] = tt


def init_dist(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)


DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
PREC = {"f32": "fp32", "bf16": "bfloat16", "f16": "fp16"}


def make_args(h, n, L, s, dt, rotary_pct, gpt_j, gelu_fusion, tp):
    return SimpleNamespace(
        hidden_size=h, num_attention_heads=n, num_layers=L, seq_length=s,
        precision=PREC[dt], params_dtype=DT[dt], pos_emb="rotary", rotary_pct=rotary_pct,
        rotary_emb_base=10000, norm="layernorm", layernorm_epsilon=1e-5, activation="gelu",
        bias_gelu_fusion=gelu_fusion, bias_dropout_fusion=False,
        scaled_upper_triang_masked_softmax_fusion=False, scaled_masked_softmax_fusion=False,
        apply_query_key_layer_scaling=False, attention_softmax_in_fp32=False,
        attention_dropout=0.0, hidden_dropout=0.0, gpt_j_residual=gpt_j, gpt_j_tied=False,
        attention_config=["global"] * L, use_mup=False, use_cpu_initialization=True,
        model_parallel_size=tp, fp32_allreduce=False, onnx_safe=False,
        mup_init_scale=1.0,
    )


from golden_params import rng_t, PARAM_ORDER, master_params, shard, causal_mask, stack_init  # noqa: E402


# ------------------------------------------------------------------- whole layer
def run_layer_case(name, s, b, h, n, dt, rotary_pct, gpt_j, gelu_fusion, tp, rank, seed, store_params=False, store_grads=True):
    from megatron.model.transformer import ParallelTransformerLayer
    from megatron.model.gpt2_model_mask import gpt2_attention_mask_func  # installed below

    rng = np.random.default_rng(seed)
    args = make_args(h, n, 2, s, dt, rotary_pct, gpt_j, gelu_fusion, tp)
    layer = ParallelTransformerLayer(
        neox_args=args, attention_mask_func=gpt2_attention_mask_func,
        init_method=torch.nn.init.zeros_, output_layer_init_method=torch.nn.init.zeros_,
        layer_number=0, rotary=True,
    )
    master = master_params(rng, h)
    x = rng_t(rng, (s, b, h))
    dy = rng_t(rng, (s, b, h))
    with torch.no_grad():
        sd = dict(layer.named_parameters())
        for k in PARAM_ORDER:
            sd[k].copy_(shard(k, master[k], rank, tp))
    layer = layer.to(DT[dt])  # the engine casts the built module (SURVEY 8c "Important")
    layer.train()
    xin = x.to(DT[dt]).requires_grad_(True)
    y = layer(xin, causal_mask(s))
    y.backward(dy.to(DT[dt]))
    out = {"x": xin.detach(), "dy": dy.to(DT[dt]), "y": y.detach(), "dx": xin.grad}
    for k, p in layer.named_parameters():
        if store_params:
            out["p." + k] = p.detach()
        if store_grads or p.dim() == 1:  # big case: only the 1-D grads are stored
            out["g." + k] = p.grad
    # parameters are regenerated by the tests from `seed` (golden_params.master_params); pin them
    psum = float(sum(p.detach().double().abs().sum() for p in layer.parameters()))
    meta = dict(param_abs_sum=psum, s=s, b=b, h=h, n=n, dtype=dt, rotary_pct=rotary_pct, gpt_j_residual=gpt_j,
                bias_gelu_fusion=gelu_fusion, tp=tp, rank=rank, seed=seed,
                source="megatron/model/transformer.py:644-801 run on CPU")
    _io.save(name, out, meta)
    print("wrote", name, "y[0,0,:3]=", y.detach().float()[0, 0, :3].tolist())


LAYER_CASES = []
_F32_COMBOS = {(True, True, 0.25), (True, False, 1.0), (False, True, 1.0), (False, False, 0.25)}
for _dt in ("f32", "bf16"):
    for _gj in (True, False):
        for _gf in (True, False):
            for _rp in (0.25, 1.0):
                if _dt == "f32" and (_gj, _gf, _rp) not in _F32_COMBOS:
                    continue
                LAYER_CASES.append(dict(s=64, b=2, h=128, n=2, dt=_dt, rotary_pct=_rp, gpt_j=_gj, gelu_fusion=_gf))
# TP=2 subset: both residual variants in both dtypes
TP_CASES = [c for c in LAYER_CASES if (c["gpt_j"], c["gelu_fusion"], c["rotary_pct"]) in {(True, True, 0.25), (False, False, 0.25)}]


def layer_case_name(c, tp, rank=None):
    nm = "layer_s{s}b{b}h{h}n{n}_{dt}_rot{r}_gptj{g}_tanh{t}_tp{tp}".format(
        s=c["s"], b=c["b"], h=c["h"], n=c["n"], dt=c["dt"], r=int(c["rotary_pct"] * 100),
        g=int(c["gpt_j"]), t=int(c["gelu_fusion"]), tp=tp)
    if rank is not None:
        nm += "_r%d" % rank
    return nm


# ------------------------------------------------------------------- head size 128 (summit-1-3B / 6-7B shapes)
# configs/summit-1-3B.yml:20-26 (h=2304, n=18), configs/6-7B.yml:9-15 (h=4096, n=32): hn = 128, sequential residual,
# exact-erf GeLU, rotary_pct 1.0
HN128_SMALL = [dict(s=64, b=2, h=256, n=2, dt=_dt, rotary_pct=1.0, gpt_j=False, gelu_fusion=False) for _dt in ("f32", "bf16")]
HN128_1_3B = dict(s=128, b=1, h=2304, n=18, dt="bf16", rotary_pct=1.0, gpt_j=False, gelu_fusion=False)   # TP = 2
HN128_6_7B = dict(s=64, b=1, h=4096, n=32, dt="bf16", rotary_pct=1.0, gpt_j=False, gelu_fusion=False)    # TP = 8
TP8_STORE_RANKS = (0, 5)


# ------------------------------------------------------------------- layer wrapped with the reference's adapters
ADAPTER_CASES = []
for _dt in ("f32", "bf16"):
    ADAPTER_CASES.append(dict(s=64, b=2, h=512, n=8, dt=_dt, rotary_pct=0.25, gpt_j=True, gelu_fusion=True))
    ADAPTER_CASES.append(dict(s=64, b=2, h=512, n=8, dt=_dt, rotary_pct=1.0, gpt_j=False, gelu_fusion=False))


def run_adapter_case(name, s, b, h, n, dt, rotary_pct, gpt_j, gelu_fusion, tp, rank, seed):
    """ParallelTransformerLayer wrapped by the reference's own add_adapters (megatron/model/adapter.py:119-151, as called at
    megatron/training.py:421-432) with the LM frozen (training.py:454-459): only adapter parameters get gradients."""
    from megatron.model.transformer import ParallelTransformerLayer
    from megatron.model.adapter import add_adapters
    from megatron.model.gpt2_model_mask import gpt2_attention_mask_func
    from golden_params import adapter_params, ADAPTER_KEYS

    rng = np.random.default_rng(seed)
    args = make_args(h, n, 2, s, dt, rotary_pct, gpt_j, gelu_fusion, tp)
    layer = ParallelTransformerLayer(
        neox_args=args, attention_mask_func=gpt2_attention_mask_func,
        init_method=torch.nn.init.zeros_, output_layer_init_method=torch.nn.init.zeros_,
        layer_number=0, rotary=True)
    master = master_params(rng, h)
    x = rng_t(rng, (s, b, h))
    dy = rng_t(rng, (s, b, h))
    ad = adapter_params(rng, h, h // 8)
    with torch.no_grad():
        sd = dict(layer.named_parameters())
        for k in PARAM_ORDER:
            sd[k].copy_(shard(k, master[k], rank, tp))
    holder = torch.nn.ModuleList([layer])
    add_adapters(args, holder, downsample_factor=8, location="mlp")
    add_adapters(args, holder, downsample_factor=8, location="attention")
    sd = dict(layer.named_parameters())
    assert "attention.attn_block.dense.weight" in sd and "mlp.adapter.2.bias" in sd, sorted(sd)
    with torch.no_grad():
        for k in ADAPTER_KEYS:  # plain nn.Linear: identical on every TP rank
            sd[k].copy_(ad[k])
    for nme, p in layer.named_parameters():
        p.requires_grad = "adapter" in nme
    layer = layer.to(DT[dt])
    layer.train()
    xin = x.to(DT[dt]).requires_grad_(True)
    y = layer(xin, causal_mask(s))
    y.backward(dy.to(DT[dt]))
    out = {"x": xin.detach(), "dy": dy.to(DT[dt]), "y": y.detach(), "dx": xin.grad}
    sd = dict(layer.named_parameters())
    for k in ADAPTER_KEYS:
        out["g." + k] = sd[k].grad
    assert all(p.grad is None for nme, p in layer.named_parameters() if "adapter" not in nme)
    psum = float(sum(p.detach().double().abs().sum() for p in layer.parameters()))
    meta = dict(param_abs_sum=psum, s=s, b=b, h=h, n=n, dtype=dt, rotary_pct=rotary_pct, gpt_j_residual=gpt_j,
                bias_gelu_fusion=gelu_fusion, tp=tp, rank=rank, seed=seed, adapter_downsample_factor=8,
                state_dict_keys=sorted(layer.state_dict().keys()),
                source="megatron/model/transformer.py:644-801 wrapped by megatron/model/adapter.py:95-151 run on CPU")
    _io.save(name, out, meta)
    print("wrote", name, "y[0,0,:3]=", y.detach().float()[0, 0, :3].tolist())


def adapter_case_name(c, tp, rank=None):
    return "adapter_" + layer_case_name(c, tp, rank)


# ------------------------------------------------------------------- vocab-parallel cross entropy
CE_CASES = [dict(b=2, s=48, V=2048, seed=7100), dict(b=1, s=4, V=50304, seed=7101)]


def run_ce_case(c, tp, rank):
    """mpu.vocab_parallel_cross_entropy(output.float().contiguous(), labels) as megatron/model/gpt2_model.py:52-70 calls it
    (megatron/mpu/cross_entropy.py:27-117), this rank's vocabulary partition of seeded bf16 logits."""
    from megatron import mpu
    rng = np.random.default_rng(c["seed"])
    b, s, V = c["b"], c["s"], c["V"]
    logits = rng_t(rng, (b, s, V), 3.0).to(torch.bfloat16)
    labels = torch.from_numpy(rng.integers(0, V, (b, s)))
    labels[0, 0], labels[0, 1] = 0, V - 1                       # partition edges
    labels[0, 2], labels[0, 3] = V // 2 - 1, V // 2
    gout = torch.from_numpy(rng.random((b, s)).astype(np.float32))  # upstream gradient (loss_mask / sum in the caller)
    vp = V // tp
    mine16 = logits[..., rank * vp:(rank + 1) * vp].contiguous().requires_grad_(True)
    mine32 = mine16.float().contiguous()
    mine32.retain_grad()
    loss = mpu.vocab_parallel_cross_entropy(mine32, labels)
    loss.backward(gout)
    name = "op_vocab_ce_V%d_tp%d" % (V, tp) + ("_r%d" % rank if tp > 1 else "")
    _io.save(name, dict(logits=logits, labels=labels, gout=gout, loss=loss.detach(), dlogits=mine32.grad, dlogits16=mine16.grad),
             dict(b=b, s=s, V=V, tp=tp, rank=rank, seed=c["seed"], source="megatron/mpu/cross_entropy.py:27-117 run on CPU (gloo)"))
    print("wrote", name, "loss[0,:3]=", loss.detach()[0, :3].tolist())


# ------------------------------------------------------------------- image prefix tail + embedding concat
def install_encoder_stubs():
    """megatron/model/image_prefix.py:7-8 imports timm and open_clip at module level for its encoder FACTORIES
    (image_prefix.py:32-149), which need network weights and are out of scope; the tail that is pinned here
    (proj -> dropout -> LayerNorm -> unsqueeze, :217-252) never calls them.  Import-only stubs, no arithmetic."""
    for name in ("timm", "open_clip"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)


EMBED_CASES = [dict(b=3, s=24, V=96, h=256, enc=64, ndim=2, seed=8100),    # pooled encoder output [b, enc] -> 1 prefix token (openclip-H as coded)
               dict(b=2, s=24, V=96, h=256, enc=64, ndim=3, P=5, seed=8101),  # un-pooled [b, P, enc] -> P prefix tokens
               dict(b=4, s=40, V=128, h=1024, enc=1024, ndim=2, seed=8102)]   # the 410M widths (proj 1024 -> 1024)


def run_embed_case(c, dt):
    """The reference's own ImagePrefix.forward (encoder replaced by the identity: its pooled output is the INPUT) and
    EmbeddingPipe.forward (megatron/model/word_embeddings.py:182-205) followed by _pre_transformer_block
    (megatron/model/gpt2_model.py:73-77), eval mode (dropout is the identity; its training-mode mask comes from the device
    generator and cannot be pinned), backward through the prefix only (frozen word embeddings, training.py:454-459)."""
    from megatron import mpu
    from megatron.model import image_prefix as IP
    from megatron.model import word_embeddings as WE
    rng = np.random.default_rng(c["seed"])
    b, s, V, h, enc = c["b"], c["s"], c["V"], c["h"], c["enc"]
    pre = object.__new__(IP.ImagePrefix)        # __init__ builds the CLIP encoder (network weights): set the tail's attributes
    torch.nn.Module.__init__(pre)
    pre.encoder_type = "openclip-H"             # in ENCODER_SEQ_LENS: no rearrange (image_prefix.py:235-240)
    pre.enc = torch.nn.Identity()
    pre.out_dim = h
    pre.proj = torch.nn.Linear(enc, h)
    pre.dropout = torch.nn.Dropout(0.1)
    pre.use_layernorm = True
    pre.ln = torch.nn.LayerNorm(h)
    with torch.no_grad():
        pre.proj.weight.copy_(rng_t(rng, (h, enc), enc ** -0.5))
        pre.proj.bias.copy_(rng_t(rng, (h,), 0.1))
        pre.ln.weight.copy_(rng_t(rng, (h,), 0.1, 1.0))
        pre.ln.bias.copy_(rng_t(rng, (h,), 0.1))
    args = make_args(h, 2, 2, s, dt, 1.0, False, False, 1)
    emb = object.__new__(WE.EmbeddingPipe)      # Embedding.__init__ is fine, EmbeddingPipe.__init__ would build the encoder
    WE.Embedding.__init__(emb, SimpleNamespace(**vars(args), mup_embedding_mult=1.0, mup_rp_embedding_mult=1.0,
                                               use_bnb_optimizer=False, opt_pos_emb_offset=0),
                          h, V, s, 0.0, torch.nn.init.zeros_, 0, True)
    emb.image_prefix = pre
    table = rng_t(rng, (V, h), 0.05)
    with torch.no_grad():
        emb.word_embeddings.weight.copy_(table)
    emb = emb.to(DT[dt])
    emb.eval()
    for nme, p_ in emb.named_parameters():
        p_.requires_grad = "image_prefix" in nme
    x_enc = rng_t(rng, (b, enc) if c["ndim"] == 2 else (b, c["P"], enc)).to(DT[dt]).requires_grad_(True)
    tokens = torch.from_numpy(rng.integers(0, V, (b, s)))
    pos = torch.arange(s).unsqueeze(0).expand(b, s)
    out, _ = emb((x_enc, tokens, pos, causal_mask(s)))
    x0 = out.transpose(0, 1).contiguous()       # gpt2_model.py:73-77
    prefix = pre(x_enc)
    g = rng_t(rng, tuple(x0.shape)).to(DT[dt])
    x0.backward(g)
    name = "op_embed_prefix_h%d_nd%d_%s" % (h, c["ndim"], dt)
    _io.save(name, dict(enc=x_enc.detach(), tokens=tokens, table=emb.word_embeddings.weight.detach(),
                        proj_w=pre.proj.weight.detach(), proj_b=pre.proj.bias.detach(), ln_w=pre.ln.weight.detach(),
                        ln_b=pre.ln.bias.detach(), prefix=prefix.detach(), x0=x0.detach(), g=g, d_enc=x_enc.grad,
                        d_proj_w=pre.proj.weight.grad, d_proj_b=pre.proj.bias.grad, d_ln_w=pre.ln.weight.grad,
                        d_ln_b=pre.ln.bias.grad),
             dict(b=b, s=s, V=V, h=h, enc=enc, P=int(prefix.shape[1]), seed=c["seed"], dtype=dt, eps=1e-5,
                  source="megatron/model/image_prefix.py:217-252 + megatron/model/word_embeddings.py:182-205 + "
                         "megatron/model/gpt2_model.py:73-77 run on CPU (eval mode, encoder = identity)"))
    print("wrote", name, "x0[0,0,:3]=", x0.detach().float()[0, 0, :3].tolist())


# ------------------------------------------------------------------- per-op cases
def gen_ops():
    from megatron.model import activations, fused_bias_dropout, positional_embeddings
    from megatron.model.fused_softmax import FusedScaleMaskSoftmax, SoftmaxFusionTypes
    from megatron.model.gpt2_model_mask import gpt2_attention_mask_func
    from megatron import mpu

    rng = np.random.default_rng(7)
    # LayerNorm (norms.py:16 -> torch.nn.LayerNorm), module cast like the engine does
    for h in (512, 1024):
        for dt in ("f32", "bf16"):
            ln = torch.nn.LayerNorm(h, eps=1e-5)
            with torch.no_grad():
                ln.weight.copy_(rng_t(rng, (h,), 0.2, 1.0))
                ln.bias.copy_(rng_t(rng, (h,), 0.2))
            ln = ln.to(DT[dt])
            x = rng_t(rng, (24, h), 2.0, 0.5).to(DT[dt]).requires_grad_(True)
            dy = rng_t(rng, (24, h)).to(DT[dt])
            y = ln(x)
            y.backward(dy)
            _io.save(f"op_layernorm_h{h}_{dt}", dict(x=x.detach(), w=ln.weight.detach(), b=ln.bias.detach(),
                     dy=dy, y=y.detach(), dx=x.grad, dw=ln.weight.grad, db=ln.bias.grad),
                     dict(eps=1e-5, source="megatron/model/norms.py:16-31"))

    # Rotary tables (positional_embeddings.py:38-63) after the engine's module cast
    for d, S in ((16, 2048), (64, 2048), (128, 512)):
        for dt in ("f32", "bf16", "f16"):
            rot = positional_embeddings.RotaryEmbedding(d, base=10000, precision=DT[dt]).to(DT[dt])
            cos, sin = rot(torch.zeros(1, dtype=DT[dt]), seq_len=S)
            _io.save(f"op_rotary_table_d{d}_{dt}", dict(cos=cos.reshape(S, d), sin=sin.reshape(S, d),
                                                         inv_freq=rot.inv_freq),
                     dict(d=d, seq_len=S, base=10000, source="megatron/model/positional_embeddings.py:38-63"))
    # Rotary apply (positional_embeddings.py:69-92 + transformer.py:566-596), partial and full
    for d in (16, 64):
        for dt in ("f32", "bf16"):
            s, b, np_, hn = 40, 2, 3, 64
            rot = positional_embeddings.RotaryEmbedding(d, base=10000, precision=DT[dt]).to(DT[dt])
            q = rng_t(rng, (s, b, np_, hn)).to(DT[dt]).requires_grad_(True)
            k = rng_t(rng, (s, b, np_, hn)).to(DT[dt]).requires_grad_(True)
            cos, sin = rot(q, seq_len=s)
            fn = positional_embeddings.apply_rotary_pos_emb_torch if dt == "bf16" else positional_embeddings.apply_rotary_pos_emb
            qr, kr = fn(q[..., :d], k[..., :d], cos, sin, offset=0)
            if d < hn:
                qr = torch.cat((qr, q[..., d:]), dim=-1)
                kr = torch.cat((kr, k[..., d:]), dim=-1)
            dq = rng_t(rng, (s, b, np_, hn)).to(DT[dt])
            dk = rng_t(rng, (s, b, np_, hn)).to(DT[dt])
            torch.autograd.backward([qr, kr], [dq, dk])
            _io.save(f"op_rotary_apply_d{d}_{dt}", dict(q=q.detach(), k=k.detach(), cos=cos.reshape(s, d), sin=sin.reshape(s, d),
                     q_out=qr.detach(), k_out=kr.detach(), dq_out=dq, dk_out=dk, dq=q.grad, dk=k.grad),
                     dict(d=d, source="megatron/model/positional_embeddings.py:69-92"))

    # bias-GeLU (activations.py:62-96) and exact gelu (transformer.py:127-130)
    for dt in ("f32", "bf16"):
        y = rng_t(rng, (16, 512), 2.0).to(DT[dt]).requires_grad_(True)
        bias = rng_t(rng, (512,), 0.5).to(DT[dt]).requires_grad_(True)
        g = rng_t(rng, (16, 512)).to(DT[dt])
        out = activations.bias_gelu_impl(y, bias)
        out.backward(g)
        y2 = y.detach().clone().requires_grad_(True)
        out_erf = torch.nn.functional.gelu(y2 + bias.detach())
        out_erf.backward(g)
        _io.save(f"op_bias_gelu_{dt}", dict(y=y.detach(), bias=bias.detach(), g=g, out_tanh=out.detach(),
                 dy_tanh=y.grad, dbias_tanh=bias.grad, out_erf=out_erf.detach(), dy_erf=y2.grad),
                 dict(source="megatron/model/activations.py:62-96; transformer.py:127-130"))

    # bias-dropout-add (fused_bias_dropout.py:28-55), p = 0 as in every listed config
    for dt in ("f32", "bf16"):
        x = rng_t(rng, (12, 2, 256)).to(DT[dt])
        bias = rng_t(rng, (256,)).to(DT[dt])
        res = rng_t(rng, (12, 2, 256)).to(DT[dt])
        o1 = fused_bias_dropout.bias_dropout_add_fused_train(x, bias.expand_as(x), res, 0.0)
        o2 = fused_bias_dropout.bias_dropout_add_fused_train(x, bias.expand_as(x), None, 0.0)
        o3 = fused_bias_dropout.get_bias_dropout_add(False)(x, bias.expand_as(x), res, 0.1)
        _io.save(f"op_bias_dropout_add_{dt}", dict(x=x, bias=bias, residual=res, out_res=o1, out_nores=o2, out_eval=o3),
                 dict(source="megatron/model/fused_bias_dropout.py:28-55"))

    # Eager softmax fallback (fused_softmax.py:184-199) - the reference's CPU-runnable softmax path
    for dt in ("f32", "bf16", "f16"):
        for sk in (17, 128, 512):
            for (scale, in32) in ((None, False), (0.5, True)):
                if dt == "f32" and in32:
                    continue
                if sk == 512 and (dt != "bf16" or in32):
                    continue
                b, np_ = (2, 2) if sk < 512 else (1, 1)
                sm = FusedScaleMaskSoftmax(input_in_fp16=(dt == "f16"), input_in_bf16=(dt == "bf16"),
                                           fusion_type=SoftmaxFusionTypes.none, mask_func=gpt2_attention_mask_func,
                                           softmax_in_fp32=in32, scale=scale)
                x = rng_t(rng, (b, np_, sk, sk), 3.0).to(DT[dt])
                xin = x.clone().requires_grad_(True)
                xc = xin * 1.0  # masked_fill_ is in place: keep the leaf intact
                mask = causal_mask(sk)
                # a general (non-causal) mask as well: random with diagonal kept open
                rm = torch.from_numpy(rng.random((b, 1, sk, sk)) < 0.3)
                rm = rm & ~torch.eye(sk, dtype=torch.bool).view(1, 1, sk, sk)
                p = sm(xc, mask)
                dyv = rng_t(rng, (b, np_, sk, sk)).to(DT[dt])
                p.backward(dyv)
                p2 = sm(x.clone(), rm)
                tag = "s%s_in32%d" % ("none" if scale is None else "05", int(in32))
                _io.save(f"op_softmax_eager_{dt}_sk{sk}_{tag}", dict(x=x, dy=dyv, probs_causal=p.detach(), dx_causal=xin.grad,
                         mask_general=rm, probs_general=p2.detach()),
                         dict(scale=scale, softmax_in_fp32=in32, source="megatron/model/fused_softmax.py:184-199"))

    # Column / Row parallel linears, TP = 1 (mpu/layers.py:549-564, 736-757)
    for dt in ("f32", "bf16"):
        a = make_args(128, 4, 2, 16, dt, 1.0, False, False, 1)
        col = mpu.ColumnParallelLinear(a, 128, 384, gather_output=False, init_method=torch.nn.init.zeros_)
        row = mpu.RowParallelLinear(a, 384, 128, input_is_parallel=True, init_method=torch.nn.init.zeros_, skip_bias_add=True)
        with torch.no_grad():
            col.weight.copy_(rng_t(rng, (384, 128), 0.1)); col.bias.copy_(rng_t(rng, (384,), 0.1))
            row.weight.copy_(rng_t(rng, (128, 384), 0.1)); row.bias.copy_(rng_t(rng, (128,), 0.1))
        x = rng_t(rng, (16, 2, 128)).to(DT[dt]).requires_grad_(True)
        yc, bc = col(x)
        yr, br = row(yc)
        dy = rng_t(rng, (16, 2, 128)).to(DT[dt])
        yr.backward(dy)
        _io.save(f"op_linear_tp1_{dt}", dict(x=x.detach(), col_w=col.weight.detach(), col_b=col.bias.detach(),
                 row_w=row.weight.detach(), row_b=row.bias.detach(), y_col=yc.detach(), y_row=yr.detach(), row_bias_out=br.detach(),
                 dy=dy, dx=x.grad, d_col_w=col.weight.grad, d_col_b=col.bias.grad, d_row_w=row.weight.grad),
                 dict(source="megatron/mpu/layers.py:549-564,736-757", col_bias_out_is_none=bc is None))


# ------------------------------------------------------------------- 2-layer stack loss curve
def gen_stack():
    from megatron.model.transformer import ParallelTransformerLayer
    from megatron.model.gpt2_model_mask import gpt2_attention_mask_func

    s, b, h, n, L, V, steps = 64, 4, 128, 2, 2, 512, 20
    for dt in ("f32", "bf16"):
        rng = np.random.default_rng(99)
        args = make_args(h, n, L, s, dt, 0.25, True, True, 1)
        emb = torch.nn.Embedding(V, h)
        layers = torch.nn.ModuleList([
            ParallelTransformerLayer(neox_args=args, attention_mask_func=gpt2_attention_mask_func,
                                     init_method=torch.nn.init.zeros_, output_layer_init_method=torch.nn.init.zeros_,
                                     layer_number=i, rotary=True) for i in range(L)])
        fln = torch.nn.LayerNorm(h, eps=1e-5)
        head = torch.nn.Linear(h, V, bias=False)
        init = stack_init(rng, V, h, L)
        with torch.no_grad():
            emb.weight.copy_(init["emb"])
            for i, l in enumerate(layers):
                sd = dict(l.named_parameters())
                for k in PARAM_ORDER:
                    sd[k].copy_(init[f"l{i}.{k}"])
            fln.weight.copy_(init["fln.w"]); fln.bias.copy_(init["fln.b"]); head.weight.copy_(init["head"])
        model = torch.nn.ModuleList([emb, layers, fln, head]).to(DT[dt])
        tokens = torch.from_numpy(rng.integers(0, V, (b, s + 1)))
        loss_mask = torch.ones(b, s)
        loss_mask[:, -5:] = 0.0
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.95), eps=1e-8)
        mask = causal_mask(s)
        losses = []
        for it in range(steps):
            opt.zero_grad()
            x = emb(tokens[:, :-1]).transpose(0, 1).contiguous()
            for l in layers:
                x = l(x, mask)
            x = x.transpose(0, 1).contiguous()
            logits = head(fln(x))
            lo = torch.nn.functional.cross_entropy(logits.float().view(-1, V), tokens[:, 1:].reshape(-1), reduction="none")
            loss = (lo * loss_mask.view(-1)).sum() / loss_mask.sum()
            loss.backward()
            opt.step()
            losses.append(float(loss))
        out = dict(tokens=tokens, loss_mask=loss_mask, losses=torch.tensor(losses, dtype=torch.float64))
        _io.save(f"stack_L2_{dt}", out, dict(seed=99, init_abs_sum=float(sum(v.double().abs().sum() for v in init.values())), s=s, b=b, h=h, n=n, L=L, V=V, steps=steps, lr=1e-3, betas=[0.9, 0.95], eps=1e-8,
                                             rotary_pct=0.25, gpt_j_residual=True, bias_gelu_fusion=True,
                                             source="2x megatron/model/transformer.py ParallelTransformerLayer + LayerNorm + Linear + CE; torch.optim.Adam"))
        print("stack", dt, ["%.4f" % l for l in losses])


# ------------------------------------------------------------------- drivers
def install_mask_module():
    """gpt2_model.py needs deepspeed.pipe; its 2-line mask func (gpt2_model.py:47-49) is re-stated."""
    m = types.ModuleType("megatron.model.gpt2_model_mask")

    def gpt2_attention_mask_func(attention_scores, ltor_mask):
        attention_scores.masked_fill_(ltor_mask, -10000.0)
        return attention_scores

    m.gpt2_attention_mask_func = gpt2_attention_mask_func
    sys.modules["megatron.model.gpt2_model_mask"] = m


def worker_tp(rank, world, port, which):
    install_shims(); install_mask_module(); install_torchtyping_stub()
    init_dist(rank, world, port)
    from megatron import mpu
    mpu.initialize_model_parallel(world)
    torch.manual_seed(0)
    if "tp2" in which:
        for i, c in enumerate(TP_CASES):
            run_layer_case(layer_case_name(c, world, rank), tp=world, rank=rank, seed=2000 + i, store_params=False, **c)
    if "hn128" in which and world == 2:
        for i, c in enumerate(HN128_SMALL):
            run_layer_case(layer_case_name(c, world, rank), tp=world, rank=rank, seed=5100 + i, store_params=False, **c)
        run_layer_case(layer_case_name(HN128_1_3B, world, rank), tp=world, rank=rank, seed=5200, store_grads=False, **HN128_1_3B)
    if "hn128" in which and world == 8:
        torch.set_num_threads(1)
        c = HN128_6_7B
        if rank in TP8_STORE_RANKS:
            run_layer_case(layer_case_name(c, world, rank), tp=world, rank=rank, seed=5300, store_grads=False, **c)
        else:  # takes part in the collectives; its own results are not stored
            run_layer_case("_discard_r%d" % rank, tp=world, rank=rank, seed=5300, store_grads=False, **c)
            os.remove(os.path.join(HERE, "_discard_r%d.npz" % rank))
    if "adapters" in which and world == 2:
        for i, c in enumerate(ADAPTER_CASES):
            run_adapter_case(adapter_case_name(c, world, rank), tp=world, rank=rank, seed=6100 + i, **c)
    if "ce" in which and world == 2:
        for c in CE_CASES:
            run_ce_case(c, world, rank)
    torch.distributed.barrier()


def main():
    which = sys.argv[1:] or ["ops", "layers", "big", "stack", "tp2", "hn128", "adapters", "ce", "embed"]
    import torch.multiprocessing as mp
    if {"tp2", "hn128", "adapters", "ce"} & set(which):
        mp.spawn(worker_tp, args=(2, 29533, which), nprocs=2, join=True)
    if "hn128" in which:
        mp.spawn(worker_tp, args=(8, 29534, which), nprocs=8, join=True)
    install_shims(); install_mask_module(); install_torchtyping_stub()
    init_dist(0, 1, 29532)
    from megatron import mpu
    mpu.initialize_model_parallel(1)
    torch.manual_seed(0)
    if "ops" in which:
        gen_ops()
    if "layers" in which:
        for i, c in enumerate(LAYER_CASES):
            run_layer_case(layer_case_name(c, 1), tp=1, rank=0, seed=1000 + i, **c)
    if "big" in which:
        # 410M layer shape at a short sequence; params are regenerated from the seed by the test
        # (master_params(default_rng(seed), h) in tests/golden/golden_params.py), only activations are stored.
        c = dict(s=128, b=2, h=1024, n=16, dt="bf16", rotary_pct=0.25, gpt_j=True, gelu_fusion=True)
        run_layer_case(layer_case_name(c, 1), tp=1, rank=0, seed=4100, store_grads=False, **c)
    if "stack" in which:
        gen_stack()
    if "hn128" in which:
        for i, c in enumerate(HN128_SMALL):
            run_layer_case(layer_case_name(c, 1), tp=1, rank=0, seed=5000 + i, **c)
    if "adapters" in which:
        for i, c in enumerate(ADAPTER_CASES):
            run_adapter_case(adapter_case_name(c, 1), tp=1, rank=0, seed=6000 + i, **c)
    if "ce" in which:
        for c in CE_CASES:
            run_ce_case(c, 1, 0)
    if "embed" in which:
        install_encoder_stubs()
        for c in EMBED_CASES:
            for dt in (("bf16",) if c["h"] >= 1024 else ("f32", "bf16")):  # the wide case: 16-bit only (fixture size)
                run_embed_case(c, dt)


if __name__ == "__main__":
    main()
