"""Minimal caller of the hot path: the MAGMA model around the layers, reduced to what a training step needs.

Mirrors what the reference assembles in GPT2ModelPipe (megatron/model/gpt2_model.py:166-305) for the
`magma_pythia_410M` config: image prefix (proj -> dropout -> LayerNorm, image_prefix.py:232-252; the CLIP encoder's
pooled output is an INPUT here, the encoder itself is out of scope), word embeddings, the concat
`[img | text[:, :s-P]]` (word_embeddings.py:197-203), [b,s,h] -> [s,b,h] (gpt2_model.py:73-77), L layers with adapters
(training.py:421-432), final LayerNorm, untied column-parallel LM head (transformer.py:816-839) and the masked-mean
cross entropy (gpt2_model.py:52-70).  `freeze_lm` keeps only `adapter` / `image_prefix` parameters trainable
(training.py:454-459).
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import functional as Fn
from . import mpu
from .checkpointing import checkpoint_policy
from .model import LayerNorm, ParallelTransformerLayer, add_adapters, gpt2_attention_mask_func

# lr / weight_decay: the YAML's optimizer block and weight-decay key (magma_pythia_410M.yml takes magma_setup.yml:38-45);
# betas (0.9, 0.95), eps 1e-8 and gradient_clipping 1.0 are the same in all of them (tests/golden/configs.json)
CONFIGS = {
    # configs/magma_pythia_410M.yml:5-33 + configs/magma_setup.yml:50-51 (micro-batch 4)
    "magma_pythia_410M": dict(hidden_size=1024, num_attention_heads=16, num_layers=24, seq_length=2048, rotary_pct=0.25,
                              gpt_j_residual=True, bias_gelu_fusion=True, scaled_upper_triang_masked_softmax_fusion=True,
                              add_adapters=True, adapter_downsample_factor=8, freeze_lm=True, micro_batch=4,
                              padded_vocab_size=50304, encoder_out_dim=1024, prefix_len=1, image_embed_dropout_prob=0.1, lr=8e-4, weight_decay=0.0),
    # configs/19M.yml:6-17 (text-only plumbing config)
    "19M": dict(hidden_size=512, num_attention_heads=8, num_layers=6, seq_length=2048, rotary_pct=1.0,
                gpt_j_residual=False, bias_gelu_fusion=False, scaled_upper_triang_masked_softmax_fusion=False,
                add_adapters=False, adapter_downsample_factor=8, freeze_lm=False, micro_batch=4,
                padded_vocab_size=50304, encoder_out_dim=0, prefix_len=0, image_embed_dropout_prob=0.0, lr=1e-3, weight_decay=0.1),
    # configs/summit-70m-openclipH.yml:16-33 (no adapter_downsample_factor key: the default, 4 - neox_args.py:98)
    "summit-70m-openclipH": dict(hidden_size=512, num_attention_heads=8, num_layers=6, seq_length=2048, rotary_pct=0.25,
                                 gpt_j_residual=True, bias_gelu_fusion=False, scaled_upper_triang_masked_softmax_fusion=False,
                                 add_adapters=True, adapter_downsample_factor=4, freeze_lm=True, micro_batch=4,
                                 padded_vocab_size=50304, encoder_out_dim=1024, prefix_len=1, image_embed_dropout_prob=0.1, lr=1e-3, weight_decay=0.1),
    # configs/summit-1-3B.yml:5-33 - BASELINE.json configs[3] runs it at tensor-parallel 2 x data-parallel 4 (the YAML's own
    # model-parallel-size 3 / pipe 2 is a Summit layout; 18 heads % 2 == 0 and 2304 % 2 == 0 satisfy transformer.py:210-217).
    # No adapters, LM trainable, image prefix on (openclip-H -> 1 prefix token)
    "summit-1-3B": dict(hidden_size=2304, num_attention_heads=18, num_layers=24, seq_length=2048, rotary_pct=1.0,
                        gpt_j_residual=False, bias_gelu_fusion=False, scaled_upper_triang_masked_softmax_fusion=False,
                        add_adapters=False, adapter_downsample_factor=8, freeze_lm=False, micro_batch=4,
                        padded_vocab_size=50304, encoder_out_dim=1024, prefix_len=1, image_embed_dropout_prob=0.1, lr=2e-4, weight_decay=0.1),
    # configs/6-7B.yml:5-22 shape as a multimodal model (BASELINE.json configs[4]: tensor-parallel 8): image prefix as in the
    # other multimodal configs, LM trainable
    "6-7B": dict(hidden_size=4096, num_attention_heads=32, num_layers=32, seq_length=2048, rotary_pct=1.0,
                 gpt_j_residual=False, bias_gelu_fusion=False, scaled_upper_triang_masked_softmax_fusion=False,
                 add_adapters=False, adapter_downsample_factor=8, freeze_lm=False, micro_batch=4,
                 padded_vocab_size=50304, encoder_out_dim=1024, prefix_len=1, image_embed_dropout_prob=0.1, lr=1.2e-4, weight_decay=0.1),
    # a small MAGMA stack with everything the 410M recipe has (adapters of a shape the fused adapter kernel takes, frozen LM,
    # image prefix with dropout): what tests/test_graph_gpu.py captures
    "tiny_magma": dict(hidden_size=512, num_attention_heads=8, num_layers=2, seq_length=256, rotary_pct=0.25,
                       gpt_j_residual=True, bias_gelu_fusion=True, scaled_upper_triang_masked_softmax_fusion=True,
                       add_adapters=True, adapter_downsample_factor=8, freeze_lm=True, micro_batch=2,
                       padded_vocab_size=1024, encoder_out_dim=64, prefix_len=1, image_embed_dropout_prob=0.1),
    # the summit-1-3B recipe (head size 128, sequential residual, erf GeLU, full rotary, everything trainable) at test size
    "tiny_1_3B": dict(hidden_size=256, num_attention_heads=2, num_layers=2, seq_length=128, rotary_pct=1.0,
                      gpt_j_residual=False, bias_gelu_fusion=False, scaled_upper_triang_masked_softmax_fusion=False,
                      add_adapters=False, adapter_downsample_factor=8, freeze_lm=False, micro_batch=4,
                      padded_vocab_size=1024, encoder_out_dim=64, prefix_len=1, image_embed_dropout_prob=0.0),
    "tiny": dict(hidden_size=256, num_attention_heads=4, num_layers=2, seq_length=256, rotary_pct=0.25,
                 gpt_j_residual=True, bias_gelu_fusion=True, scaled_upper_triang_masked_softmax_fusion=True,
                 add_adapters=False, adapter_downsample_factor=8, freeze_lm=False, micro_batch=2,
                 padded_vocab_size=1024, encoder_out_dim=64, prefix_len=1, image_embed_dropout_prob=0.0),
}


def neox_args_for(cfg, dtype=torch.bfloat16, tp=1):
    """cfg["use_cpu_initialization"] (default False): build every sharded weight from its full master on the host and keep this
    rank's slices (mpu/layers.py:53-93) - the same seed then gives the same model at every tensor-parallel degree."""
    return SimpleNamespace(
        hidden_size=cfg["hidden_size"], num_attention_heads=cfg["num_attention_heads"], num_layers=cfg["num_layers"],
        seq_length=cfg["seq_length"], precision={torch.bfloat16: "bfloat16", torch.float16: "fp16"}[dtype],
        params_dtype=dtype, pos_emb="rotary", rotary_pct=cfg["rotary_pct"], rotary_emb_base=10000, norm="layernorm",
        layernorm_epsilon=1e-5, activation="gelu", bias_gelu_fusion=cfg["bias_gelu_fusion"], bias_dropout_fusion=False,
        scaled_upper_triang_masked_softmax_fusion=cfg["scaled_upper_triang_masked_softmax_fusion"],
        scaled_masked_softmax_fusion=False, apply_query_key_layer_scaling=False, attention_softmax_in_fp32=False,
        attention_dropout=0.0, hidden_dropout=0.0, gpt_j_residual=cfg["gpt_j_residual"], gpt_j_tied=False,
        attention_config=["global"] * cfg["num_layers"], use_mup=False,
        use_cpu_initialization=bool(cfg.get("use_cpu_initialization", False)),
        model_parallel_size=tp, fp32_allreduce=False, onnx_safe=False)


def small_init(dim):
    """init_functions.py:138-152."""
    std = (2 / (5 * dim)) ** 0.5
    return lambda t: torch.nn.init.normal_(t, mean=0.0, std=std)


def wang_init(n_layers, dim):
    """init_functions.py:155-167."""
    std = 2 / n_layers / dim ** 0.5
    return lambda t: torch.nn.init.normal_(t, mean=0.0, std=std)


class ImagePrefixProj(nn.Module):
    """ImagePrefix minus the encoder (image_prefix.py:207-252): proj -> dropout -> LayerNorm -> [b, P, h]."""

    def __init__(self, enc_dim, out_dim, dropout_p, dtype):
        super().__init__()
        self.proj = nn.Linear(enc_dim, out_dim)
        self.ln = LayerNorm(out_dim)
        self.dropout_p = dropout_p
        self.to(dtype)

    def forward(self, enc_out):  # [b, enc_dim] (pooled) or [b, P, enc_dim]
        state, pos = None, 0
        drop = self.training and self.dropout_p > 0
        if drop:
            state, pos = mpu.get_cuda_rng_tracker().draw_device(enc_out.device)  # fresh mask at every HIP-graph replay too
        return Fn.image_prefix_tail(enc_out, self.proj.weight, self.proj.bias, self.ln.weight, self.ln.bias, self.ln.eps,
                                    self.dropout_p, drop, state, pos)


class MagmaStack(nn.Module):
    def __init__(self, cfg, dtype=torch.bfloat16, device="cuda"):
        super().__init__()
        self.cfg = cfg
        args = neox_args_for(cfg, dtype, tp=mpu.get_model_parallel_world_size())
        self.args = args
        h, L, V = cfg["hidden_size"], cfg["num_layers"], cfg["padded_vocab_size"]
        init, out_init = small_init(h), wang_init(L, h)
        self.word_embeddings = nn.Embedding(V, h, dtype=dtype, device=device)
        init(self.word_embeddings.weight.data)
        self.image_prefix = ImagePrefixProj(cfg["encoder_out_dim"], h, cfg["image_embed_dropout_prob"], dtype) \
            if cfg["prefix_len"] > 0 else None
        self.layers = nn.ModuleList([
            ParallelTransformerLayer(neox_args=args, attention_mask_func=gpt2_attention_mask_func, init_method=init,
                                     output_layer_init_method=out_init, layer_number=i, rotary=True) for i in range(L)])
        self.final_norm = LayerNorm(h, eps=args.layernorm_epsilon)
        self.final_linear = mpu.ColumnParallelLinear(neox_args=args, input_size=h, output_size=V, bias=False,
                                                     init_method=init, gather_output=False)
        self.to(device=device, dtype=dtype)
        if cfg["add_adapters"]:
            add_adapters(args, self.layers, downsample_factor=cfg["adapter_downsample_factor"], location="mlp")
            add_adapters(args, self.layers, downsample_factor=cfg["adapter_downsample_factor"], location="attention")
        if cfg["freeze_lm"]:  # training.py:454-459
            for name, p in self.named_parameters():
                p.requires_grad = ("adapter" in name) or ("image_prefix" in name)
        self._mask = None

    def trainable_parameters(self):
        return [p for p in self.parameters() if p.requires_grad]

    def attention_mask(self, s, device):
        if self._mask is None or self._mask.size(-1) != s:
            self._mask = torch.triu(torch.ones(s, s, dtype=torch.bool, device=device), 1).view(1, 1, s, s)
        return self._mask

    def _run_layers(self, start, stop):
        def run(x, mask):
            for layer in self.layers[start:stop]:
                x = layer(x, mask)
            return x
        return run

    def _head_loss_parts(self, x, labels, loss_mask):
        """(sum of masked token losses, number of unmasked tokens) of one (sub-)batch: final LayerNorm, LM head
        (transformer.py:816-839) and the vocabulary-parallel loss over the TP group (gpt2_model.py:52-70 ->
        mpu/cross_entropy.py:27-117: three [b, s] exchanges; each rank's softmax covers its own vocabulary partition)."""
        xn = Fn.final_layer_norm_sb(x, self.final_norm.weight, self.final_norm.bias, self.final_norm.eps)  # [b, s, h], no copy
        logits, _ = self.final_linear(xn)  # [b, s, V/p]: gather_output=False (transformer.py:151-161)
        losses = mpu.vocab_parallel_cross_entropy(logits, labels)
        lm = loss_mask.reshape(-1).float()
        return torch.sum(losses.reshape(-1) * lm), lm.sum()

    def forward(self, enc_out, tokens, labels, loss_mask):
        """enc_out [b, P, enc_dim] | None, tokens [b, s] int64, labels [b, s], loss_mask [b, s] float -> scalar loss."""
        b, s = tokens.shape
        pre = self.image_prefix(enc_out) if self.image_prefix is not None else None
        # gather + [prefix | text[:, :s-P]] + [b,s,h] -> [s,b,h] in one kernel
        x = Fn.embed_prefix(pre, self.word_embeddings.weight, tokens)
        mask = self.attention_mask(s, x.device)
        tp = mpu.get_model_parallel_world_size()
        chunks = int(self.cfg.get("tp_overlap_chunks", 2 if tp > 1 else 1))
        recompute = self.cfg.get("checkpoint_activations", False)
        if tp > 1 and chunks > 1 and b % chunks == 0 and not recompute:
            # tensor parallel: sub-batches walk the layers in lock step on separate streams, so that each sub-batch's
            # all-reduces overlap the other's GEMMs / attention (mpu/overlap.py); same activations as the unsplit run
            xs = mpu.split_batch(x, chunks, dim=1)
            lbl, lms = torch.chunk(labels, chunks, dim=0), torch.chunk(loss_mask, chunks, dim=0)
            stages = [(lambda t, layer=layer: layer(t, mask)) for layer in self.layers]
            parts = mpu.run_interleaved(stages, xs, finish=lambda i, t: self._head_loss_parts(t, lbl[i], lms[i]))
            return sum(p[0] for p in parts) / sum(p[1] for p in parts)
        for start, stop, recompute in checkpoint_policy(len(self.layers), recompute, self.cfg.get("checkpoint_num_layers", 1)):
            if recompute and torch.is_grad_enabled():  # gpt2_model.py:123-132
                if not x.requires_grad:
                    x = x.detach().requires_grad_(True)  # the re-entrant recompute needs a differentiable input
                x = mpu.checkpoint(self._run_layers(start, stop), x, mask)
            else:
                x = self._run_layers(start, stop)(x, mask)
        if tp > 1:
            total, count = self._head_loss_parts(x, labels, loss_mask)
            return total / count
        w = self.final_linear.weight
        fn = self.final_norm
        if self.cfg.get("fused_head_loss", True) and torch.is_grad_enabled() and Fn.lm_head_loss_supported(x, w):
            # transpose + final norm + head + loss as one node, logits read once, no [b, s, h] copy
            return Fn.final_norm_head_loss(x, fn.weight, fn.bias, fn.eps, w, labels, loss_mask)
        xn = Fn.final_layer_norm_sb(x, fn.weight, fn.bias, fn.eps)
        logits, _ = self.final_linear(xn)
        return Fn.masked_cross_entropy(logits, labels, loss_mask)


def end_of_step():
    """Call once per training step after the backward: moves the device-resident dropout counters past this step's draws
    (one capturable launch), so that the next step - or the next replay of a captured step - draws fresh masks."""
    mpu.get_cuda_rng_tracker().commit()


def synthetic_batch(cfg, device, seed, dtype=torch.bfloat16):
    """mytests/test_model_forward.py:33-41 / training.py:346-368 shapes: pooled encoder output stands in for the image."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    b, s = cfg["micro_batch"], cfg["seq_length"]
    tokens = torch.randint(0, 50277 if cfg["padded_vocab_size"] > 50277 else cfg["padded_vocab_size"], (b, s), generator=g)
    loss_mask = torch.ones(b, s)
    tail = torch.randint(1, 64, (b,), generator=g)
    for i in range(b):
        loss_mask[i, s - int(tail[i]):] = 0.0  # padded caption tail (utils.py:96-101)
    enc = None
    if cfg["prefix_len"] > 0:
        enc = torch.randn(b, cfg["prefix_len"], cfg["encoder_out_dim"], generator=g).to(dtype).to(device)
    return enc, tokens.to(device), tokens.to(device), loss_mask.to(device)


def model_flops_per_step(cfg, causal_exact=False):
    """Useful FLOPs of one micro-batch step under the BASELINE.md section 4 work model (full s x s attention counting,
    recompute not counted; frozen LM => backward = 1x linear + 2x attention; adapters trainable => 3x).
    causal_exact: count only the unmasked half of the attention products (what the causal kernels execute)."""
    h, L, s, b, V = cfg["hidden_size"], cfg["num_layers"], cfg["seq_length"], cfg["micro_batch"], cfg["padded_vocab_size"]
    T = s * b
    f_lin = 24 * T * h * h
    f_attn = 4 * T * s * h // (2 if causal_exact else 1)
    f_ad = 8 * T * h * (h // cfg["adapter_downsample_factor"]) if cfg["add_adapters"] else 0
    f_head = 2 * T * h * V
    if cfg["freeze_lm"]:
        layer = (f_lin + f_attn) + (f_lin + 2 * f_attn) + 3 * f_ad
        head = 2 * f_head
    else:
        layer = 3 * (f_lin + f_attn) + 3 * f_ad
        head = 3 * f_head
    return L * layer + head
