"""DeepSpeed-format NeoX checkpoints <-> the stack in harness.py, and the activation-recompute policy.

What the reference does: `load_checkpoint` (megatron/checkpointing.py:225-329) hands the directory to DeepSpeed's
`PipelineModule`, which keeps one file per pipeline layer and tensor-parallel rank,

    <load>/latest                                              text: the tag, "global_step<N>"
    <load>/<tag>/layer_<LL>-model_<MM>-model_states.pt         flat state dict of pipeline layer LL, TP rank MM
    <load>/<tag>/mp_rank_<MM>_model_states.pt                  client state: iteration, args, rng states

with LL counting the specs of GPT2ModelPipe (gpt2_model.py:166-305): 0 = EmbeddingPipe (word_embeddings.*,
image_prefix.*), 1 = a lambda, 2 .. L+1 = the transformer layers, L+2 = a lambda, L+3 = NormPipe (norm.*),
L+4 = ParallelLinearPipe (final_linear.*).  The non-pipeline ("sequential") variant keeps the same tensors in
`mp_rank_<MM>_model_states.pt` under `module["sequential.<LL>.<key>"]` (tools/convert_sequential_to_hf.py:43-69).
Both are read here.  The tensor-parallel merge rules are the ones the reference's converter applies
(tools/convert_sequential_to_hf.py:169-275): column-parallel tensors concatenate on dim 0, row-parallel weights on
dim 1, norms average, row-parallel biases SUM (under gpt_j_residual every rank's copy is added before the single
all-reduce, transformer.py:677-678,766, so the effective bias is the sum).

Files are read with `torch.load(weights_only=True)` only: nothing in a checkpoint is executed.  A client-state file
that this loader refuses (DeepSpeed pickles arbitrary Python objects into it) costs the rng states and args check,
not the weights: the iteration then comes from the tag.

No real Pythia / MAGMA checkpoint is available in this environment (no network): the tests write checkpoints in this
layout themselves, so key names follow the reference's modules but no file it produced has been read ("parity
unpinned" for this row).
"""
import os
import re
import warnings

import torch

from . import mpu
from .model.adapter import AdapterWrapper

# how a tensor is laid out across tensor-parallel ranks, by key suffix (convert_sequential_to_hf.py:188-228,169-175,264-275)
_COLUMN = ("query_key_value.weight", "query_key_value.bias", "dense_h_to_4h.weight", "dense_h_to_4h.bias",
           "word_embeddings.weight", "final_linear.weight")
_ROW_WEIGHT = ("attention.dense.weight", "attn_block.dense.weight", "dense_4h_to_h.weight")
_ROW_BIAS = ("attention.dense.bias", "attn_block.dense.bias", "dense_4h_to_h.bias")
_TAKE_ONE = ("rotary_emb.inv_freq",)


def _kind(key):
    if key.endswith(_COLUMN):
        return "column"
    if key.endswith(_ROW_WEIGHT):
        return "row_weight"
    if key.endswith(_ROW_BIAS):
        return "row_bias"
    if key.endswith(_TAKE_ONE):
        return "one"
    return "replicated"  # norms, adapters, image-prefix projection: identical on every rank


def merge_partitions(states, gpt_j_residual=True):
    """One layer's state dicts from all tensor-parallel ranks of a checkpoint -> the unpartitioned state dict."""
    if len(states) == 1:
        return dict(states[0])
    out = {}
    for key in states[0]:
        parts = [sd[key] for sd in states]
        kind = _kind(key)
        if kind == "column":
            out[key] = torch.cat(parts, dim=0)
        elif kind == "row_weight":
            out[key] = torch.cat(parts, dim=1)
        elif kind == "row_bias":
            # gpt-j residual: every rank adds its copy before the one all-reduce -> the effective bias is the sum;
            # otherwise the bias is added once after the reduce (mpu/layers.py:745-751) and the copies are identical
            out[key] = sum(p.float() for p in parts).to(parts[0].dtype) if gpt_j_residual else parts[0].clone()
        elif kind == "one":
            out[key] = parts[0].clone()
        else:
            out[key] = (sum(p.float() for p in parts) / len(parts)).to(parts[0].dtype)
    return out


def split_partition(full, rank, world, gpt_j_residual=True):
    """Inverse of merge_partitions for one rank: the unpartitioned state dict -> what TP rank `rank` of `world` holds."""
    if world == 1:
        return dict(full)
    out = {}
    for key, t in full.items():
        kind = _kind(key)
        if kind == "column":
            out[key] = t.chunk(world, dim=0)[rank].clone()
        elif kind == "row_weight":
            out[key] = t.chunk(world, dim=1)[rank].clone()
        elif kind == "row_bias" and gpt_j_residual:
            out[key] = (t.float() / world).to(t.dtype)  # the ranks' copies sum to the effective bias
        else:
            out[key] = t.clone()
    return out


# ---------------------------------------------------------------------------------------------------- layout
def read_tag(load_dir, iteration=None):
    """megatron/checkpointing.py:235-238: a given iteration names its tag; otherwise DeepSpeed's `latest` file does."""
    if iteration is not None:
        return "global_step%d" % iteration
    path = os.path.join(load_dir, "latest")
    if not os.path.isfile(path):
        return None
    with open(path) as f:
        return f.read().strip()


def available_iterations(load_dir):
    """megatron/checkpointing.py:252-257."""
    its = []
    for name in os.listdir(load_dir) if os.path.isdir(load_dir) else []:
        m = re.fullmatch(r"global_step(\d+)", name)
        if m:
            its.append(int(m.group(1)))
    return sorted(its)


def layer_file(ckpt_dir, layer_idx, mp_rank):
    return os.path.join(ckpt_dir, "layer_%02d-model_%02d-model_states.pt" % (layer_idx, mp_rank))


def client_file(ckpt_dir, mp_rank):
    return os.path.join(ckpt_dir, "mp_rank_%02d_model_states.pt" % mp_rank)


def count_partitions(ckpt_dir):
    """Tensor-parallel degree the checkpoint was written with."""
    n = 0
    while os.path.isfile(client_file(ckpt_dir, n)) or os.path.isfile(layer_file(ckpt_dir, 0, n)):
        n += 1
    return n


def _safe_load(path):
    return torch.load(path, map_location="cpu", weights_only=True)


class _Source:
    """Per-layer state dicts of a checkpoint directory, either layout."""

    def __init__(self, ckpt_dir, partitions):
        self.dir, self.partitions = ckpt_dir, partitions
        self.pipeline = os.path.isfile(layer_file(ckpt_dir, 0, 0))
        self._sequential = None
        if not self.pipeline:
            mods = []
            for r in range(partitions):
                sd = _safe_load(client_file(ckpt_dir, r))
                if "module" not in sd or sd["module"] is None:
                    raise FileNotFoundError("%s holds neither layer_XX files nor a 'module' state dict" % ckpt_dir)
                mods.append(sd["module"])
            self._sequential = mods

    def layer(self, idx):
        """-> list over TP ranks of {key: tensor} for pipeline layer idx (empty dicts when the layer has no file)."""
        if self.pipeline:
            out = []
            for r in range(self.partitions):
                p = layer_file(self.dir, idx, r)
                out.append(_safe_load(p) if os.path.isfile(p) else {})
            return out
        prefix = "sequential.%d." % idx
        return [{k[len(prefix):]: v for k, v in m.items() if k.startswith(prefix)} for m in self._sequential]


def _module_table(model):
    """(pipeline layer index, module, key prefix inside that layer's file) for every part of the stack that the
    checkpoint covers.  The image encoder (`image_prefix.enc.*`) is outside the hot path and is not loaded."""
    L = len(model.layers)
    table = [(0, model.word_embeddings, "word_embeddings.")]
    if model.image_prefix is not None:
        table.append((0, model.image_prefix, "image_prefix."))
    table += [(i + 2, layer, "") for i, layer in enumerate(model.layers)]
    table += [(L + 3, model.final_norm, "norm."), (L + 4, model.final_linear, "final_linear.")]
    return table


def _wrap_keys(module, sd):
    """A language-model checkpoint written before the adapters were added (training.py:421-432 loads, then wraps) has
    `attention.dense.weight` where the wrapped layer has `attention.attn_block.dense.weight` (adapter.py:106-107)."""
    out = dict(sd)
    for name in ("attention", "mlp"):
        sub = getattr(module, name, None)
        if isinstance(sub, AdapterWrapper) and not any(k.startswith(name + ".attn_block.") for k in sd):
            for k in list(out):
                if k.startswith(name + ".") and not k.startswith(name + ".adapter."):
                    out[name + ".attn_block." + k[len(name) + 1:]] = out.pop(k)
    return out


def optimizer_file(ckpt_dir, mp_rank):
    return os.path.join(ckpt_dir, "mm_fused_adam_mp_rank_%02d_optim_states.pt" % mp_rank)


def load_checkpoint(model, load_dir, iteration=None, strict=False, finetune=False, optimizer=None):
    """Load the language-model / adapter / image-prefix weights of a DeepSpeed-format NeoX checkpoint into a
    `harness.MagmaStack`, resharding from the checkpoint's tensor-parallel degree to the running one.

    Returns (iteration, report): iteration is 0 when there is no checkpoint or `finetune` is set
    (checkpointing.py:248-270); report lists `missing` (model tensors the checkpoint lacks: e.g. fresh adapters),
    `unexpected` (checkpoint tensors without a home: e.g. the image encoder) and `client_state`.
    With strict=True either list being non-empty raises (the reference loads with load_module_strict=False, :245)."""
    tag = read_tag(load_dir, iteration)
    ckpt_dir = os.path.join(load_dir, tag) if tag else None
    if ckpt_dir is None or not os.path.isdir(ckpt_dir):
        if iteration is not None:
            raise ValueError("Unable to load checkpoint for iteration %d. Available iterations: %s"
                             % (iteration, available_iterations(load_dir)))
        return 0, dict(missing=[], unexpected=[], client_state=None, tag=None)
    parts = count_partitions(ckpt_dir)
    if parts == 0:
        raise FileNotFoundError("no mp_rank_XX_model_states.pt / layer_XX files under %s" % ckpt_dir)
    world, rank = mpu.get_model_parallel_world_size(), mpu.get_model_parallel_rank()
    gptj = bool(model.args.gpt_j_residual)
    src = _Source(ckpt_dir, parts)
    missing, unexpected, used = [], [], {}
    for idx, module, prefix in _module_table(model):
        if idx not in used:
            full = merge_partitions(src.layer(idx), gptj)
            used[idx] = (split_partition(full, rank, world, gptj), set(), full)
        layer_sd, taken, full = used[idx]
        sub = {k[len(prefix):]: v for k, v in layer_sd.items() if k.startswith(prefix)}
        taken.update(prefix + k for k in sub)
        sub = _wrap_keys(module, sub)
        own = module.state_dict()
        for k, v in list(sub.items()):
            if k in own and own[k].shape != v.shape:
                whole = _wrap_keys(module, {k2[len(prefix):]: v2 for k2, v2 in full.items() if k2.startswith(prefix)})[k]
                if own[k].shape == whole.shape:  # kept whole on every rank here (the stack's word embedding is not
                    sub[k] = whole               # vocab-parallel, harness.py; the reference's is, mpu/layers.py:184-210)
                    continue
                raise ValueError("layer %d key %s%s: checkpoint shape %s, model shape %s (after resharding %d -> %d)"
                                 % (idx, prefix, k, tuple(v.shape), tuple(own[k].shape), parts, world))
        res = module.load_state_dict(sub, strict=False)
        missing += ["layer_%02d:%s%s" % (idx, prefix, k) for k in res.missing_keys]
        unexpected += ["layer_%02d:%s%s" % (idx, prefix, k) for k in res.unexpected_keys]
    for idx, (layer_sd, taken, _) in used.items():
        unexpected += ["layer_%02d:%s" % (idx, k) for k in layer_sd if k not in taken]
    if strict and (missing or unexpected):
        raise RuntimeError("load_checkpoint(strict=True): missing %s, unexpected %s" % (missing, unexpected))
    client = None
    if os.path.isfile(client_file(ckpt_dir, min(rank, parts - 1))):
        try:
            client = {k: v for k, v in _safe_load(client_file(ckpt_dir, min(rank, parts - 1))).items() if k != "module"}
        except Exception as e:  # pickled Python objects: refused, not executed
            warnings.warn("client state of %s not read (safe loader refused it: %s); iteration taken from the tag"
                          % (ckpt_dir, str(e).splitlines()[0]))
    if finetune:
        it = 0
    else:
        it = (client or {}).get("iteration") or (client or {}).get("total_iters")
        if it is None:
            m = re.fullmatch(r"global_step(\d+)", tag)
            if m is None:
                raise ValueError("Unable to load iteration from checkpoint %s" % ckpt_dir)
            it = int(m.group(1))
    opt_loaded = False
    if optimizer is not None:
        # The reference builds the optimizer BEFORE load_checkpoint (megatron/training.py:646,690) and lets the DeepSpeed
        # engine restore / refresh its fp32 masters.  Here: restore FusedAdam's state when this library saved one next to the
        # layer files (same TP degree only: the flat state is per-rank), else re-derive the fp32 master copy from the weights
        # just loaded - without either the first opt.step() would write the stale master back over them.
        of = optimizer_file(ckpt_dir, rank)
        if not finetune and parts == world and os.path.isfile(of):
            optimizer.load_state_dict(_safe_load(of))
            opt_loaded = True
        else:
            optimizer.refresh_master_from_params()
            if not finetune:
                # resuming (not fine-tuning) without this library's optimizer file: the Adam moments and step count start
                # from zero.  The reference restores them from its DeepSpeed ZeRO files (megatron/checkpointing.py:231-242),
                # whose format only the DeepSpeed engine reads - parity unpinned, so say so loudly instead of silently resetting.
                warnings.warn("load_checkpoint: no optimizer state restored from %s (%s); Adam moments and step restart from zero, "
                              "fp32 master weights were refreshed from the loaded parameters"
                              % (ckpt_dir, "tensor-parallel degree differs: %d -> %d" % (parts, world) if parts != world
                                 else "no %s" % os.path.basename(of)))
    return int(it), dict(missing=missing, unexpected=unexpected, client_state=client, tag=tag, optimizer_state_loaded=opt_loaded)


def save_checkpoint(model, save_dir, iteration, client_state=None, optimizer=None):
    """Write this rank's tensor-parallel partition in the pipeline layout above (checkpointing.py:165-205; one caller
    per TP rank, data-parallel rank 0 only) and point `latest` at it."""
    tag = "global_step%d" % iteration
    ckpt_dir = os.path.join(save_dir, tag)
    os.makedirs(ckpt_dir, exist_ok=True)
    rank = mpu.get_model_parallel_rank()
    per_layer = {}
    for idx, module, prefix in _module_table(model):
        sd = per_layer.setdefault(idx, {})
        for k, v in module.state_dict().items():
            sd[prefix + k] = v.detach().to("cpu").clone()
    for idx, sd in per_layer.items():
        torch.save(sd, layer_file(ckpt_dir, idx, rank))
    state = dict(client_state or {})
    state["iteration"] = int(iteration)
    state["module"] = None  # PipelineModule keeps the weights in the layer files
    torch.save(state, client_file(ckpt_dir, rank))
    if optimizer is not None:  # step, fp32 master, moments (plain tensors and numbers: readable with weights_only=True)
        sd = optimizer.state_dict()
        sd["master"], sd["exp_avg"], sd["exp_avg_sq"] = (sd[k].to("cpu") for k in ("master", "exp_avg", "exp_avg_sq"))
        torch.save(sd, optimizer_file(ckpt_dir, rank))
    if rank == 0:
        with open(os.path.join(save_dir, "latest"), "w") as f:
            f.write(tag)
    return ckpt_dir


# ---------------------------------------------------------------------------------------------------- recompute policy
def checkpoint_policy(num_layers, checkpoint_activations, checkpoint_num_layers=1):
    """Which runs of layers execute under activation recompute (gpt2_model.py:123-132: DeepSpeed's
    `activation_checkpoint_interval = checkpoint_num_layers if checkpoint_activations else 0`; every run of that
    many consecutive transformer layers is one recompute segment).  Returns a list of (start, stop, recompute).

    The HIP path saves ~0.3 GB per layer at the 410M shapes (no s x s tensor survives the forward, DESIGN.md
    section 4), so at 288 GB the fastest policy is `checkpoint_activations=False`; the flag is honoured for models
    whose saved activations would not fit."""
    if not checkpoint_activations or checkpoint_num_layers <= 0:
        return [(0, num_layers, False)]
    return [(s, min(s + checkpoint_num_layers, num_layers), True) for s in range(0, num_layers, checkpoint_num_layers)]
