#!/usr/bin/env python3
"""Writes tests/golden/configs.json: the shape / flag keys of the reference's YAML configs that BASELINE.json names, read with
yaml.safe_load from /root/reference/configs (data, not code), with the reference's documented defaults filled in
(megatron/neox_arguments/neox_args.py:98 adapter_downsample_factor = 4, :296 rotary_pct = 1.0, :572 image_embed_dropout_prob = 0.1).
tests/test_configs_cpu.py checks multimodal_amd.harness.CONFIGS against it.  Run in the build container only."""
import json
import os

import yaml

REF = "/root/reference/configs"
HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["19M", "magma_pythia_410M", "summit-70m-openclipH", "summit-1-3B", "6-7B"]
DEFAULTS = {"rotary_pct": 1.0, "adapter_downsample_factor": 4, "image_embed_dropout_prob": 0.1, "add_adapters": False,
            "freeze_lm": False, "gpt_j_residual": False, "bias_gelu_fusion": False,
            "scaled_upper_triang_masked_softmax_fusion": False, "hidden_dropout": 0.0, "attention_dropout": 0.0,
            "encoder_name": None}
KEYS = ["hidden_size", "num_attention_heads", "num_layers", "seq_length", "pos_emb", "model_parallel_size",
        "pipe_parallel_size"] + list(DEFAULTS)

out = {}
setup = {k.replace("-", "_"): v for k, v in yaml.safe_load(open(os.path.join(REF, "magma_setup.yml"))).items()}
for n in NAMES:
    d = {k.replace("-", "_"): v for k, v in yaml.safe_load(open(os.path.join(REF, n + ".yml"))).items()}
    row = {k: d.get(k, DEFAULTS.get(k)) for k in KEYS}
    row["train_micro_batch_size_per_gpu"] = d.get("train_micro_batch_size_per_gpu", setup["train_micro_batch_size_per_gpu"])
    opt = d.get("optimizer", setup["optimizer"])  # magma_pythia_410M.yml has none of its own: magma_setup.yml:38-45
    row["optimizer"] = {"type": opt["type"], "lr": opt["params"]["lr"], "betas": opt["params"]["betas"], "eps": opt["params"]["eps"]}
    row["weight_decay"] = d.get("weight_decay", 0.0)
    row["gradient_clipping"] = d.get("gradient_clipping", 1.0)
    out[n] = row
with open(os.path.join(HERE, "configs.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
