"""DeepSpeed-format checkpoint loader + recompute policy: host logic, runs without a GPU.

No checkpoint written by the reference exists in this environment, so these tests write the layout
(megatron/checkpointing.py:165-205, DeepSpeed PipelineModule file names; tools/convert_sequential_to_hf.py:43-69 for the
sequential variant) themselves and check the reference's merge rules (convert_sequential_to_hf.py:169-275)."""
import os

import pytest
import torch

from multimodal_amd import checkpointing as ck
from multimodal_amd import harness, mpu


@pytest.fixture
def single_rank():
    mpu.destroy_model_parallel()
    mpu.initialize_model_parallel(1)
    yield
    mpu.destroy_model_parallel()


def tiny_cfg(**over):
    cfg = dict(harness.CONFIGS["tiny"])
    cfg.update(over)
    return cfg


def build(cfg, seed):
    torch.manual_seed(seed)
    return harness.MagmaStack(cfg, dtype=torch.bfloat16, device="cpu")


def layer_state(h=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).bfloat16()
    return {"input_layernorm.weight": r(h), "input_layernorm.bias": r(h),
            "attention.query_key_value.weight": r(3 * h, h), "attention.query_key_value.bias": r(3 * h),
            "attention.rotary_emb.inv_freq": torch.rand(8, generator=g),
            "attention.dense.weight": r(h, h), "attention.dense.bias": r(h),
            "mlp.dense_h_to_4h.weight": r(4 * h, h), "mlp.dense_h_to_4h.bias": r(4 * h),
            "mlp.dense_4h_to_h.weight": r(h, 4 * h), "mlp.dense_4h_to_h.bias": r(h),
            "mlp.adapter.0.weight": r(h // 8, h), "mlp.adapter.2.bias": r(h)}


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("gptj", [True, False])
def test_split_then_merge_is_identity(world, gptj):
    full = layer_state()
    parts = [ck.split_partition(full, r, world, gptj) for r in range(world)]
    # shapes a rank holds (mpu/layers.py:549-564,745: column = rows / p, row = columns / p)
    assert parts[0]["attention.query_key_value.weight"].shape == (3 * 64 // world, 64)
    assert parts[0]["attention.dense.weight"].shape == (64, 64 // world)
    assert parts[0]["mlp.dense_4h_to_h.weight"].shape == (64, 4 * 64 // world)
    assert parts[0]["mlp.dense_h_to_4h.bias"].shape == (4 * 64 // world,)
    assert parts[1]["input_layernorm.weight"].shape == (64,)
    merged = ck.merge_partitions(parts, gptj)
    assert merged.keys() == full.keys()
    for k in full:
        assert torch.equal(merged[k], full[k]), k  # power-of-two split of a bf16 bias and its sum are exact


def test_merge_rules_follow_the_converter():
    """convert_sequential_to_hf.py:188-228: cat dim 1 (row weights), cat dim 0 (column), mean (norms), sum (row biases)."""
    a, b = layer_state(seed=1), layer_state(seed=2)
    m = ck.merge_partitions([a, b], gpt_j_residual=True)
    assert torch.equal(m["attention.dense.weight"], torch.cat([a["attention.dense.weight"], b["attention.dense.weight"]], 1))
    assert torch.equal(m["mlp.dense_h_to_4h.weight"], torch.cat([a["mlp.dense_h_to_4h.weight"], b["mlp.dense_h_to_4h.weight"]], 0))
    assert torch.equal(m["input_layernorm.weight"],
                       ((a["input_layernorm.weight"].float() + b["input_layernorm.weight"].float()) / 2).bfloat16())
    assert torch.equal(m["attention.dense.bias"], (a["attention.dense.bias"].float() + b["attention.dense.bias"].float()).bfloat16())
    assert torch.equal(m["attention.rotary_emb.inv_freq"], a["attention.rotary_emb.inv_freq"])
    # without the gpt-j single reduce the bias is added once after the all-reduce: the copies are one bias
    m2 = ck.merge_partitions([a, a], gpt_j_residual=False)
    assert torch.equal(m2["attention.dense.bias"], a["attention.dense.bias"])


def test_save_load_round_trip(tmp_path, single_rank):
    cfg = tiny_cfg(add_adapters=True, freeze_lm=True)
    src, dst = build(cfg, 1), build(cfg, 2)
    assert not torch.equal(src.layers[0].mlp.attn_block.dense_h_to_4h.weight, dst.layers[0].mlp.attn_block.dense_h_to_4h.weight)
    d = ck.save_checkpoint(src, str(tmp_path), 1234)
    L = cfg["num_layers"]
    names = sorted(os.listdir(d))
    assert names == sorted(["layer_%02d-model_00-model_states.pt" % i for i in [0] + list(range(2, L + 2)) + [L + 3, L + 4]]
                           + ["mp_rank_00_model_states.pt"])
    assert open(os.path.join(str(tmp_path), "latest")).read() == "global_step1234"
    emb = torch.load(os.path.join(d, "layer_00-model_00-model_states.pt"), weights_only=True)
    assert set(emb) == {"word_embeddings.weight", "image_prefix.proj.weight", "image_prefix.proj.bias",
                        "image_prefix.ln.weight", "image_prefix.ln.bias"}
    it, rep = ck.load_checkpoint(dst, str(tmp_path), strict=True)
    assert it == 1234 and rep["tag"] == "global_step1234" and not rep["missing"] and not rep["unexpected"]
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    assert ck.load_checkpoint(dst, str(tmp_path), finetune=True)[0] == 0  # checkpointing.py:269-270
    with pytest.raises(ValueError, match="Available iterations: \\[1234\\]"):
        ck.load_checkpoint(dst, str(tmp_path), iteration=7)  # checkpointing.py:248-260
    assert ck.load_checkpoint(dst, str(tmp_path / "nothing_here"))[0] == 0  # :261-264


def test_lm_checkpoint_into_adapter_model_and_foreign_tensors(tmp_path, single_rank):
    """A language-model checkpoint (no adapters; image encoder tensors present) loads under the wrapped names."""
    lm = build(tiny_cfg(add_adapters=False), 3)
    d = ck.save_checkpoint(lm, str(tmp_path), 10)
    f0 = os.path.join(d, "layer_00-model_00-model_states.pt")
    sd0 = torch.load(f0, weights_only=True)
    sd0["image_prefix.enc.visual.conv1.weight"] = torch.zeros(4, 3, 2, 2)  # the encoder is outside the hot path
    torch.save(sd0, f0)
    model = build(tiny_cfg(add_adapters=True, freeze_lm=True), 4)
    before = model.layers[1].mlp.adapter[0].weight.clone()
    with pytest.raises(RuntimeError, match="strict"):
        ck.load_checkpoint(model, str(tmp_path), strict=True)
    it, rep = ck.load_checkpoint(model, str(tmp_path))
    assert it == 10
    assert rep["unexpected"] == ["layer_00:image_prefix.enc.visual.conv1.weight"]
    assert sorted(rep["missing"])[0] == "layer_02:attention.adapter.0.bias" and len(rep["missing"]) == 8 * len(model.layers)
    assert torch.equal(model.layers[1].mlp.attn_block.dense_4h_to_h.weight, lm.layers[1].mlp.dense_4h_to_h.weight)
    assert torch.equal(model.layers[0].attention.attn_block.query_key_value.bias, lm.layers[0].attention.query_key_value.bias)
    assert torch.equal(model.layers[1].mlp.adapter[0].weight, before)  # fresh adapters untouched


def test_two_partition_checkpoint_merges_into_one_rank(tmp_path, single_rank):
    """Checkpoint written at tensor-parallel degree 2 (pipeline layout) and the same tensors in the sequential layout."""
    cfg = tiny_cfg()
    src, dst, dst2 = build(cfg, 5), build(cfg, 6), build(cfg, 7)
    one = ck.save_checkpoint(src, str(tmp_path / "tp1"), 3)
    two = tmp_path / "tp2" / "global_step3"
    seq = tmp_path / "seq" / "global_step3"
    os.makedirs(two), os.makedirs(seq)
    modules = [{}, {}]
    for name in os.listdir(one):
        if not name.startswith("layer_"):
            continue
        idx = int(name[6:8])
        full = torch.load(os.path.join(one, name), weights_only=True)
        for r in range(2):
            part = ck.split_partition(full, r, 2, gpt_j_residual=True)
            torch.save(part, ck.layer_file(str(two), idx, r))
            modules[r].update({"sequential.%d.%s" % (idx, k): v for k, v in part.items()})
    for r in range(2):
        torch.save({"iteration": 3, "module": None}, ck.client_file(str(two), r))
        torch.save({"iteration": 3, "module": modules[r]}, ck.client_file(str(seq), r))
    for root in ("tp2", "seq"):
        with open(str(tmp_path / root / "latest"), "w") as f:
            f.write("global_step3")
    assert ck.count_partitions(str(two)) == 2 and ck.count_partitions(str(seq)) == 2
    for model, root in ((dst, "tp2"), (dst2, "seq")):
        it, rep = ck.load_checkpoint(model, str(tmp_path / root), strict=True)
        assert it == 3
        for (k, a), (_, b) in zip(src.state_dict().items(), model.state_dict().items()):
            assert torch.equal(a, b), (root, k)


def test_load_reshards_to_the_running_tensor_parallel_rank(tmp_path, single_rank):
    cfg = tiny_cfg()
    src = build(cfg, 8)
    ck.save_checkpoint(src, str(tmp_path), 1)
    mpu.set_model_parallel_world_size(2)
    mpu.set_model_parallel_rank(1)
    shard = build(cfg, 9)  # a rank-1-of-2 model: column layers hold half the rows
    assert shard.layers[0].mlp.dense_h_to_4h.weight.shape == (2 * cfg["hidden_size"], cfg["hidden_size"])
    it, rep = ck.load_checkpoint(shard, str(tmp_path))
    assert it == 1 and not rep["missing"]
    h = cfg["hidden_size"]
    assert torch.equal(shard.layers[0].mlp.dense_h_to_4h.weight, src.layers[0].mlp.dense_h_to_4h.weight[2 * h:])
    assert torch.equal(shard.layers[1].attention.dense.weight, src.layers[1].attention.dense.weight[:, h // 2:])
    assert torch.equal(shard.layers[1].attention.dense.bias, (src.layers[1].attention.dense.bias.float() / 2).bfloat16())
    assert torch.equal(shard.final_linear.weight, src.final_linear.weight[cfg["padded_vocab_size"] // 2:])
    assert torch.equal(shard.word_embeddings.weight, src.word_embeddings.weight)  # whole on every rank in this stack


class Opaque:  # DeepSpeed pickles argparse namespaces etc. into the client-state file; the safe loader must refuse, not run
    pass


def test_refused_client_state_falls_back_to_the_tag(tmp_path, single_rank):
    cfg = tiny_cfg()
    src = build(cfg, 10)
    d = ck.save_checkpoint(src, str(tmp_path), 77)
    torch.save({"iteration": 5, "args": Opaque()}, ck.client_file(d, 0))
    with pytest.warns(UserWarning, match="safe loader refused"):
        it, rep = ck.load_checkpoint(build(cfg, 11), str(tmp_path))
    assert it == 77 and rep["client_state"] is None


def test_checkpoint_policy():
    """gpt2_model.py:123-132: interval = checkpoint_num_layers if checkpoint_activations else 0."""
    assert ck.checkpoint_policy(6, False, 1) == [(0, 6, False)]
    assert ck.checkpoint_policy(3, True, 1) == [(0, 1, True), (1, 2, True), (2, 3, True)]
    assert ck.checkpoint_policy(5, True, 2) == [(0, 2, True), (2, 4, True), (4, 5, True)]
    assert ck.checkpoint_policy(5, True, 0) == [(0, 5, False)]
