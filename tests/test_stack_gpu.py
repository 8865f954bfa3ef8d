"""The stack around the layers on the GPU: checkpoint round trip and the activation-recompute policy."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stack(cuda_device, seed, **over):
    from multimodal_amd import harness, mpu
    mpu.model_parallel_cuda_manual_seed(seed)  # the GPU initialisers draw from the tracker's model-parallel stream
    cfg = dict(harness.CONFIGS["tiny"])
    cfg.update(add_adapters=True, freeze_lm=True, image_embed_dropout_prob=0.1)
    cfg.update(over)
    torch.manual_seed(seed)
    model = harness.MagmaStack(cfg, dtype=torch.bfloat16, device=cuda_device)
    model.train()
    return cfg, model


def _loss_and_grads(model, batch, seed):
    from multimodal_amd import mpu
    mpu.model_parallel_cuda_manual_seed(seed)  # same dropout counters for every run compared
    for p in model.parameters():
        p.grad = None
    loss = model(*batch)
    loss.backward()
    return loss.detach().clone(), [p.grad.clone() for p in model.trainable_parameters()]


def test_checkpoint_round_trip_reproduces_the_step(cuda_device, tmp_path):
    """Weights saved in the DeepSpeed pipeline layout (checkpointing.py:165-205) and loaded into a differently
    initialised stack (checkpointing.py:225-329) give the same loss and gradients bit for bit."""
    from multimodal_amd import checkpointing as ck, harness, mpu
    mpu.destroy_model_parallel()
    mpu.initialize_model_parallel(1)
    cfg, a = _stack(cuda_device, 11)
    _, b = _stack(cuda_device, 12)
    batch = harness.synthetic_batch(cfg, cuda_device, seed=4)
    la, ga = _loss_and_grads(a, batch, 21)
    lb, _ = _loss_and_grads(b, batch, 21)
    assert not torch.equal(la, lb)
    ck.save_checkpoint(a, str(tmp_path), 5)
    it, rep = ck.load_checkpoint(b, str(tmp_path), strict=True)
    assert it == 5 and all(p.is_cuda for p in b.parameters())
    lb, gb = _loss_and_grads(b, batch, 21)
    assert torch.equal(la, lb)
    for x, y in zip(ga, gb):
        assert torch.equal(x, y)


@pytest.mark.parametrize("interval", [1, 2])
def test_activation_recompute_is_bit_identical(cuda_device, interval):
    """gpt2_model.py:123-132: recomputing the layers in the backward replays the same kernels on the same inputs (and the
    same dropout counters, mpu/random.py), so loss and gradients equal the run that kept its activations."""
    from multimodal_amd import harness, mpu
    mpu.destroy_model_parallel()
    mpu.initialize_model_parallel(1)
    cfg, plain = _stack(cuda_device, 13, num_layers=3)
    _, recomp = _stack(cuda_device, 13, num_layers=3, checkpoint_activations=True, checkpoint_num_layers=interval)
    batch = harness.synthetic_batch(cfg, cuda_device, seed=6)
    l0, g0 = _loss_and_grads(plain, batch, 33)
    l1, g1 = _loss_and_grads(recomp, batch, 33)
    assert torch.equal(l0, l1)
    assert len(g0) == len(g1) > 0
    for x, y in zip(g0, g1):
        assert torch.equal(x, y)


@pytest.mark.parametrize("freeze", [True, False])
def test_fused_head_loss_equals_linear_plus_masked_cross_entropy(cuda_device, freeze):
    """LM head + masked-mean loss as one node (logits read once, gradient left in the logits buffer) against
    the two-node path: same loss to the last place of logsumexp, same gradients to one bf16 ulp of the gradient's scale;
    an upstream gradient other than 1 (loss scaling, gradient accumulation) scales the head's outputs."""
    from multimodal_amd import functional as Fn
    torch.manual_seed(3)
    b, s, h, V = 2, 64, 256, 1024
    x = torch.randn(b, s, h, device=cuda_device).bfloat16()
    w = (torch.randn(V, h, device=cuda_device) * 0.05).bfloat16()
    labels = torch.randint(0, V, (b, s), device=cuda_device)
    mask = (torch.rand(b, s, device=cuda_device) > 0.2).float()

    def run(fused, up):
        xx = x.clone().requires_grad_(True)
        ww = w.clone().requires_grad_(not freeze)
        if fused:
            assert Fn.lm_head_loss_supported(xx, ww)
            loss = Fn.lm_head_loss(xx, ww, labels, mask)
        else:
            loss = Fn.masked_cross_entropy(Fn.linear(xx, ww), labels, mask)
        (loss * up).backward()
        return loss.detach(), xx.grad, ww.grad

    for up in (1.0, 0.25):
        l0, dx0, dw0 = run(False, up)
        l1, dx1, dw1 = run(True, up)
        assert abs(float(l0) - float(l1)) <= 2e-6 * abs(float(l0))
        assert (dx0.float() - dx1.float()).abs().max() <= 2.0 ** -8 * dx0.float().abs().max()
        if freeze:
            assert dw0 is None and dw1 is None
        else:
            assert (dw0.float() - dw1.float()).abs().max() <= 2.0 ** -8 * dw0.float().abs().max()


def test_stack_uses_the_fused_head_loss_and_matches_the_unfused_stack(cuda_device):
    from multimodal_amd import harness, mpu
    mpu.destroy_model_parallel()
    mpu.initialize_model_parallel(1)
    cfg, fused = _stack(cuda_device, 17)
    _, plain = _stack(cuda_device, 17, fused_head_loss=False)
    batch = harness.synthetic_batch(cfg, cuda_device, seed=8)
    l1, g1 = _loss_and_grads(fused, batch, 5)
    l0, g0 = _loss_and_grads(plain, batch, 5)
    assert abs(float(l0) - float(l1)) <= 2e-6 * abs(float(l0))
    for a, b in zip(g0, g1):
        assert (a.float() - b.float()).abs().max() <= 2.0 ** -6 * a.float().abs().max() + 1e-9


def test_two_stream_layer_equals_one_stream_layer(cuda_device, monkeypatch):
    """The GPT-J layer runs its attention branch and its MLP GEMMs on two HIP streams (model/transformer.py): same kernels on
    the same data, so loss and gradients must equal the single-stream run bit for bit, step after step (a missing
    dependency between the streams would show up here as a mismatch)."""
    from multimodal_amd import harness, mpu
    mpu.destroy_model_parallel()
    mpu.initialize_model_parallel(1)
    cfg, model = _stack(cuda_device, 19, num_layers=4)
    batches = [harness.synthetic_batch(cfg, cuda_device, seed=40 + i) for i in range(3)]
    results = {}
    for mode in ("0", "1", "0", "1"):
        monkeypatch.setenv("MM_BRANCH_STREAMS", mode)
        out = [_loss_and_grads(model, b, 9) for b in batches]
        torch.cuda.synchronize()
        if mode in results:
            ref = results[mode]
        else:
            ref = results.setdefault(mode, out)
        for (l0, g0), (l1, g1) in zip(results["0"], out):
            assert torch.equal(l0, l1)
            for a, b in zip(g0, g1):
                assert torch.equal(a, b)


def test_summit_70m_recipe_stack_fused_adapters_equal_gemm_chain(cuda_device, monkeypatch):
    """BASELINE.json configs[2] (configs/summit-70m-openclipH.yml): h = 512, 8 heads, GPT-J residual, erf GeLU, unfused softmax
    flags, adapters with the DEFAULT downsample factor 4 (r = 128) on a frozen LM - at reduced depth / length / vocabulary.
    The fused adapter kernels (h = 512, r = 128 instantiation) must reproduce the GEMM-chain adapters bit for bit in a whole
    training step, and a few optimizer steps must lower the loss."""
    from multimodal_amd import harness, mpu, ops
    from multimodal_amd.optim import FusedAdam
    mpu.destroy_model_parallel()
    mpu.initialize_model_parallel(1)
    cfg = dict(harness.CONFIGS["summit-70m-openclipH"])
    assert cfg["adapter_downsample_factor"] == 4 and ops.adapter_supported(cfg["hidden_size"], cfg["hidden_size"] // 4)
    cfg.update(num_layers=2, seq_length=256, padded_vocab_size=2048, image_embed_dropout_prob=0.0)
    mpu.model_parallel_cuda_manual_seed(23)
    torch.manual_seed(23)
    model = harness.MagmaStack(cfg, dtype=torch.bfloat16, device=cuda_device)
    model.train()
    batch = harness.synthetic_batch(cfg, cuda_device, seed=3)
    l_fused, g_fused = _loss_and_grads(model, batch, 5)
    monkeypatch.setattr(ops, "adapter_supported", lambda h, r: False)
    l_chain, g_chain = _loss_and_grads(model, batch, 5)
    monkeypatch.undo()
    assert torch.equal(l_fused, l_chain)
    for a, b in zip(g_fused, g_chain):
        assert torch.equal(a, b)
    opt = FusedAdam(model.trainable_parameters(), lr=2e-3, max_grad_norm=1.0)
    losses = []
    for _ in range(8):
        opt.zero_grad(set_to_none=True)
        loss = model(*batch)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0] - 0.05, losses
