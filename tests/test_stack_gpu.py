"""The stack around the layers on the GPU: checkpoint round trip and the activation-recompute policy."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stack(cuda_device, seed, **over):
    from multimodal_amd import harness
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
