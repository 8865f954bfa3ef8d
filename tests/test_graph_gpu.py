"""The C-ABI entry points neither allocate nor synchronise, so a whole training step (forward, backward, gradient
clipping, fused Adam) can be captured in a HIP graph and replayed: the replayed step must reproduce the eager step bit
for bit (same kernels, same order, same data)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_hip_graph_replay_equals_eager_step(cuda_device):
    from multimodal_amd import harness, mpu
    mpu.destroy_model_parallel()
    mpu.initialize_model_parallel(1)
    mpu.model_parallel_cuda_manual_seed(7)
    cfg = harness.CONFIGS["tiny"]

    def build():
        torch.manual_seed(5)
        model = harness.MagmaStack(cfg, dtype=torch.bfloat16, device=cuda_device)
        model.train()
        params = model.trainable_parameters()
        opt = torch.optim.Adam(params, lr=1e-2, fused=True, capturable=True)
        return model, params, opt

    batch = harness.synthetic_batch(cfg, cuda_device, seed=3)

    def step(model, params, opt):
        opt.zero_grad(set_to_none=False)
        loss = model(*batch)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        return loss

    # eager: warm-up steps + 2 more
    m1, p1, o1 = build()
    for _ in range(3):
        step(m1, p1, o1)
    eager_losses = [float(step(m1, p1, o1).detach()) for _ in range(2)]

    # graph: identical warm-up on a side stream (capture needs warmed-up allocator / optimizer state), then capture one step and replay it
    m2, p2, o2 = build()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step(m2, p2, o2)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = step(m2, p2, o2)        # capture only records: nothing runs here
    graph_losses = []
    for _ in range(2):
        g.replay()
        graph_losses.append(float(loss.detach()))
    assert graph_losses == eager_losses
    for a, b in zip(p1, p2):
        assert torch.equal(a, b)
