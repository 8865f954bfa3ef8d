#!/usr/bin/env python3
"""Developer experiment: capture forward+backward (+clip+Adam) of the 410M step in a HIP graph and compare replay time
with eager launches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import harness, mpu
dev = torch.device("cuda", 0)
mpu.initialize_model_parallel(1); mpu.model_parallel_cuda_manual_seed(1234)
cfg = harness.CONFIGS["magma_pythia_410M"]
model = harness.MagmaStack(cfg, dtype=torch.bfloat16, device=dev); model.train()
params = model.trainable_parameters()
from multimodal_amd.optim import FusedAdam
opt = FusedAdam(params, lr=8e-4, betas=(0.9, 0.95), max_grad_norm=1.0)
batch = harness.synthetic_batch(cfg, dev, seed=0)
def step():
    opt.zero_grad(set_to_none=True)
    loss = model(*batch); loss.backward()
    opt.step()
    return loss
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print("eager  %.2f ms/step" % ((time.perf_counter() - t0) * 100))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
for _ in range(2): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize()
print("graph  %.2f ms/step  loss %.4f" % ((time.perf_counter() - t0) * 100, float(loss)))
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print("eager  %.2f ms/step (again)" % ((time.perf_counter() - t0) * 100))
