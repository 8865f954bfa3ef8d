#!/usr/bin/env python3
"""Developer helper: how long does the host take to ENQUEUE one training step (no sync) vs the device to run it?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import harness, mpu
dev = torch.device("cuda", 0)
mpu.initialize_model_parallel(1); mpu.model_parallel_cuda_manual_seed(1234)
# usage: cpu_overhead.py [config [layers]] - e.g. "6-7B 8": the per-layer enqueue cost of a wide config without building all of it
cfg = dict(harness.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "magma_pythia_410M"])
if len(sys.argv) > 2:
    cfg["num_layers"] = int(sys.argv[2])
print("config %s, %d layers, h=%d" % (sys.argv[1] if len(sys.argv) > 1 else "magma_pythia_410M", cfg["num_layers"], cfg["hidden_size"]))
model = harness.MagmaStack(cfg, dtype=torch.bfloat16, device=dev); model.train()
params = model.trainable_parameters()
opt = torch.optim.Adam(params, lr=8e-4, fused=True)
batch = harness.synthetic_batch(cfg, dev, seed=0)
def step():
    opt.zero_grad(set_to_none=True)
    loss = model(*batch); loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for tag in ("a", "b"):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("enqueue %.2f ms, until device idle %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("5 steps: enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) * 200, (t2 - t0) * 200))
