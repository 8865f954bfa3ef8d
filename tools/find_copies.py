#!/usr/bin/env python3
"""Developer helper: which host-side ops of one eager training step launch torch's own kernels (copies, fills, casts, elementwise)
rather than the HIP library's?  Prints every device kernel that is not an mm:: kernel with the chain of CPU ops above it."""
import os, sys, collections
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import harness, mpu
from multimodal_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
mpu.initialize_model_parallel(1); mpu.model_parallel_cuda_manual_seed(1234)
cfg = dict(harness.CONFIGS["magma_pythia_410M"]); cfg["num_layers"] = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = harness.MagmaStack(cfg, dtype=torch.bfloat16, device=dev); model.train()
params = model.trainable_parameters()
opt = FusedAdam(params, lr=8e-4, max_grad_norm=1.0)
batch = harness.synthetic_batch(cfg, dev, seed=0)
def step():
    opt.zero_grad(set_to_none=True)
    loss = model(*batch); loss.backward(); harness.end_of_step(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
ev = prof.events()
chains = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        for k in e.kernels:
            if "mm::" in k.name:
                continue
            names, p = [], e
            while p is not None:
                names.append(p.name); p = p.cpu_parent
            chains[(k.name[:60], " < ".join(names[:6]))] += 1
for (k, c), n in sorted(chains.items(), key=lambda kv: -kv[1]):
    print("%4d  %-60s %s" % (n, k, c))
