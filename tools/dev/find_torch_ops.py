"""Developer tool: which framework (non-libmm_hip) device work does one eager training step launch, and from where?"""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
from torch.profiler import profile, ProfilerActivity
from multimodal_amd import harness, mpu
from multimodal_amd.optim import FusedAdam
dev = torch.device("cuda:0")
mpu.initialize_model_parallel(1)
mpu.model_parallel_cuda_manual_seed(1234)
cfg = dict(harness.CONFIGS["magma_pythia_410M"])
cfg["num_layers"] = int(os.environ.get("LAYERS", "2"))
torch.manual_seed(0)
model = harness.MagmaStack(cfg, dtype=torch.bfloat16, device=dev)
model.train()
params = model.trainable_parameters()
opt = FusedAdam(params, lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0)
batch = harness.synthetic_batch(cfg, dev, seed=1)
def step():
    opt.zero_grad(set_to_none=True)
    loss = model(*batch)
    loss.backward()
    harness.end_of_step()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
seen = {}
for e in evs:
    if e.name in ("aten::copy_", "aten::to", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::mul", "aten::add", "aten::add_", "aten::cat",
                  "aten::clone", "aten::contiguous", "aten::sum", "aten::div", "aten::embedding", "aten::zeros", "aten::ones_like", "aten::zeros_like"):
        st = [s for s in (e.stack or []) if "multimodal_amd" in s or "bench" in s or "tools" in s]
        key = (e.name, str(e.input_shapes)[:80], st[0] if st else "<autograd engine>")
        seen[key] = seen.get(key, 0) + 1
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
print("---- device kernels not from libmm_hip")
ker = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA and "mm::" not in e.name:
        ker[e.name[:90]] = ker.get(e.name[:90], 0) + 1
for k, v in sorted(ker.items(), key=lambda kv: -kv[1]):
    print(v, k)
