#!/usr/bin/env python3
"""Developer helper: a few launches of the fused adapter kernels (for rocprofv3 --pmc runs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops
T, H, r = 8192, 1024, 128
rnd = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).bfloat16()
o, w1, b1, w2, b2 = rnd(T, H), rnd(r, H, sc=0.03), rnd(r), rnd(H, r, sc=0.1), rnd(H)
for _ in range(4):
    out, h1 = ops.adapter_fwd(o, w1, b1, w2, b2)
torch.cuda.synchronize()
print("ok")
