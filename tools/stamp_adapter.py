#!/usr/bin/env python3
"""Diagnostic (needs a -DMM_AD_STAMP build of adapter.hip): per-phase cycle stamps of the fused adapter kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops
T, H, r = 8192, 1024, 128
rnd = lambda *s, sc=1.0: (torch.randn(*s, device="cuda") * sc).bfloat16()
o, w1, b1, w2, b2 = rnd(T, H), rnd(r, H, sc=0.03), rnd(r), rnd(H, r, sc=0.1), rnd(H)
for name in ("fwd", "bwd"):
    for _ in range(5):
        if name == "fwd":
            out, h1 = ops.adapter_fwd(o, w1, b1, w2, b2)
        else:
            out, _ = ops.adapter_bwd_dx(o, w2, w1, h1)
    torch.cuda.synchronize()
    d = out.view(torch.int32).view(T, H // 2)[::32, :4].cpu().float()
    print(name, "prologue / product1 / product2 cycles: median", d[:, :3].median(0).values.tolist(), "max", d[:, :3].max(0).values.tolist(),
          "start spread", float(d[:, 3].max() - d[:, 3].min()))
