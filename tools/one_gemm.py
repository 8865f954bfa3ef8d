#!/usr/bin/env python3
"""Developer helper: launch one gemm_nt shape a few times (for rocprofv3 --pmc / --kernel-trace runs).
    python tools/one_gemm.py M N K [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
a = torch.randn(M, K, device="cuda").bfloat16()
b = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(reps):
    ops.gemm_nt(a, b, out=out)
torch.cuda.synchronize()
print("ok", float(out.float().abs().mean()))
