#!/usr/bin/env python3
"""Developer A/B: GEMM shapes on the 128x128 kernel (MM_GEMM256=-1) and the 256x256 pipelined kernel (MM_GEMM256=1), in ONE
process on one device (boxes differ by up to ~10 %, so only same-run ratios mean anything).  A 256x128 variant of the
pipelined kernel was measured here too (round 1): within 2 % of the 128x128 kernel on every N = h shape, dropped."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import _lib, ops
from tools.bench_kernels import timeit, rnd, T, H, DEV

shapes = [("4hh fwd", T, H, 4 * H, _lib.EPI_NONE), ("dense", T, H, H, _lib.EPI_NONE), ("qkv dx", T, H, 3 * H, _lib.EPI_NONE),
          ("qkv fwd", T, 3 * H, H, _lib.EPI_BIAS), ("h4h gelu", T, 4 * H, H, _lib.EPI_BIAS_GELU_TANH)]
for name, M, N, K, epi in shapes:
    a = [rnd(M, K) for _ in range(3)]
    w = [rnd(N, K, scale=0.02) for _ in range(3)]
    bias = rnd(N) if epi != _lib.EPI_NONE else None
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    i = [0]

    def f():
        i[0] += 1
        return ops.gemm_nt(a[i[0] % 3], w[i[0] % 3], bias, epi, out=out)
    res = []
    for force in ("-1", "1", "-1", "1"):
        os.environ["MM_GEMM256"] = force
        res.append(timeit(f, n=40, warm=8))
    os.environ.pop("MM_GEMM256")
    print("%-9s M=%d N=%d K=%d   128x128 %.1f / %.1f us   256x256 %.1f / %.1f us" % (name, M, N, K, res[0], res[2], res[1], res[3]))
