#!/usr/bin/env python3
"""Developer experiment: how much of a 256x256 tile's time on the h->4h GEMM is the epilogue, and is that part bound by the
chip (every CU bursting its stores at once) or by the CU?  Same kernel at M = 2048 .. 16384 (128 .. 1024 tiles: half the
CUs busy, one round, two rounds, four rounds), with the plain and the bias-GeLU epilogue."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import _lib, ops
_lib.use_dev_lib().enter()  # developer library (make DEV=1): the kernel-forcing switches / mm_debug_* live there only
from tools.bench_kernels import timeit, rnd, H, DEV

os.environ["MM_GEMM256"] = "1"
N, K = 4 * H, H
for M in (2048, 4096, 8192, 16384):
    a = [rnd(M, K) for _ in range(3)]
    w = [rnd(N, K, scale=0.02) for _ in range(3)]
    bias = rnd(N)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    i = [0]
    line = "M=%5d tiles=%4d " % (M, M // 256 * (N // 256))
    for name, epi in (("none", _lib.EPI_NONE), ("bias", _lib.EPI_BIAS), ("gelu", _lib.EPI_BIAS_GELU_TANH)):
        def f():
            i[0] += 1
            return ops.gemm_nt(a[i[0] % 3], w[i[0] % 3], bias if epi != _lib.EPI_NONE else None, epi, out=out)
        t = [timeit(f, n=40, warm=8) for _ in range(2)]
        line += "  %s %.1f / %.1f us (%.0f TF/s)" % (name, t[0], t[1], 2.0 * M * N * K / min(t) / 1e6)
    print(line)
