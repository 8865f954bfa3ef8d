#!/usr/bin/env python3
"""Developer micro-benchmark: fused attention at head size 128 (summit-1-3B TP2: 9 heads per rank; 6-7B TP8: 4 heads per rank)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops
from tools.bench_kernels import timeit, rnd
S, B = 2048, 4
for NP, HN in [(9, 128), (4, 128), (16, 64)]:
    qkv = rnd(S, B, NP, 3 * HN)
    scale = HN ** -0.5
    t = timeit(lambda: ops.attn_fwd(qkv, S, B, NP, HN, scale, None), n=30, warm=5)
    fl = 4.0 * S * S * HN * B * NP / 2
    ctx, lse = ops.attn_fwd(qkv, S, B, NP, HN, scale, None)
    dctx = rnd(S, B, NP * HN)
    t2 = timeit(lambda: ops.attn_bwd(dctx, ctx, lse, qkv, S, B, NP, HN, scale, None), n=30, warm=5)
    print("np=%2d hn=%3d  fwd %7.1f us %6.1f TF/s   bwd %7.1f us %6.1f TF/s (causal-exact flops)" % (NP, HN, t, fl / t / 1e6, t2, 2.5 * fl / t2 / 1e6))
