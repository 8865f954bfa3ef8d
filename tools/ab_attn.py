#!/usr/bin/env python3
"""Developer A/B of attention-forward scheduling experiments (MM_ATTN_X bit mask), one process, interleaved."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops
from tools.bench_kernels import timeit, rnd, S, B, NP, HN

qkvs = [rnd(S, B, NP, 3 * HN) for _ in range(4)]
i = [0]
def fwd():
    i[0] += 1
    return ops.attn_fwd(qkvs[i[0] % 4], S, B, NP, HN, HN ** -0.5, None)
ctx, lse = fwd()
dctx = rnd(S, B, NP * HN)
def bwd():
    return ops.attn_bwd(dctx, ctx, lse, qkvs[i[0] % 4], S, B, NP, HN, HN ** -0.5, None)
modes = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 3, 5]
ref = None
for rep in range(2):
    for m in modes:
        os.environ["MM_ATTN_X"] = str(m)
        t = timeit(fwd, n=60, warm=10)
        tb = timeit(bwd, n=40, warm=5)
        out = ops.attn_fwd(qkvs[0], S, B, NP, HN, HN ** -0.5, None)[0]
        if ref is None:
            ref = out.clone()
        print("mode %d: fwd %.1f us  bwd %.1f us  same=%s" % (m, t, tb, torch.equal(out, ref)))
