#!/usr/bin/env python3
"""Developer timing of the attention kernels at the magma_pythia_410M shape, repeated in one process.

Two library builds are compared on ONE box with tools/ab_libs.sh (boxes differ by up to 25 % on these kernels):
    tools/ab_libs.sh prepare && gpurun -- 'bash tools/ab_libs.sh run python -u tools/ab_attn.py'
Round 1 used it for: MFMA groups at raised wave priority (forward 55.1 -> 52.1 us, backward 169 -> 163 us: kept), a
launch-id stagger of the two workgroups of a CU (no change: dropped) and a sched_group_barrier interleave of the P.V MFMAs
with the second half's exponentials (within noise: dropped)."""
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


for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    print("fwd %.1f us  bwd %.1f us (dq with the delta row sums, then dkv)" % (timeit(fwd, n=60, warm=10), timeit(bwd, n=40, warm=5)))
