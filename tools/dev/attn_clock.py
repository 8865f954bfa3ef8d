"""Diagnostic (variant library built with -DMM_AB_CLOCK): the shader clock the chip holds inside attn_fwd_k's tile loop."""
import os, sys, statistics
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
from multimodal_amd import ops
from tools.bench_kernels import rnd, S, B, NP, HN
qkvs = [rnd(S, B, NP, 3 * HN) for _ in range(4)]
for i in range(400):   # ~20 ms of back-to-back launches: the clock settles
    ctx, lse = ops.attn_fwd(qkvs[i % 4], S, B, NP, HN, HN ** -0.5, None)
torch.cuda.synchronize()
st = lse.view(B * NP, S // 128, 128)[:, :, :4].reshape(-1, 4).cpu()
assert bool((st[:, 3] == -12345.0).all())
ghz = (st[:, 0] / st[:, 1] * 0.1)
per_tile = st[:, 0] / st[:, 2]
print("in-kernel clock: median %.3f GHz (p10 %.3f, p90 %.3f); shader cycles per 64-key tile per workgroup: median %.0f (heaviest block %.0f, lightest %.0f)"
      % (statistics.median(ghz.tolist()), sorted(ghz.tolist())[102], sorted(ghz.tolist())[921], statistics.median(per_tile.tolist()),
         float(per_tile[st[:, 2] == st[:, 2].max()].median()), float(per_tile[st[:, 2] == st[:, 2].min()].median())))
