#!/usr/bin/env python3
"""Developer A/B: the training step's GEMM shapes on every MFMA kernel of the library, in ONE process on one device (boxes
differ by up to ~10 %, so only same-run ratios mean anything): 128x128 (gemm.hip), pipelined 256x256 (gemm256.hip), ping-pong
256x256 and 256x128 (gemm_pp.hip)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import _lib, ops
_lib.use_dev_lib().enter()  # developer library (make DEV=1): the kernel-forcing switches / mm_debug_* live there only
from tools.bench_kernels import timeit, rnd, T, H, DEV

KINDS = [("128x128", {"MM_GEMMPP": "-1", "MM_GEMM256": "-1"}), ("256x256", {"MM_GEMMPP": "-1", "MM_GEMM256": "1"}),
         ("pp256", {"MM_GEMMPP": "256"}), ("pp128", {"MM_GEMMPP": "128"}), ("policy", {})]
shapes = [("h4h gelu", T, 4 * H, H, _lib.EPI_BIAS_GELU_TANH), ("4hh dx gelu'", T, 4 * H, H, _lib.EPI_GELU_TANH_BWD),
          ("qkv fwd", T, 3 * H, H, _lib.EPI_BIAS), ("4hh fwd", T, H, 4 * H, _lib.EPI_NONE), ("dense", T, H, H, _lib.EPI_NONE),
          ("qkv dx", T, H, 3 * H, _lib.EPI_NONE), ("lm head", T, 50304, H, _lib.EPI_NONE), ("head dx", T, H, 50304, _lib.EPI_NONE)]
import re
custom = [a for a in sys.argv[1:] if re.fullmatch(r"\d+,\d+,\d+", a)]   # M,N,K triples: plain epilogue
names = [a for a in sys.argv[1:] if a not in custom]
if names or custom:
    shapes = [s for s in shapes if any(a in s[0] for a in names)]
    shapes += [("custom", *(int(v) for v in c.split(",")), _lib.EPI_NONE) for c in custom]
for name, M, N, K, epi in shapes:
    a = [rnd(M, K) for _ in range(2)]
    w = [rnd(N, K, scale=0.02) for _ in range(2)]
    bias = rnd(N) if epi != _lib.EPI_NONE else None
    res = rnd(M, N) if epi == _lib.EPI_GELU_TANH_BWD else None
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    i = [0]

    def f():
        i[0] += 1
        return ops.gemm_nt(a[i[0] % 2], w[i[0] % 2], bias, epi, residual=res, out=out)
    line = []
    for rep in range(2):
        for kname, env in KINDS:
            for k in ("MM_GEMMPP", "MM_GEMM256"):
                os.environ.pop(k, None)
            os.environ.update(env)
            us = timeit(f, n=30, warm=6)
            line.append("%s %.1f" % (kname, us))
    for k in ("MM_GEMMPP", "MM_GEMM256"):
        os.environ.pop(k, None)
    fl = 2.0 * M * N * K
    print("%-13s M=%d N=%d K=%d (%.1f GF; 1 PF = %.1f us):  %s" % (name, M, N, K, fl / 1e9, fl / 1e9, "  ".join(line)), flush=True)
