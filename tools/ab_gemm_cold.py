#!/usr/bin/env python3
"""Developer A/B: the h->4h bias-GeLU GEMM (M=8192 N=4096 K=1024) with a working set that does NOT fit the 256 MiB Infinity
Cache (R rotating sets of A, W, out, aux), as inside the training step, against the 2-set microbenchmark where it does."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import _lib, ops
_lib.use_dev_lib().enter()  # developer library (make DEV=1): the kernel-forcing switches / mm_debug_* live there only
from tools.bench_kernels import timeit, rnd, T, H, DEV
M, N, K = T, 4 * H, H
for R in (1, 2, 8, 16):
    a = [rnd(M, K) for _ in range(R)]
    w = [rnd(N, K, scale=0.02) for _ in range(R)]
    bias = rnd(N)
    outs = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(R)]
    auxs = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(R)]
    lib = _lib.get_lib()
    st = torch.cuda.current_stream().cuda_stream
    i = [0]
    for epi, ename in ((_lib.EPI_BIAS_GELU_TANH, "gelu"), (_lib.EPI_BIAS, "bias")):
        def f():
            i[0] += 1
            j = i[0] % R
            _lib.check(lib.mm_gemm_nt(outs[j].data_ptr(), auxs[j].data_ptr(), a[j].data_ptr(), w[j].data_ptr(), bias.data_ptr(), None,
                                      M, N, K, K, K, N, epi, 0, st))
        line = []
        for kname, env in (("128x128", {"MM_GEMMPP": "-1", "MM_GEMM256": "-1"}), ("256x256", {"MM_GEMMPP": "-1", "MM_GEMM256": "1"}),
                           ("pp256", {"MM_GEMMPP": "256"}), ("pp128", {"MM_GEMMPP": "128"})):
            for k in ("MM_GEMMPP", "MM_GEMM256"):
                os.environ.pop(k, None)
            os.environ.update(env)
            line.append("%s %.1f" % (kname, timeit(f, n=48, warm=16)))
        print("R=%2d (%4d MiB working set) %-5s  %s" % (R, R * (16 + 8 + 128), ename, "  ".join(line)), flush=True)
    del a, w, outs, auxs
