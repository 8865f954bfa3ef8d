#!/usr/bin/env python3
"""Developer experiment: which operand of the h->4h bias-GeLU GEMM makes it slower when the working set is cache-cold -
rotate only A, only W, only the outputs, or everything (16 sets), policy kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import _lib
from tools.bench_kernels import timeit, rnd, T, H, DEV
M, N, K = T, 4 * H, H
R = 16
a = [rnd(M, K) for _ in range(R)]
w = [rnd(N, K, scale=0.02) for _ in range(R)]
bias = rnd(N)
outs = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(R)]
auxs = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(R)]
lib = _lib.get_lib()
st = torch.cuda.current_stream().cuda_stream
i = [0]
for name, ra, rw, ro in (("nothing rotates", 1, 1, 1), ("A rotates", R, 1, 1), ("W rotates", 1, R, 1), ("outputs rotate", 1, 1, R),
                         ("A + W rotate", R, R, 1), ("all rotate", R, R, R), ("nothing rotates", 1, 1, 1)):
    def f():
        i[0] += 1
        j = i[0]
        _lib.check(lib.mm_gemm_nt(outs[j % ro].data_ptr(), auxs[j % ro].data_ptr(), a[j % ra].data_ptr(), w[j % rw].data_ptr(),
                                  bias.data_ptr(), None, M, N, K, K, K, N, _lib.EPI_BIAS_GELU_TANH, 0, st))
    print("%-18s %.1f us" % (name, timeit(f, n=64, warm=16)), flush=True)
