#!/usr/bin/env python3
"""Developer experiment: which operand of the h->4h bias-GeLU GEMM (policy kernel) makes it slower when cache-cold.  Configs are
interleaved over many rounds and the MEDIAN of per-launch HIP-event times is reported (the chip's clock state drifts by +-10 %
over a process, so back-to-back blocks of one config each say little)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import _lib
_lib.use_dev_lib().enter()  # developer library (make DEV=1): the kernel-forcing switches / mm_debug_* live there only
from tools.bench_kernels import rnd, T, H, DEV
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (T, 4 * H, H)
EPI = int(sys.argv[4]) if len(sys.argv) > 4 else _lib.EPI_BIAS_GELU_TANH
R = max(2, min(16, (3 << 30) // (2 * (M * K + N * K + 2 * M * N))))
a = [rnd(M, K) for _ in range(R)]
w = [rnd(N, K, scale=0.02) for _ in range(R)]
bias = rnd(N)
outs = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(R)]
auxs = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(R)]
big = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)   # flushing this evicts the Infinity Cache
lib = _lib.get_lib()
st = torch.cuda.current_stream().cuda_stream
configs = [("all hot", 1, 1, 1, False, False), ("A cold", R, 1, 1, False, False), ("W cold", 1, R, 1, False, False),
           ("outputs cold", 1, 1, R, False, False), ("all cold", R, R, R, False, False), ("all cold, W touched first", R, R, R, True, False),
           ("all cold, A+W touched first", R, R, R, True, True)]
times = {c[0]: [] for c in configs}
ctr = 0
for rnd_i in range(12):
    for name, ra, rw, ro, touch_w, touch_a in configs:
        for rep in range(4):
            ctr += 1
            j = ctr
            if touch_w:
                w[j % rw].view(torch.int32).sum()
            if touch_a:
                a[j % ra].view(torch.int32).sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.mm_gemm_nt(outs[j % ro].data_ptr(), auxs[j % ro].data_ptr(), a[j % ra].data_ptr(), w[j % rw].data_ptr(),
                                      bias.data_ptr(), None, M, N, K, K, K, N, EPI, 0, st))
            e1.record()
            torch.cuda.synchronize()
            if rnd_i >= 2:
                times[name].append(e0.elapsed_time(e1) * 1e3)
print("M=%d N=%d K=%d epilogue %d, %d rotating sets" % (M, N, K, EPI, R))
for name, *_ in configs:
    v = sorted(times[name])
    print("%-30s median %.1f us   p10 %.1f   p90 %.1f   (%d launches)" % (name, statistics.median(v), v[len(v) // 10], v[-len(v) // 10 - 1], len(v)))
