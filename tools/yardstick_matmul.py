#!/usr/bin/env python3
"""Developer yardstick (NOT part of the product path): what the vendor GEMM library reaches on the benchmark shapes."""
import torch, torch.nn.functional as F
T, H = 8192, 1024
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, M, N, K in [("qkv fwd", T, 3 * H, H), ("h4h fwd", T, 4 * H, H), ("4hh fwd", T, H, 4 * H), ("dense", T, H, H), ("qkv dx", T, H, 3 * H), ("head", T, 50432, H)]:
    a = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(3)]
    w = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(3)]
    i = [0]
    def f():
        i[0] += 1
        return F.linear(a[i[0] % 3], w[i[0] % 3])
    t = timeit(f)
    print("F.linear %-8s M=%5d N=%5d K=%5d %8.1f us %7.1f TF/s" % (name, M, N, K, t, 2.0 * M * N * K / t / 1e6))
