#!/usr/bin/env python3
"""Developer micro-benchmark: the hot kernels at the magma_pythia_410M shapes, each timed alone with HIP events
(50 launches after 10 warm-ups, input buffers rotated so the working set does not sit in L2/MALL).

    python tools/bench_kernels.py [attn] [gemm] [tn] [small]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodal_amd import _lib, ops  # noqa: E402

DEV = torch.device("cuda", 0)
S, B, NP, HN, H = 2048, 4, 16, 64, 1024
T = S * B


def timeit(fn, n=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV, dtype=torch.float32) * scale).to(torch.bfloat16)


def bench_attn():
    qkvs = [rnd(S, B, NP, 3 * HN) for _ in range(4)]
    scale = HN ** -0.5
    i = [0]

    def fwd():
        i[0] += 1
        return ops.attn_fwd(qkvs[i[0] % 4], S, B, NP, HN, scale, None)
    t = timeit(fwd)
    fl = 4.0 * S * S * HN * B * NP / 2
    print("attn_fwd   %8.1f us  %7.1f TF/s (causal-exact flops)" % (t, fl / t / 1e6))
    ctx, lse = fwd()
    dctx = rnd(S, B, NP * HN)

    def bwd():
        return ops.attn_bwd(dctx, ctx, lse, qkvs[i[0] % 4], S, B, NP, HN, scale, None)
    t = timeit(bwd)
    print("attn_bwd   %8.1f us  %7.1f TF/s (dq with the delta row sums, then dkv)" % (t, 2.5 * fl / t / 1e6))


def bench_gemm():
    shapes = [("qkv  fwd", T, 3 * H, H, _lib.EPI_BIAS), ("h4h  gelu", T, 4 * H, H, _lib.EPI_BIAS_GELU_TANH),
              ("4hh  fwd", T, H, 4 * H, _lib.EPI_NONE), ("dense fwd", T, H, H, _lib.EPI_NONE),
              ("h4h  dx", T, H, 4 * H, _lib.EPI_NONE), ("4hh  dx", T, 4 * H, H, _lib.EPI_NONE),
              ("4hh dx gelu'", T, 4 * H, H, _lib.EPI_GELU_TANH_BWD),
              ("qkv  dx", T, H, 3 * H, _lib.EPI_NONE), ("adapter dn", T, H // 8, H, _lib.EPI_BIAS_RELU),
              ("adapter up", T, H, H // 8, _lib.EPI_BIAS), ("lm head", T, 50432, H, _lib.EPI_NONE)]
    for name, M, N, K, epi in shapes:
        a = [rnd(M, K) for _ in range(3)]
        w = [rnd(N, K, scale=0.02) for _ in range(3)]
        bias = rnd(N) if epi != _lib.EPI_NONE else None
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        res = rnd(M, N) if epi == _lib.EPI_GELU_TANH_BWD else None
        i = [0]

        def f():
            i[0] += 1
            return ops.gemm_nt(a[i[0] % 3], w[i[0] % 3], bias, epi, residual=res, out=out)
        t = timeit(f, n=30, warm=5)
        print("gemm_nt %-10s M=%5d N=%5d K=%5d %8.1f us %7.1f TF/s" % (name, M, N, K, t, 2.0 * M * N * K / t / 1e6))


def bench_tn():
    for name, N, K in [("adapter dn", H // 8, H), ("adapter up", H, H // 8), ("h4h dW", 4 * H, H), ("dense dW", H, H)]:
        dy, x = rnd(T, N), rnd(T, K)
        t = timeit(lambda: ops.gemm_tn(dy, x), n=30, warm=5)
        print("gemm_tn %-10s N=%5d K=%5d (M=%d) %8.1f us %7.1f TF/s" % (name, N, K, T, t, 2.0 * T * N * K / t / 1e6))


def bench_small():
    x = rnd(T, H)
    x4 = rnd(T, 4 * H)
    b4 = rnd(4 * H)
    w = rnd(H)
    for name, fn, nbytes in [
        ("colsum [T,h]", lambda: ops.colsum(x), 2 * T * H),
        ("colsum [T,4h]", lambda: ops.colsum(x4), 2 * T * 4 * H),
        ("bias_gelu_bwd", lambda: ops.bias_gelu_bwd(x4, x4, b4, 0), 6 * T * 4 * H),
        ("layernorm_fwd dual", lambda: ops.layernorm_fwd(x, w, w, 1e-5, w, w), 6 * T * H),
    ]:
        t = timeit(fn)
        print("%-20s %8.1f us %7.2f TB/s" % (name, t, nbytes / t / 1e6))


def bench_softmax():
    """The standalone scaled_*_softmax kernels at the 410M attention shape: [b*np = 64, s = 2048, s = 2048] bf16.
    Algorithmic bytes: causal forward 3 * rows * sk (reads the lower triangle = half, writes every element: 1 + 2),
    masked forward 4 * rows * sk (+ 1 byte per element of mask), backward 6 * rows * sk (grad + y in, grad out)."""
    nb, s = B * NP, S
    xs = [rnd(nb, s, s) for _ in range(2)]
    i = [0]

    def nxt():
        i[0] += 1
        return xs[i[0] % 2]
    el = nb * s * s
    t = timeit(lambda: ops.softmax_causal_fwd(nxt(), 1.0), n=20, warm=4)
    print("softmax_causal_fwd  %8.1f us  %6.2f TB/s (3 B/element)" % (t, 3.0 * el / t / 1e6))
    y = ops.softmax_causal_fwd(xs[0], 1.0)
    gs = [rnd(nb, s, s) for _ in range(2)]
    t = timeit(lambda: ops.softmax_causal_bwd(gs[i[0] % 2], y, 1.0), n=20, warm=4)
    print("softmax_causal_bwd  %8.1f us  %6.2f TB/s (6 B/element; the causal kernel touches the lower triangle only: 3)" % (t, 6.0 * el / t / 1e6))
    mask = torch.triu(torch.ones(s, s, dtype=torch.bool, device=DEV), 1).view(1, 1, s, s)
    x4 = [x.view(B, NP, s, s) for x in xs]
    t = timeit(lambda: ops.softmax_masked_fwd(x4[i[0] % 2], mask, 1.0), n=20, warm=4)
    print("softmax_masked_fwd  %8.1f us  %6.2f TB/s (4 B/element + the shared mask)" % (t, 4.0 * el / t / 1e6))
    y4 = ops.softmax_masked_fwd(x4[0], mask, 1.0)
    t = timeit(lambda: ops.softmax_masked_bwd(gs[i[0] % 2].view(B, NP, s, s), y4, 1.0), n=20, warm=4)
    print("softmax_masked_bwd  %8.1f us  %6.2f TB/s (6 B/element)" % (t, 6.0 * el / t / 1e6))


def bench_adapter():
    r = H // 8
    o, g = rnd(T, H), rnd(T, H)
    w1, b1, w2, b2 = rnd(r, H, scale=0.03), rnd(r), rnd(H, r, scale=0.1), rnd(H)
    t = timeit(lambda: ops.adapter_fwd(o, w1, b1, w2, b2))
    print("adapter_fwd    %8.1f us  (%.2f TB/s of o+out+h1)" % (t, (2 * T * H * 2 + T * r * 2) / t / 1e6))
    _, h1 = ops.adapter_fwd(o, w1, b1, w2, b2)
    w2t, w1t = ops.transpose2d(w2), ops.transpose2d(w1)
    t = timeit(lambda: ops.adapter_bwd_dx(g, w2, w1, h1))
    print("adapter_bwd_dx %8.1f us" % t)
    dpre = rnd(T, r)
    t = timeit(lambda: ops.gemm_tn_pair(dpre, o, g, h1))
    print("adapter dW pair %8.1f us (host-timed: use the kernel trace for device time)" % t)
    t = timeit(lambda: (ops.transpose2d(w2), ops.transpose2d(w1)))
    print("transposes x2  %8.1f us" % t)


if __name__ == "__main__":
    which = sys.argv[1:] or ["attn", "gemm", "tn", "small", "adapter", "softmax"]
    print("device:", torch.cuda.get_device_name(0))
    for w in which:
        {"attn": bench_attn, "gemm": bench_gemm, "tn": bench_tn, "small": bench_small, "adapter": bench_adapter, "softmax": bench_softmax}[w]()
