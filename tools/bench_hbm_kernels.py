#!/usr/bin/env python3
"""Developer timing of the HBM-bound kernels of the step (dual LayerNorm forward / backward, fused cross entropy) without host launch
overhead: 16 launches over rotating operand sets (larger than the 256 MiB Infinity Cache in total) captured into one HIP graph,
replayed 10 times; us per launch and fraction of the 8 TB/s HBM peak for the algorithmic bytes."""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
from multimodal_amd import ops
dev = torch.device("cuda:0")
T, H, V = 8192, 1024, 50304
rnd = lambda *s: torch.randn(*s, device=dev).bfloat16()

def graph_time(fn, nsets, per_graph=16, reps=10):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for i in range(nsets):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for i in range(per_graph):
                fn(i % nsets)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * per_graph)

which = sys.argv[1:] or ["ln", "ce"]
if "ln" in which:
    n = 12
    xs = [rnd(T, H) for _ in range(n)]
    w = [rnd(H) for _ in range(4)]
    t = graph_time(lambda i: ops.layernorm_fwd(xs[i], w[0], w[1], 1e-5, w[2], w[3]), n)
    print("layernorm_fwd dual      %7.1f us  %.2f TB/s  %.3f of peak" % (t, 6 * T * H / t / 1e6, 6 * T * H / t / 1e6 / 8))
    y1, y2, mean, rstd = ops.layernorm_fwd(xs[0], w[0], w[1], 1e-5, w[2], w[3])
    dys = [rnd(T, H) for _ in range(n)]
    t = graph_time(lambda i: ops.layernorm_bwd(dys[i], xs[i], w[0], mean, rstd, dy2=dys[(i + 1) % n], w2=w[2], dresidual=dys[(i + 2) % n], param_grads=False), n)
    print("layernorm_bwd dual+res  %7.1f us  %.2f TB/s  %.3f of peak" % (t, 10 * T * H / t / 1e6, 10 * T * H / t / 1e6 / 8))
if "ce" in which:
    lg = [rnd(T, V) for _ in range(2)]
    labels = torch.randint(0, V, (T,), device=dev)
    roww = torch.full((T,), 1.0 / T, device=dev)
    t = graph_time(lambda i: ops.cross_entropy_fwd_grad(lg[i], labels, roww), 2, per_graph=4, reps=5)
    print("ce_fwd_grad             %7.1f us  %.2f TB/s  %.3f of peak" % (t, 4 * T * V / t / 1e6, 4 * T * V / t / 1e6 / 8))
