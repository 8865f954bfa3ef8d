import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
from multimodal_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
dt = torch.bfloat16
rows, enc, h = 2, 64, 256
w = (torch.randn(h, enc, device=dev) * 0.1).to(dt); b = torch.zeros(h, device=dev, dtype=dt)
lw = torch.ones(h, device=dev, dtype=dt); lb = torch.zeros(h, device=dev, dtype=dt)
xs = [torch.randn(rows, enc, device=dev).to(dt) for _ in range(3)]
static = xs[0].clone()
eager = [ops.prefix_tail_fwd(x, w, b, lw, lb, 1e-5)[0].clone() for x in xs]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    ops.prefix_tail_fwd(static, w, b, lw, lb, 1e-5)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = ops.prefix_tail_fwd(static, w, b, lw, lb, 1e-5)[0]
for i, x in enumerate(xs):
    static.copy_(x)
    g.replay()
    torch.cuda.synchronize()
    print(i, "replay equals eager:", torch.equal(out, eager[i]), float((out.float() - eager[i].float()).abs().max()))
