#!/usr/bin/env python3
"""Same-process A/B of mm_gemm_tn_pair (the adapter's two weight gradients + bias gradients: launch pair = split kernel + slab
reduce) on two builds of the library: interleaved, alternating order, rotating operand sets (cold), medians of HIP-event times.
    python tools/ab_tn_pair.py libA.so libB.so [M h r]"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import _lib
pa, pb = sys.argv[1:3]
M, H, R = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (8192, 1024, 128)
libs = {}
for tag, path in (("A", pa), ("B", pb)):
    L = ctypes.CDLL(os.path.abspath(path))
    for n in ("mm_gemm_tn_pair", "mm_gemm_tn_pair_splits"):
        getattr(L, n).restype, getattr(L, n).argtypes = _lib.PROTOTYPES[n]
    libs[tag] = L
dev = torch.device("cuda:0")
NS = 6
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *sh: torch.randn(*sh, device=dev, generator=g).bfloat16()
sets = [(rnd(M, R), rnd(M, H), rnd(M, H), rnd(M, R)) for _ in range(NS)]   # dpre, o, dy, h1
splits = libs["A"].mm_gemm_tn_pair_splits(M, R, H, H, R)
mk = lambda *sh: torch.empty(*sh, dtype=torch.bfloat16, device=dev)
out = {t: (mk(R, H), mk(R), mk(H, R), mk(H)) for t in libs}
ws = torch.empty(splits * (R * H + R + H * R + H), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def launch(tag, i):
    a0, b0, a1, b1 = sets[i % NS]
    c0, s0, c1, s1 = out[tag]
    rc = libs[tag].mm_gemm_tn_pair(c0.data_ptr(), s0.data_ptr(), a0.data_ptr(), b0.data_ptr(), R, H, c1.data_ptr(), s1.data_ptr(),
                                   a1.data_ptr(), b1.data_ptr(), H, R, M, ws.data_ptr(), splits, 0, st)
    assert rc == 0, rc


times = {"A": [], "B": []}
j = 0
for rnd_i in range(60):
    for tag in (("A", "B") if rnd_i & 1 else ("B", "A")):
        j += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(tag, j); e1.record()
        torch.cuda.synchronize()
        if rnd_i >= 10:
            times[tag].append(e0.elapsed_time(e1) * 1e3)
launch("A", 0); launch("B", 0); torch.cuda.synchronize()
same = all(torch.equal(x, y) for x, y in zip(out["A"], out["B"]))
print("M=%d h=%d r=%d splits=%d; outputs equal: %s" % (M, H, R, splits, same))
for tag, path in (("A", pa), ("B", pb)):
    v = sorted(times[tag])
    print("%s %-36s median %.1f us  p10 %.1f  p90 %.1f (pair = split kernel + reduce)" % (tag, path, statistics.median(v), v[len(v) // 10], v[-len(v) // 10 - 1]))
