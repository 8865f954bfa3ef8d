#!/usr/bin/env python3
"""Same-process A/B of ONE mm_gemm_nt problem on two builds of the library (interleaved launches, cold operands, medians).
    make -C multimodal_amd/csrc BUILD=build_ab OUT=$PWD/ab_tmp/libmm_hip_b.so EXTRA=-DMM_AB_...      (variant B)
    python tools/ab_two_libs.py multimodal_amd/libmm_hip.so ab_tmp/libmm_hip_b.so M N K epilogue [residual]
`residual` = 1 passes a [M, N] tensor in the residual slot (the saved forward tensor of a backward epilogue)."""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import _lib

pa, pb = sys.argv[1:3]
M, N, K, EPI = (int(v) for v in sys.argv[3:7])
RES = len(sys.argv) > 7 and sys.argv[7] == "1"
libs = {}
for tag, path in (("A", pa), ("B", pb)):
    L = ctypes.CDLL(os.path.abspath(path))
    res, args = _lib.PROTOTYPES["mm_gemm_nt"]
    L.mm_gemm_nt.restype, L.mm_gemm_nt.argtypes = res, args
    libs[tag] = L
dev = torch.device("cuda:0")
R = max(2, min(8, (3 << 30) // (2 * (M * K + N * K + 3 * M * N))))
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *sh, scale=1.0: (torch.randn(*sh, device=dev, generator=g) * scale).bfloat16()
a = [rnd(M, K) for _ in range(R)]
w = [rnd(N, K, scale=K ** -0.5) for _ in range(R)]
rs = [rnd(M, N) for _ in range(R)] if RES else [None] * R
bias = rnd(N)
out = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
aux = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
st = torch.cuda.current_stream().cuda_stream
times = {"A": [], "B": []}
j = 0
for rnd_i in range(56):
    for tag in (("A", "B") if (rnd_i & 1) else ("B", "A")):  # alternate the order: the second launch of a pair runs a little faster
        for rep in range(1):
            j += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = libs[tag].mm_gemm_nt(out[j % R].data_ptr(), aux[j % R].data_ptr(), a[j % R].data_ptr(), w[j % R].data_ptr(),
                                      bias.data_ptr(), rs[j % R].data_ptr() if RES else None, M, N, K, K, K, N, EPI, 0, st)
            assert rc == 0, rc
            e1.record()
            torch.cuda.synchronize()
            if rnd_i >= 8:
                times[tag].append(e0.elapsed_time(e1) * 1e3)
o = {}
for tag in ("A", "B"):
    libs[tag].mm_gemm_nt(out[0].data_ptr(), aux[0].data_ptr(), a[0].data_ptr(), w[0].data_ptr(), bias.data_ptr(),
                         rs[0].data_ptr() if RES else None, M, N, K, K, K, N, EPI, 0, st)
    torch.cuda.synchronize()
    o[tag] = out[0].clone()
print("M=%d N=%d K=%d epilogue %d residual %d, %d rotating operand sets; outputs equal: %s" % (M, N, K, EPI, RES, R, torch.equal(o["A"], o["B"])))
for tag, path in (("A", pa), ("B", pb)):
    v = sorted(times[tag])
    print("%s %-40s median %.1f us   p10 %.1f   p90 %.1f   (%d launches)" % (tag, path, statistics.median(v), v[len(v) // 10], v[-len(v) // 10 - 1], len(v)))
