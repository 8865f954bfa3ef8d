#!/usr/bin/env python3
"""Same-process A/B of mm_gemm_nt (the library's policy: 128x128 kernel for the N = h class) against mm_gemm_nt_splitk (ping-pong
256x256 kernel, two K-slices per tile) at M = 8192: launches interleaved in blocks of four, rotating operand sets larger than
L2, medians of per-launch HIP-event times (an earlier version ran all of one arm, then all of the other, and took the minimum:
the arm that ran second measured 2-3 us slower even when both arms were the SAME kernel).
    python tools/ab_splitk.py [library]     library: a variant build whose split-K policy starts at a lower K
                                            (make -C multimodal_amd/csrc BUILD=build_ab2 OUT=... EXTRA=-DMM_AB_SPLITK_MINK=2048)"""
import ctypes, os, statistics, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import torch
from multimodal_amd import _lib
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(_lib.__file__), "libmm_hip.so")
L = ctypes.CDLL(os.path.abspath(path))
for n in ("mm_gemm_nt", "mm_gemm_nt_splitk", "mm_gemm_nt_splitk_workspace_bytes"):
    getattr(L, n).restype, getattr(L, n).argtypes = _lib.PROTOTYPES[n]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
M = 8192
for N, K in [(1024, 4096), (1024, 3072), (1024, 1024), (1024, 50304), (2048, 4096)]:
    ws_bytes = L.mm_gemm_nt_splitk_workspace_bytes(M, N, K)
    if not ws_bytes:
        print("N=%d K=%d: not a split-K shape in this build" % (N, K))
        continue
    nb = 3 if K < 20000 else 2
    a = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nb)]
    w = [(torch.randn(N, K, device=dev) * K ** -0.5).bfloat16() for _ in range(nb)]
    c = {m: torch.empty(M, N, device=dev, dtype=torch.bfloat16) for m in (0, 1)}
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

    def launch(mode, i):
        if mode:
            rc = L.mm_gemm_nt_splitk(c[1].data_ptr(), a[i].data_ptr(), w[i].data_ptr(), None, ws.data_ptr(), M, N, K, K, K, N, 0, 0, st)
        else:
            rc = L.mm_gemm_nt(c[0].data_ptr(), None, a[i].data_ptr(), w[i].data_ptr(), None, None, M, N, K, K, K, N, 0, 0, st)
        assert rc == 0, rc
    times = {0: [], 1: []}
    j = 0
    rounds = 12 if K < 20000 else 5
    for r in range(rounds):
        for mode in ((0, 1) if r % 2 == 0 else (1, 0)):
            for rep in range(4):
                j += 1
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(mode, j % nb)
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    times[mode].append(e0.elapsed_time(e1) * 1e3)
    launch(0, 0); launch(1, 0)
    torch.cuda.synchronize()
    err = float((c[0].float() - c[1].float()).abs().max())
    fl = 2.0 * M * N * K
    m0, m1 = statistics.median(times[0]), statistics.median(times[1])
    print("N=%d K=%d: policy %.1f us (%.0f TF/s)  split-K %.1f us (%.0f TF/s; includes its flag-zeroing launch)  max|diff| %.4g" %
          (N, K, m0, fl / m0 / 1e6, m1, fl / m1 / 1e6, err), flush=True)
