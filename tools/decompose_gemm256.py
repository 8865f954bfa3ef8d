#!/usr/bin/env python3
"""Developer tool: where the 256x256 GEMM kernel's time goes.  Runs the debug variants (mm_debug_gemm256) of one shape in one
process and prints the average kernel time of each: full, without the K loop's global loads, without its LDS reads, without the
epilogue, without MFMAs, and combinations."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import _lib
_lib.use_dev_lib().enter()  # developer library (make DEV=1): the kernel-forcing switches / mm_debug_* live there only
lib = _lib.get_lib()
names = {0: "full", 4: "no epilogue", 1: "no loads", 2: "no lds reads", 3: "no loads, no lds reads", 5: "no loads, no epilogue",
         6: "no lds reads, no epilogue", 7: "mfma + barrier only", 8: "no mfma", 12: "no mfma, no epilogue (loads + lds reads)",
         13: "lds reads + barrier only", 14: "loads + barrier only"}
shapes = [(8192, 4096, 1024, 1), (8192, 4096, 1024, 0), (8192, 1024, 4096, 0), (8192, 50432, 1024, 0)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1:5])]
for M, N, K, gelu in shapes:
    a = [(torch.randn(M, K, device="cuda")).bfloat16() for _ in range(2)]
    b = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(2)]
    bias = torch.randn(N, device="cuda").bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    print("M=%d N=%d K=%d gelu=%d  (%.1f GFLOP)" % (M, N, K, gelu, 2.0 * M * N * K / 1e9))
    names.update({16 + k: "pp: " + v for k, v in names.items() if k < 8})
    for dbg in (0, 4, 5, 6, 7, 16, 20, 17, 18, 21, 22, 23, 16, 0):
        def run(i):
            _lib.check(lib.mm_debug_gemm256(c.data_ptr(), aux.data_ptr(), a[i & 1].data_ptr(), b[i & 1].data_ptr(), bias.data_ptr(),
                                            M, N, K, gelu, dbg, st))
        for i in range(5):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for i in range(n):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print("  dbg %2d  %-42s %8.1f us   (%.0f TF/s if it were the full product)" % (dbg, names[dbg], us, 2.0 * M * N * K / us / 1e6))
