#!/usr/bin/env python3
"""Per-kernel roofline table of the magma_pythia_410M step from a rocprofv3 kernel-stats CSV (+ optional PMC per-kernel CSVs).

    python tools/roofline_table.py profiles/rNN_x_bench_kernel_stats.csv [fetch_by_kernel.csv write_by_kernel.csv] > profiles/rNN_x_roofline.md

Algorithmic work per launch follows SURVEY.md 8(d) / DESIGN.md 4 at T = 8192 tokens, h = 1024, 16 heads x 64, s = 2048, V = 50304,
2-byte dtype.  Kernels that serve several shapes (gemm_nt_128_k<none>) are listed with their launch-weighted mix.
Peaks: 2.5 PFLOP/s dense bf16 MFMA, 8 TB/s HBM3E (MI355X_MICROARCH.md); FETCH_SIZE is doubled (gfx950 correction)."""
import csv, sys

T, H, S, B, NP, HN, V, R = 8192, 1024, 2048, 4, 16, 64, 50304, 128
MiB = 1 << 20
GF = 1e9
attn_f = 4.0 * T * S * H  # full s x s convention; the kernels execute the causal half
# name fragment -> (label, bound, work per launch [flops or bytes], note)
ROWS = [
    ("gemm_nt_pp_k<mm::BF16, 2, false, 256", "h->4h GEMM + bias-GeLU (saves pre-activation), ping-pong 256x256", "mfma", 2.0 * T * 4 * H * H, ""),
    ("gemm_nt_pp_k<mm::BF16, 6, true, 256", "4h->h data-gradient GEMM + GeLU', ping-pong 256x256", "mfma", 2.0 * T * 4 * H * H, ""),
    ("gemm_nt_pp_k<mm::BF16, 8, false, 256", "QKV GEMM + bias + rotary, ping-pong 256x256", "mfma", 2.0 * T * 3 * H * H, ""),
    ("gemm_nt_pp_k<mm::BF16, 0, false, 256, 0, 1>", "LM head GEMM, ping-pong 256x256", "mfma", 2.0 * T * V * H, ""),
    ("gemm_nt_pp_k<mm::BF16, 0, false, 256, 0, 2>", "LM head data gradient (N = h, K = V), ping-pong 256x256 with in-kernel split-K", "mfma",
     2.0 * T * V * H, "round 3; was on gemm_nt_128_k"),
    ("gemm_nt_pp_k<mm::BF16, 0, false, 128", "N = h GEMMs (dense, 4h->h, and the dgrads of QKV / dense / h->4h), ping-pong 256x128", "mfma",
     (24 * 2.0 * T * H * (H + 4 * H + 3 * H + H + 4 * H)) / (24 * 5),
     "launch-weighted mean of 3 shapes, K = 1024 (x2), 3072, 4096 (x2); round 3, third set: was on gemm_nt_128_k"),
    ("gemm_nt_256_k<mm::BF16, 2, false>", "h->4h GEMM + bias-GeLU (saves pre-activation)", "mfma", 2.0 * T * 4 * H * H, ""),
    ("gemm_nt_256_k<mm::BF16, 6, true>", "4h->h data-gradient GEMM + GeLU'", "mfma", 2.0 * T * 4 * H * H, ""),
    ("gemm_nt_128_k<mm::BF16, 8, false>", "QKV GEMM + bias + rotary", "mfma", 2.0 * T * 3 * H * H, ""),
    ("gemm_nt_256_k<mm::BF16, 0, false>", "LM head GEMM", "mfma", 2.0 * T * V * H, ""),
    ("gemm_nt_128_k<mm::BF16, 0, false>", "N = h GEMMs (dense, 4h->h, and the dgrads of QKV / dense / h->4h / head)", "mfma",
     (24 * 2.0 * T * H * (H + 4 * H + 3 * H + H + 4 * H)) / (24 * 5), "launch-weighted mean of 3 shapes, K = 1024 (x2), 3072, 4096 (x2)"),
    ("attn_fwd_k<mm::BF16, 64, 0", "attention forward", "mfma", attn_f, "full s x s convention (causal-exact is half)"),
    ("attn_bwd_dkv_k<mm::BF16, 64, 0", "attention backward dK, dV (+ rotary backward on dK)", "mfma", attn_f * 2.0, "4 of 5 products; convention as above"),
    ("attn_bwd_dq_k<mm::BF16, 64, 0", "attention backward dQ (+ rotary backward)", "mfma", attn_f * 1.5, "3 products (S and dP recomputed)"),
    ("bottleneck_k<mm::BF16, 1024, 128, 0>", "adapter forward", "hbm", 2.0 * (2 * T * H + T * R), "o in, out + h1 out"),
    ("bottleneck_k<mm::BF16, 1024, 128, 1>", "adapter data gradient", "hbm", 2.0 * (2 * T * H + 2 * T * R), "dy + h1 in, dx + dpre out"),
    ("bottleneck_dma_k<mm::BF16, 1024, 128, 0, false>", "adapter forward", "hbm", 2.0 * (2 * T * H + T * R), "o in, out + h1 out"),
    ("bottleneck_dma_k<mm::BF16, 1024, 128, 0, true>", "MLP adapter forward + GPT-J residual tail", "hbm", 2.0 * (4 * T * H + T * R),
     "o, x, attn in, out + h1 out"),
    ("bottleneck_dma_k<mm::BF16, 1024, 128, 1, false>", "adapter data gradient", "hbm", 2.0 * (2 * T * H + 2 * T * R), "dy + h1 in, dx + dpre out"),
    ("gemm_tn_k<mm::BF16, true>", "adapter weight + bias gradients (pair)", "hbm", 2.0 * (2 * T * H + 2 * T * R), "o, dy, dpre, h1 read once"),
    ("splitk_reduce_k", "split-K slab reduce of the pair", "hbm", 16 * 4.0 * 2 * (R * H + H) + 2.0 * 2 * R * H, "16 fp32 slabs in, 2 weights out"),
    ("layernorm_fwd_k<mm::BF16, 2, true>", "dual LayerNorm forward", "hbm", 6.0 * T * H, "x in, two outputs"),
    ("layernorm_bwd_dx_k<mm::BF16, 2, true, true>", "dual LayerNorm backward + residual gradient", "hbm", 10.0 * T * H, "dy1, dy2, dres, x in, dx out"),
    ("gptj_combine_k", "GPT-J residual tail", "hbm", 8.0 * T * H, "x, attn, mlp in, out"),
    ("attn_delta_k", "attention backward delta", "hbm", 4.0 * T * H, "dO, O in"),
    ("ce_fwd_k", "cross entropy forward", "hbm", 2.0 * T * V, "logits in"),
    ("ce_bwd_k", "cross entropy backward (in place)", "hbm", 4.0 * T * V, "logits in, gradient out"),
    ("ce_fwd_grad_k", "cross entropy loss + gradient in one pass (in place)", "hbm", 4.0 * T * V, "logits in, gradient out"),
    ("adam_update_k", "clip + Adam update", "hbm", 28.0 * 13.7e6, "13.7 M parameters x 28 B"),
]

def load_pmc(path):
    out = {}
    if path:
        for r in csv.DictReader(open(path)):
            out[r["Kernel_Name"]] = float(list(r.values())[2])
    return out

stats = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
fetch = load_pmc(sys.argv[2] if len(sys.argv) > 2 else None)
write = load_pmc(sys.argv[3] if len(sys.argv) > 3 else None)
print("| kernel | what | launches | avg us | bound | achieved | frac of peak | HBM traffic / launch (PMC) | algorithmic |")
print("|---|---|---|---|---|---|---|---|---|")
for frag, label, bound, work, note in ROWS:
    hit = [k for k in stats if frag in k]
    if not hit:
        continue
    r = stats[hit[0]]
    us = float(r["AverageNs"]) / 1e3
    if bound == "mfma":
        ach, frac, unit = work / us / 1e6, work / us / 1e6 / 2500.0, "TFLOP/s"
        alg = "%.1f GFLOP" % (work / GF)
    else:
        ach, frac, unit = work / us / 1e6, work / us / 1e6 / 8.0, "TB/s"
        alg = "%.1f MiB" % (work / MiB)
    tr = ""
    fk = [k for k in fetch if frag in k]
    wk = [k for k in write if frag in k]
    if fk and wk:
        tr = "%.1f MiB (fetch x2: %.1f, write %.1f)" % ((2 * fetch[fk[0]] + write[wk[0]]) / 1024, 2 * fetch[fk[0]] / 1024, write[wk[0]] / 1024)
    print("| `%s` | %s | %s | %.1f | %s | %.2f %s | %.3f | %s | %s%s |" % (frag, label, r["Calls"], us, bound, ach, unit, frac, tr, alg, (" - " + note) if note else ""))
