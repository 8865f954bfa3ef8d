#!/usr/bin/env python3
"""bench.py - train tokens/s of `magma_pythia_410M` image-prefix training (bf16) on N MI355X of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full training step of the reference's magma_pythia_410M recipe on one synthetic micro-batch per GPU
(4 x 2048 tokens: 1 image-prefix token + 2047 caption tokens): prefix projection + embedding + 24
ParallelTransformerLayers with MAGMA adapters (frozen LM) + final LayerNorm + LM head + masked cross entropy,
backward, data-parallel gradient all-reduce (RCCL), gradient clipping + mixed-precision Adam on the trainable parameters.  Every matmul / attention /
norm / activation / loss kernel is this repo's HIP code (libmm_hip.so); nothing is skipped or cached.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel: the h->4h MFMA GEMM with
the fused bias-GeLU epilogue, timed live with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle's
layer, a bounded sample, on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_FLOPS = 2.5e15   # dense bf16 MFMA, MI355X (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_BPS = 8.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="magma_pythia_410M")
    ap.add_argument("--tp", type=int, default=1, help="tensor-parallel degree (TP ranks innermost, mpu/initialize.py); DP = gpus / tp")
    ap.add_argument("--layers", type=int, default=0, help="REHEARSAL ONLY: override num_layers (the JSON line says so; not a valid bench number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches only (no HIP-graph replay)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    return ap.parse_args()


def roofline_from_probe(cfg, probe, tp=1):
    """Dominant kernel = the h->4h GEMM with the fused bias-GeLU epilogue (one launch per layer forward).  Its launches
    inside the TIMED steps were bracketed by HIP events on the launch stream (ops.GEMM_EVENT_PROBE); average them."""
    from multimodal_amd import ops
    T = cfg["micro_batch"] * cfg["seq_length"]
    h = cfg["hidden_size"]
    N = 4 * h // tp
    # (with tensor parallelism the stack walks the layers in sub-batches - mpu/overlap.py - so a launch covers T / chunks rows)
    hits = [(a.elapsed_time(b), M) for a, b, M, Nn, K, epi in probe["events"] if (Nn, K) == (N, h) and epi == probe["epilogue"]]
    if not hits:
        raise RuntimeError("bench.py: no launch of the roofline kernel (N=%d K=%d, epilogue %d) was seen by the event probe"
                           % (N, h, probe["epilogue"]))
    T = max(M for _, M in hits)
    ms = sorted(t for t, M in hits if M == T)
    avg_ms = sum(ms) / len(ms)
    flops = 2.0 * T * h * N
    achieved = flops / (avg_ms * 1e-3) / 1e12
    label = "%s<bf16, %s> M=%d N=%d K=%d" % (ops.gemm_nt_kernel_name(T, N, h, probe["epilogue"]),
                                               "bias_gelu" if cfg["bias_gelu_fusion"] else "bias_gelu_erf", T, N, h)
    return {"bound": "mfma", "kernel": label,
            "achieved": round(achieved, 2), "peak": PEAK_BF16_FLOPS / 1e12, "unit": "TFLOP/s",
            "frac": round(achieved / (PEAK_BF16_FLOPS / 1e12), 4), "avg_kernel_ms": round(avg_ms, 4),
            "median_kernel_ms": round(ms[len(ms) // 2], 4), "launches_timed": len(ms), "flops_per_launch": flops,
            "traffic": pmc_traffic(label)}


def roofline_by_time(cfg, probe):
    """The kernel class with the largest share of the step's time is not the roofline kernel above but the plain N = h GEMMs
    (attention dense, 4h->h, and the data gradients of QKV / dense / h->4h; `gemm_nt_pp_k<128>`, 256 x 128 tiles, from round 3 on -
    `gemm_nt_128_k` before): the same in-situ HIP events, all of its launches in the timed steps, launch-weighted (total FLOPs /
    total kernel time) and per shape."""
    from multimodal_amd import ops, _lib
    h = cfg["hidden_size"]
    by_shape = {}
    for e0, e1, M, N, K, epi in probe["events"]:
        if epi != _lib.EPI_NONE or N != h or K > 4 * h or K % 64:   # (K > 4h: the LM head's data gradient, a split-K launch of its own)
            continue
        by_shape.setdefault((M, N, K), []).append(e0.elapsed_time(e1))
    if not by_shape:
        return None
    names = sorted({ops.gemm_nt_kernel_name(M, N, K, _lib.EPI_NONE) for (M, N, K) in by_shape})
    flops = sum(2.0 * M * N * K * len(v) for (M, N, K), v in by_shape.items())
    ms = sum(sum(v) for v in by_shape.values())
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "%s<bf16, plain> (the N = h class)" % " / ".join(names), "achieved": round(achieved, 2),
            "peak": PEAK_BF16_FLOPS / 1e12, "unit": "TFLOP/s", "frac": round(achieved / (PEAK_BF16_FLOPS / 1e12), 4),
            "launches_timed": sum(len(v) for v in by_shape.values()), "kernel_ms_total": round(ms, 3),
            "shapes": [{"M": M, "N": N, "K": K, "launches": len(v), "avg_kernel_ms": round(sum(v) / len(v), 4),
                        "tflops": round(2.0 * M * N * K / (sum(v) / len(v) * 1e-3) / 1e12, 1)}
                       for (M, N, K), v in sorted(by_shape.items())]}


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json: FETCH_SIZE x2 (gfx950 correction) +
    WRITE_SIZE, collected with rocprofv3 --pmc in separate runs of this same script); None when not collected for this shape."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)[kernel]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(cfg, budget_s):
    """The CPU oracle's layer (+ adapters, frozen LM), fp32, full layer shape, forward + backward, on the host cores.
    Bounded sample: ONE of the num_layers layers, repeated while the budget lasts; tokens/s is scaled to the whole
    model by layers (the head is left out, which flatters the CPU)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import golden_params
    from oracle import neox_layer_oracle as O
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)))
    torch.set_num_threads(threads)
    h, n, s, b = cfg["hidden_size"], cfg["num_attention_heads"], cfg["seq_length"], cfg["micro_batch"]
    rng = np.random.default_rng(0)
    p = {k: v for k, v in golden_params.master_params(rng, h).items()}
    r = h // cfg["adapter_downsample_factor"]
    if cfg["add_adapters"]:
        for loc in ("attention", "mlp"):
            p[f"{loc}.adapter.0.weight"] = golden_params.rng_t(rng, (r, h), 1e-3).requires_grad_(True)
            p[f"{loc}.adapter.0.bias"] = golden_params.rng_t(rng, (r,), 1e-3).requires_grad_(True)
            p[f"{loc}.adapter.2.weight"] = golden_params.rng_t(rng, (h, r), 1e-3).requires_grad_(True)
            p[f"{loc}.adapter.2.bias"] = golden_params.rng_t(rng, (h,), 1e-3).requires_grad_(True)
    if not cfg["freeze_lm"]:
        for k in golden_params.PARAM_ORDER:
            p[k].requires_grad_(True)
    ocfg = O.LayerConfig(h, n, rotary_pct=cfg["rotary_pct"], gpt_j_residual=cfg["gpt_j_residual"],
                         bias_gelu_fusion=cfg["bias_gelu_fusion"])
    x = golden_params.rng_t(rng, (s, b, h)).requires_grad_(True)
    dy = golden_params.rng_t(rng, (s, b, h))
    mask = O.causal_mask(s)
    times = []
    t_start = time.time()
    while True:
        t0 = time.time()
        y = O.layer_forward(ocfg, [p], x, mask)
        y.backward(dy)
        times.append(time.time() - t0)
        if time.time() - t_start > budget_s or len(times) >= 6:
            break
    t_layer = min(times)
    tok_s = (s * b) / (t_layer * cfg["num_layers"])
    # BASELINE.md section 3 workload A (BASELINE.json configs[0]): the 19M text-only train step, whole model, on the same cores
    from oracle import workload_a as WA
    wl, wt = WA.train_steps(WA.CONFIG_19M, steps=3, budget_s=max(5.0, budget_s * 0.6))
    wa_step = min(wt[1:]) if len(wt) > 1 else wt[0]
    workload_a = {"tokens_per_s": round(WA.CONFIG_19M["micro_batch"] * WA.CONFIG_19M["seq_length"] / wa_step, 1),
                  "s_per_step": round(wa_step, 3), "steps_run": len(wt), "losses": [round(v, 4) for v in wl],
                  "gflops_per_s": round(WA.flops_per_step(WA.CONFIG_19M) / wa_step / 1e9, 1),
                  "what": "19M.yml text-only GPT-NeoX train step (L=6 h=512 heads=8 s=2048 micro_batch=4 V=50304, fp32, eager "
                          "softmax / erf GeLU, Adam) on the CPU oracle stack (oracle/workload_a.py): embedding, 6 layers, final "
                          "LayerNorm, LM head, masked cross entropy, backward, optimizer step"}
    return {"value": round(tok_s, 2), "unit": "tokens/s", "cores": threads, "kind": "port", "workload_a": workload_a,
            "sample": "CPU oracle (oracle/neox_layer_oracle.py, fp32 eager torch) of 1 of %d layers fwd+bwd at the full shape "
                      "(s=%d, b=%d, h=%d), best of %d runs = %.3f s/layer, scaled by layers; head/embedding/optimizer excluded"
                      % (cfg["num_layers"], s, b, h, len(times), t_layer)}


def main():
    a = parse()
    if os.environ.get("MM_BENCH_WATCHDOG"):  # developer aid: dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MM_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # MM_BENCH_BACKEND=gloo is a REHEARSAL switch for one-GPU boxes (RCCL refuses two ranks on one device): ranks share GPUs
    # round-robin and gradients travel as fp32 through gloo.  The driver's multi-GPU runs use the default, RCCL ("nccl").
    backend = os.environ.get("MM_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    from multimodal_amd import harness, mpu
    # MM_BENCH_FORCE_PG=1: rehearsal of the N > 1 control flow with ONE rank (a one-rank RCCL group: communicator set-up, the
    # bucketed gradient all-reduce on device tensors after each graph replay, device barriers) - the one multi-rank path a
    # one-GPU box can run through RCCL itself.  Needs MASTER_ADDR / MASTER_PORT in the environment.
    multi = world > 1 or os.environ.get("MM_BENCH_FORCE_PG") == "1"
    if multi:
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
    if a.tp < 1 or world % a.tp:
        raise SystemExit("bench.py: --tp %d does not divide the %d ranks" % (a.tp, world))
    mpu.initialize_model_parallel(a.tp)  # TP ranks adjacent (innermost), DP groups stride tp: mpu/initialize.py
    mpu.model_parallel_cuda_manual_seed(1234)
    cfg = dict(harness.CONFIGS[a.config])
    if a.layers:
        cfg["num_layers"] = a.layers
    dp_rank, dp_world = mpu.get_data_parallel_rank(), world // a.tp
    torch.manual_seed(1234)  # every replica starts from the same weights (synthetic init; sharded weights repeat across TP ranks)

    model = harness.MagmaStack(cfg, dtype=torch.bfloat16, device=device)
    model.train()
    params = model.trainable_parameters()
    # FusedAdam (fp32 master weights + moments) with the engine's global-norm clipping folded in: magma_setup.yml:38-45,
    # gradient_clipping: 1.0 (magma_pythia_410M.yml:57), megatron/training.py:571-580
    from multimodal_amd.optim import FusedAdam
    opt = FusedAdam(params, lr=cfg.get("lr", 8e-4), betas=(0.9, 0.95), eps=1e-8, weight_decay=cfg.get("weight_decay", 0.0),
                    max_grad_norm=1.0)
    batches = [harness.synthetic_batch(cfg, device, seed=1000 * dp_rank + i) for i in range(4)]  # one batch per TP group
    dp_group = mpu.get_data_parallel_group()

    # static input buffers: the captured graph reads these addresses; each step copies its synthetic batch in first
    static = [t.clone() if t is not None else None for t in batches[0]]

    def load_batch(i):
        for dst, src in zip(static, batches[i % len(batches)]):
            if dst is not None:
                dst.copy_(src)

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)
        loss = model(*static)
        loss.backward()
        harness.end_of_step()  # device-resident dropout counters move on: a replayed step draws fresh masks
        return loss

    # Eagerly launched steps (every tensor-parallel config, or --no-graph) overlap the data-parallel exchange with the backward:
    # a bucket's all-reduce starts on a communication stream as soon as its last gradient exists (DeepSpeed's overlap_comm,
    # configs/magma_pythia_410M.yml:39-48).  A graph-replayed step runs no Python during its backward, so its exchange
    # (27 MB at this recipe) follows the replay; both use the same bucket plan and give the same bits (tests/test_mpu_cpu.py).
    transport = None if backend == "nccl" else torch.float32
    reducer = None
    if multi and (a.tp > 1 or a.no_graph):
        reducer = mpu.OverlappedGradReducer(params, group=dp_group, transport_dtype=transport, even_if_alone=multi)

    def reduce_and_update():
        if reducer is not None:
            reducer.finish()
        else:
            mpu.allreduce_gradients(params, group=dp_group, transport_dtype=transport, even_if_alone=multi)
        opt.step()  # clip + Adam: two launches (csrc/optimizer.hip)

    def step(i):  # eager launches
        load_batch(i)
        loss = fwd_bwd()
        reduce_and_update()
        return loss

    def fence():
        if multi:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    losses = []
    side = torch.cuda.Stream()  # warm-up on a side stream, as graph capture wants (allocator / autograd state settled)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(a.warmup):
            losses.append(float(step(i).detach()))
    torch.cuda.current_stream().wait_stream(side)
    fence()

    # ---- the step as a HIP graph: forward + backward (+ clip + Adam when there is no collective in between) captured once and
    # replayed; with N > 1 ranks the gradient all-reduce (RCCL) and the optimizer's two launches follow the replay eagerly.
    # Replay reproduces the eager step bit for bit (tests/test_graph_gpu.py); it removes the host's per-launch cost (~650
    # launches per step) from the critical path.
    graph, graph_loss, graph_err = None, None, None
    if a.tp > 1:
        graph_err = "tensor-parallel step: the RCCL all-reduces inside the layers stay outside graph capture"
    elif not a.no_graph:
        try:
            load_batch(0)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):  # the warm-up's stream: autograd's AccumulateGrad nodes were created on it
                graph_loss = fwd_bwd()
                if not multi:
                    opt.step()
            fence()  # (capture launches nothing: the gradients hold no values until the first replay)
        except Exception as e:  # fall back to eager launches and say so in the JSON line
            graph, graph_err = None, "%s: %s" % (type(e).__name__, str(e)[:200])
            torch.cuda.synchronize()

    def graph_step(i):
        load_batch(i)
        graph.replay()
        if multi:
            reduce_and_update()
        return graph_loss

    def timed(fn):
        fence()
        t0 = time.perf_counter()
        for i in range(a.steps):
            loss = fn(a.warmup + i)
        fence()
        dt = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, loss

    from multimodal_amd import _lib, ops
    dt_graph = None
    if graph is not None:
        for i in range(2):
            graph_step(i)
        dt_graph, loss = timed(graph_step)
        losses.append(float(loss.detach()))
    dt_plain = None
    if dt_graph is None:  # no graph: the reported value is an eager region with the layer's two streams as configured
        with torch.cuda.stream(side):  # the stream the warm-up ran on (autograd replays each node on its forward stream)
            dt_plain, loss = timed(step)
    # eager launches of the same K steps, with HIP events around the roofline kernel on its launch stream.  The layer keeps
    # to ONE stream here: on two, the attention kernels of the other stream share the CUs with the probed GEMM and the
    # events would time that mix, not the kernel.
    probe = {"epilogue": _lib.EPI_BIAS_GELU_TANH if cfg["bias_gelu_fusion"] else _lib.EPI_BIAS_GELU_ERF, "events": []}
    probe["epilogues"] = {probe["epilogue"], _lib.EPI_NONE}  # + every plain GEMM: the N = h class (roofline_by_time)
    if rank == 0:
        ops.GEMM_EVENT_PROBE = probe
    streams_env = os.environ.get("MM_BRANCH_STREAMS")
    os.environ["MM_BRANCH_STREAMS"] = "0"
    with torch.cuda.stream(side):
        dt_eager, loss = timed(step)
    if streams_env is None:
        del os.environ["MM_BRANCH_STREAMS"]
    else:
        os.environ["MM_BRANCH_STREAMS"] = streams_env
    ops.GEMM_EVENT_PROBE = None
    losses.append(float(loss.detach()))
    # The same probe once more with the layer's streams AS CONFIGURED (two branch streams by default).  In that form - the one the
    # timed region replays - the layer gives each branch half the CUs (ops.cu_share), so the h->4h GEMM runs with 128 workgroups
    # beside the QKV GEMM of the other stream: its own duration is longer than alone on the whole chip, the pair is shorter.
    probe2 = None
    if rank == 0 and os.environ.get("MM_BRANCH_STREAMS", "1") != "0":
        probe2 = {"epilogue": probe["epilogue"], "events": [], "epilogues": {probe["epilogue"]}}
        ops.GEMM_EVENT_PROBE = probe2
        with torch.cuda.stream(side):
            timed(step)
        ops.GEMM_EVENT_PROBE = None
    dt = dt_graph if dt_graph is not None else dt_plain

    tokens_per_step = cfg["micro_batch"] * cfg["seq_length"] * dp_world  # a TP group works on ONE micro-batch
    value = tokens_per_step * a.steps / dt
    flops_step = harness.model_flops_per_step(cfg) / a.tp   # per GPU (weak scaling over DP; TP splits a replica's work)
    flops_causal = harness.model_flops_per_step(cfg, causal_exact=True) / a.tp
    mfma_util = flops_step * a.steps / dt / PEAK_BF16_FLOPS

    if rank == 0:
        roof = roofline_from_probe(cfg, probe, a.tp)
        out = {
            "metric": "train tokens/sec (node) %s" % a.config, "value": round(value, 1), "unit": "tokens/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s training step%s: L=%d h=%d heads=%d s=%d micro_batch=%d/GPU, %s, %s, bf16, no activation recompute"
                                   % (a.config, " (BASELINE.json configs[1])" if a.config == "magma_pythia_410M" else "",
                                      cfg["num_layers"], cfg["hidden_size"], cfg["num_attention_heads"], cfg["seq_length"],
                                      cfg["micro_batch"],
                                      ("%d image-prefix token + text" % cfg["prefix_len"]) if cfg["prefix_len"] else "text only",
                                      ("adapters r=h/%d trainable, LM frozen" % cfg["adapter_downsample_factor"])
                                      if (cfg["add_adapters"] and cfg["freeze_lm"]) else "all parameters trainable"),
                       "global_batch": cfg["micro_batch"] * dp_world, "seq_len": cfg["seq_length"],
                       "parallelism": ("tp%d x dp%d" % (a.tp, dp_world)) if a.tp > 1 else "dp%d" % world},
            "mfma_util_model": round(mfma_util, 4),
            "mfma_util_convention": "useful FLOPs / (time x 2.5 PF); attention counted at the full s x s (Megatron convention, "
                                    "SURVEY 8d); causal-exact (half the attention FLOPs): %.4f" % (flops_causal * a.steps / dt / PEAK_BF16_FLOPS),
            "model_tflops_per_gpu": round(flops_step * a.steps / dt / 1e12, 1),
            "loss_first_last": [round(losses[0], 4) if losses else None, round(losses[-1], 4)],
            "launch": ("hipGraph replay of the captured step (forward + backward%s), %d timed steps; the same %d steps with eager "
                       "launches on one stream: %.3f ms/step (the roofline kernel's HIP events are recorded in those)"
                       % (" + clip + Adam" if not multi else "; all-reduce + clip + Adam follow eagerly", a.steps, a.steps,
                          dt_eager / a.steps * 1e3)) if dt_graph is not None else
                      ("eager launches" + ((" (%s)" if a.tp > 1 else " (graph capture failed: %s)") % graph_err if graph_err else "")),
            "eager_ms_per_step": round(dt_eager / a.steps * 1e3, 3),
            "roofline": roof,
            "roofline_by_time": roofline_by_time(cfg, probe),
        }
        if probe2 is not None and probe2["events"]:
            r2 = roofline_from_probe(cfg, probe2, a.tp)
            share = int(os.environ.get("MM_CU_SHARE", str(torch.cuda.get_device_properties(device).multi_processor_count // 2)).split(",")[-1])
            cus = torch.cuda.get_device_properties(device).multi_processor_count
            out["roofline_in_step"] = {
                "what": "the same launch as `roofline`, timed in an eager region with the layer's two branch streams: it walks its tiles "
                        "with %d workgroups (mm_set_cu_share) beside the QKV GEMM on the other stream; `frac` is against the whole "
                        "chip's peak, `frac_of_share` against the peak of the CUs it occupies" % (share if 0 < share < cus else cus),
                "workgroups": share if 0 < share < cus else cus, "achieved": r2["achieved"], "peak": r2["peak"], "unit": "TFLOP/s",
                "frac": r2["frac"], "frac_of_share": round(r2["frac"] * cus / (share if 0 < share < cus else cus), 4),
                "avg_kernel_ms": r2["avg_kernel_ms"], "launches_timed": r2["launches_timed"]}
        if a.layers:
            out["rehearsal"] = "num_layers overridden to %d: NOT a valid bench number" % a.layers
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_baseline_seconds)
        print(json.dumps(out), flush=True)
    if multi:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
