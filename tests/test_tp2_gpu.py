"""Tensor-parallel (TP = 2) layer parity on the GPU: two processes share the one MI355X of the test box and exchange
through gloo (RCCL refuses two ranks on one device; the collective's arithmetic - a 2-term sum - is the same), and each
rank's output / input-gradient / parameter gradients are checked against the fixtures the REFERENCE produced with two
gloo ranks on CPU (tests/golden/layer_*_tp2_r{0,1}.npz), including the GPT-J mode that counts Row biases twice."""
import glob
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, names, q):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import golden_io
    import golden_params
    import layer_harness as LH
    from oracle import neox_layer_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from multimodal_amd import mpu
    mpu.initialize_model_parallel(2, fp32_allreduce=True)  # fp32 transport: exact for a 2-term sum, and gloo-safe
    fails = []
    try:
        for name in names:
            t, meta = golden_io.load(f"{name}_r{rank}")
            dtype = torch.bfloat16
            master = golden_params.master_params(np.random.default_rng(meta["seed"]), meta["h"])
            args = LH.make_neox_args(meta["h"], meta["n"], dtype, meta["rotary_pct"], meta["gpt_j_residual"],
                                     meta["bias_gelu_fusion"], tp=2)
            layer = LH.build_hip_layer(args, master, dev, rank=rank, tp=2)
            y, dx, grads = LH.run_hip_layer(layer, t["x"], t["dy"], dev)
            cfg = O.LayerConfig(meta["h"], meta["n"], rotary_pct=meta["rotary_pct"], gpt_j_residual=meta["gpt_j_residual"],
                                bias_gelu_fusion=meta["bias_gelu_fusion"], tp=2)
            y32, dx32, shards = LH.oracle_layer_fp32(cfg, master, t["x"], t["dy"], dtype, tp=2)

            def chk(what, got, ref16, ref32, factor=2.0):
                got, ref16 = got.float(), ref16.float()
                e_ref = (ref16 - ref32).abs().max().item()
                e_hip = (got - ref32).abs().max().item()
                tol = factor * e_ref + 2.0 ** -7 * ref32.abs().max().item()
                if e_hip > tol:
                    fails.append(f"{name} r{rank} {what}: HIP err {e_hip:.4g} > tol {tol:.4g} (reference-bf16 err {e_ref:.4g})")

            chk("y", y, t["y"], y32)
            chk("dx", dx, t["dx"], dx32)
            for k in ("attention.query_key_value.bias", "mlp.dense_h_to_4h.bias", "attention.dense.bias",
                      "mlp.dense_4h_to_h.bias", "input_layernorm.weight", "attention.dense.weight"):
                if "g." + k not in t:
                    continue
                g32 = shards[rank][k].grad
                if k.endswith("dense.bias") or k.endswith("4h_to_h.bias"):
                    if meta["gpt_j_residual"]:
                        g32 = g32 / 2  # shared tensor in the simulation, one copy per rank in reality
                chk("grad " + k, grads[k], t["g." + k], g32, factor=3.0)
    except Exception as e:  # noqa: BLE001
        import traceback
        fails.append(f"rank {rank}: {type(e).__name__}: {e}\n{traceback.format_exc()}")
    q.put((rank, fails))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_tp2_layer_matches_reference_tp2_fixtures(cuda_device):
    import golden_io
    names = sorted({os.path.basename(p)[:-7] for p in glob.glob(os.path.join(golden_io.HERE, "layer_*_bf16_*_tp2_r0.npz"))})
    assert len(names) >= 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, _free_port_shared(), names, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=100) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    fails = [f for _, fl in results for f in fl]
    assert not fails, "\n".join(fails)


_PORT = None


def _free_port_shared():
    global _PORT
    if _PORT is None:
        _PORT = _free_port()
    return _PORT


def _ce_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from multimodal_amd import mpu
    mpu.initialize_model_parallel(2, fp32_allreduce=True)
    fails = []
    try:
        T, V = 96, 2048
        g = torch.Generator().manual_seed(11)
        logits = (torch.randn(T, V, generator=g) * 3).bfloat16()
        target = torch.randint(0, V, (T,), generator=g)
        target[0], target[1] = 0, V - 1                      # partition edges
        gout = torch.rand(T, generator=g)
        vp = V // world
        mine = logits[:, rank * vp:(rank + 1) * vp].contiguous().to(dev).requires_grad_(True)
        loss = mpu.vocab_parallel_cross_entropy(mine, target.to(dev))
        loss.backward(gout.to(dev))
        x = logits.double().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(x, target, reduction="none")   # == the reference's formula
        ref.backward(gout.double())
        if not torch.allclose(loss.double().cpu(), ref.detach(), rtol=1e-5, atol=1e-5):
            fails.append(f"rank {rank}: loss max err {(loss.double().cpu() - ref.detach()).abs().max().item():.3g}")
        gref = x.grad[:, rank * vp:(rank + 1) * vp]
        err = (mine.grad.double().cpu() - gref).abs().max().item()
        if err > 2.0 ** -8 * gref.abs().max().item() + 1e-6:
            fails.append(f"rank {rank}: dlogits max err {err:.3g}")
    except Exception as e:  # noqa: BLE001
        import traceback
        fails.append(f"rank {rank}: {type(e).__name__}: {e}\n{traceback.format_exc()}")
    q.put((rank, fails))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_tp2_vocab_parallel_cross_entropy(cuda_device):
    """mpu.vocab_parallel_cross_entropy over two vocabulary partitions (two processes, gloo) == full-vocabulary cross
    entropy (the reference's formula, megatron/mpu/cross_entropy.py:27-112) for the loss and for each rank's d logits."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=100) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    fails = [f for _, fl in results for f in fl]
    assert not fails, "\n".join(fails)
