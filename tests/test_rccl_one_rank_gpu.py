"""The collectives of the multi-GPU path through RCCL itself (backend "nccl") on the one GPU a test box has: a one-rank group.
RCCL refuses two ranks on one device, so the two- and eight-rank tests (tests/test_tp2_gpu.py) travel over gloo; what they
cannot show is that the calls the product makes - bf16 and fp32 all-reduce of device tensors, MAX reduce, the bucketed gradient
all-reduce, all-gather, device barriers - are accepted by RCCL with the tensors the product hands it.  One rank shows that."""
import multiprocessing as mp
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(port, q):
    fails = []
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from multimodal_amd import mpu
        from multimodal_amd.mpu import mappings
        from multimodal_amd.mpu.data_parallel import allreduce_gradients
        mpu.initialize_model_parallel(1)
        g = mpu.get_model_parallel_group()
        torch.manual_seed(0)
        for dt in (torch.bfloat16, torch.float16, torch.float32):
            t = torch.randn(2048, 4, 64, device=dev).to(dt)       # an [s, b, h] activation
            ref = t.clone()
            mappings._all_reduce(t, g)
            if not torch.equal(t, ref):
                fails.append("all_reduce(%s) over one rank changed the tensor" % dt)
        m = torch.randn(8192, device=dev)
        ref = m.clone()
        mappings._all_reduce(m, g, dist.ReduceOp.MAX)
        if not torch.equal(m, ref):
            fails.append("MAX all_reduce changed the tensor")
        # the bucketed gradient all-reduce as bench.py calls it (bf16 on the wire), forced to run although alone
        params = [torch.nn.Parameter(torch.randn(n, device=dev).to(torch.bfloat16)) for n in (1024, 128 * 1024, 7, 1024 * 128)]
        for p in params:
            p.grad = torch.randn_like(p)
        want = [p.grad.clone() for p in params]
        n_calls = allreduce_gradients(params, group=mpu.get_data_parallel_group(), even_if_alone=True, bucket_bytes=300 * 1024)
        if n_calls < 2:
            fails.append("expected several buckets, got %d" % n_calls)
        for p, w in zip(params, want):
            if not torch.equal(p.grad, w):
                fails.append("bucketed all-reduce over one rank changed a gradient")
        x = torch.randn(64, 2, 32, device=dev).to(torch.bfloat16)
        if not torch.equal(mappings._gather(x), x) or not torch.equal(mappings._split(x), x):
            fails.append("gather / split over one rank are not the identity")
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        fails.append("worker: %s\n%s" % (e, traceback.format_exc()))
    q.put(fails)


def test_rccl_one_rank_collectives(cuda_device):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    fails = q.get(timeout=150)
    p.join(timeout=60)
    assert not fails, "\n".join(fails)
