"""Data-parallel gradient reduction on RCCL (the reference delegates this to DeepSpeed's engine, configured at
megatron/training.py:659-668 with ZeRO-1 reduce-scatter buckets of 5e8 elements, configs/magma_pythia_410M.yml:39-48).

MI355X choice: xGMI is a point-to-point mesh, so collectives are per-link bound - few, large messages.  All trainable
gradients (27 MB of adapter + prefix-projection grads at the 410M config when the LM is frozen) are flattened into
buckets of `bucket_bytes` and averaged with one all-reduce per bucket.
"""
import torch


def allreduce_gradients(params, group=None, bucket_bytes=256 << 20, average=True, transport_dtype=None, even_if_alone=False):
    """In-place average (or sum) of .grad over the data-parallel group.  No-op for a single-process group.
    transport_dtype (e.g. torch.float32) up-casts the flat bucket for the wire: the fp32_allreduce option of the reference's
    mappings (megatron/mpu/mappings.py:36-46), and what a gloo rehearsal of the multi-rank path needs for 16-bit grads."""
    if group == "single-process" or not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return 0
    world = torch.distributed.get_world_size(group=group)
    if world == 1 and not even_if_alone:  # even_if_alone: a one-rank rehearsal that still wants the collective calls made
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    n_calls = 0
    by_dtype = {}
    for g in grads:
        by_dtype.setdefault(g.dtype, []).append(g)
    for dt, gs in by_dtype.items():
        bucket, size = [], 0
        esz = gs[0].element_size()

        def flush(bucket):
            flat = torch.cat([g.reshape(-1) for g in bucket]) if len(bucket) > 1 else bucket[0].reshape(-1).clone()
            if transport_dtype is not None and flat.dtype != transport_dtype:
                flat = flat.to(transport_dtype)
            if average:
                flat.div_(world)  # pre-divide: keeps 16-bit sums in range
            if flat.is_cuda and torch.distributed.get_backend(group) == "gloo":
                # gloo rehearsal of the multi-rank path on a one-GPU box (its device-tensor ring stalled there on a 55 MB
                # bucket): stage through host memory; RCCL ("nccl") takes the device tensor directly
                host = flat.cpu()
                torch.distributed.all_reduce(host, group=group)
                flat.copy_(host)
            else:
                torch.distributed.all_reduce(flat, group=group)
            views, off = [], 0
            for g in bucket:
                n = g.numel()
                views.append(flat[off:off + n].view_as(g))
                off += n
            torch._foreach_copy_(bucket, views)  # one multi-tensor launch instead of one copy per gradient (~200 at 410M)

        for g in gs:
            if bucket and (size + g.numel()) * esz > bucket_bytes:
                flush(bucket)
                n_calls += 1
                bucket, size = [], 0
            bucket.append(g)
            size += g.numel()
        if bucket:
            flush(bucket)
            n_calls += 1
    return n_calls
