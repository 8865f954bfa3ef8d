"""Data-parallel gradient reduction on RCCL (the reference delegates this to DeepSpeed's engine, configured at
megatron/training.py:659-668 with ZeRO-1 reduce-scatter buckets of 5e8 elements and `overlap_comm`,
configs/magma_pythia_410M.yml:39-48).

MI355X choice: xGMI is a point-to-point mesh, so collectives are per-link bound - few, large messages.  The trainable
gradients are flattened into buckets of `bucket_bytes` and averaged with one all-reduce per bucket.  Two drivers share ONE
bucket plan (`plan_buckets`: per dtype, parameters in REVERSE registration order - the order their gradients appear in the
backward), so that their results are bit-identical:

  * `allreduce_gradients(params, ...)` - after the backward has returned (what a HIP-graph-replayed step uses: the replay
    launches no Python, so nothing can hook into it; at the 410M recipe with the LM frozen the exchange is 27 MB);
  * `OverlappedGradReducer(params, ...)` - `overlap_comm`: a post-accumulate-grad hook per parameter launches a bucket's
    all-reduce on a communication stream as soon as the bucket's last gradient exists, while the backward of the earlier
    layers is still running on the compute stream; `finish()` joins the streams before the optimizer.  This is what an
    eagerly launched step (every tensor-parallel config: 1.3 GB of gradients per rank at summit-1-3B TP 2 x DP 4) uses.
"""
import torch


def _dist_ready():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def plan_buckets(params, bucket_bytes):
    """[[param, ...], ...]: per gradient dtype (= parameter dtype), parameters in reverse order, cut where a bucket would
    exceed `bucket_bytes`.  Depends only on the parameter list - every rank and both drivers derive the same plan."""
    by_dtype = {}
    for p in reversed(list(params)):
        by_dtype.setdefault(p.dtype, []).append(p)
    buckets = []
    for ps in by_dtype.values():
        cur, size = [], 0
        for p in ps:
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            buckets.append(cur)
    return buckets


def _reduce_bucket(grads, world, group, average, transport_dtype):
    """Average (or sum) the gradients `grads` (one bucket) over the group, in place."""
    flat = torch.cat([g.reshape(-1) for g in grads]) if len(grads) > 1 else grads[0].reshape(-1).clone()
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
    for g in grads:
        n = g.numel()
        views.append(flat[off:off + n].view_as(g))
        off += n
    torch._foreach_copy_(list(grads), views)  # one multi-tensor launch instead of one copy per gradient (~200 at 410M)


def allreduce_gradients(params, group=None, bucket_bytes=256 << 20, average=True, transport_dtype=None, even_if_alone=False):
    """In-place average (or sum) of .grad over the data-parallel group.  No-op for a single-process group.
    transport_dtype (e.g. torch.float32) up-casts the flat bucket for the wire: the fp32_allreduce option of the reference's
    mappings (megatron/mpu/mappings.py:36-46), and what a gloo rehearsal of the multi-rank path needs for 16-bit grads."""
    if group == "single-process" or not _dist_ready():
        return 0
    world = torch.distributed.get_world_size(group=group)
    if world == 1 and not even_if_alone:  # even_if_alone: a one-rank rehearsal that still wants the collective calls made
        return 0
    n_calls = 0
    for bucket in plan_buckets(params, bucket_bytes):
        grads = [p.grad for p in bucket if p.grad is not None]
        if grads:
            _reduce_bucket(grads, world, group, average, transport_dtype)
            n_calls += 1
    return n_calls


class OverlappedGradReducer:
    """Bucketed gradient all-reduce overlapped with the backward (DeepSpeed's `overlap_comm`, which the recipe switches on:
    configs/magma_pythia_410M.yml:39-48).  Same buckets and arithmetic as `allreduce_gradients` - bit-identical results
    (tests/test_mpu_cpu.py, world 2 over gloo).  Usage per step: backward (hooks fire), then `finish()`, then the optimizer.
    The hooks need Python to run during the backward: an eagerly launched step, not a HIP-graph replay."""

    def __init__(self, params, group=None, bucket_bytes=64 << 20, average=True, transport_dtype=None, even_if_alone=False):
        self.params = [p for p in params if p.requires_grad]
        self.group, self.average, self.transport_dtype = group, average, transport_dtype
        self.active = group != "single-process" and _dist_ready()
        self.world = torch.distributed.get_world_size(group=group) if self.active else 1
        if self.world == 1 and not even_if_alone:
            self.active = False
        self.buckets = plan_buckets(self.params, bucket_bytes)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._ready = [0] * len(self.buckets)
        self._done = [False] * len(self.buckets)
        self._comm = None
        self.launched_in_backward = 0  # buckets whose all-reduce was issued from a hook (the rest: in finish())
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params] if self.active else []

    def _comm_stream(self, device):
        if device.type != "cuda":
            return None
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=device)
        return self._comm

    def _launch(self, i):
        grads = [p.grad for p in self.buckets[i] if p.grad is not None]
        self._done[i] = True
        if not grads:
            return
        comm = self._comm_stream(grads[0].device)
        if comm is None:
            _reduce_bucket(grads, self.world, self.group, self.average, self.transport_dtype)
            return
        comm.wait_stream(torch.cuda.current_stream(grads[0].device))  # the bucket's gradients exist on the compute stream
        with torch.cuda.stream(comm):
            _reduce_bucket(grads, self.world, self.group, self.average, self.transport_dtype)
        for g in grads:
            g.record_stream(comm)

    def _on_grad(self, p):
        i = self._bucket_of[id(p)]
        self._ready[i] += 1
        if self._ready[i] == len(self.buckets[i]) and not self._done[i]:
            self._launch(i)
            self.launched_in_backward += 1

    def finish(self):
        """After the backward: reduce what no hook completed (a parameter that received no gradient keeps its bucket open),
        make the compute stream wait for the communication stream, and re-arm for the next step."""
        if not self.active:
            return 0
        for i in range(len(self.buckets)):
            if not self._done[i]:
                self._launch(i)
        if self._comm is not None:
            torch.cuda.current_stream(self._comm.device).wait_stream(self._comm)
        n = len(self.buckets)
        self._ready = [0] * n
        self._done = [False] * n
        return n

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
