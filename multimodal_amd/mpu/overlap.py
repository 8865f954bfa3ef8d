"""Overlap of the tensor-parallel all-reduces with compute by splitting the micro-batch (no reference counterpart: the
reference issues one blocking all-reduce per Row linear on the compute stream, megatron/mpu/mappings.py:29-48, so at TP = 8
and h = 4096 its layer waits on 4 x 64 MiB collectives per layer and direction).

MI355X design: xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU) - an all-reduce of the [s, b, h] activation costs
about as much as the layer's GEMMs at TP = 8, and inside ONE dependent chain (LayerNorm -> GEMM ->
all-reduce -> LayerNorm -> ...) nothing can run beside it.  So the micro-batch is split along b into `chunks` sub-batches that
walk the layers in lock step on separate HIP streams: while sub-batch 0 waits for its all-reduce, sub-batch 1's GEMMs and
attention run.  RCCL executes the collectives on its own stream in issue order (the same order on every rank: same program),
each compute stream waits only for its own collective.  Every op of the layer is independent across b (attention works per
(batch element, head); LayerNorm, GEMMs and the elementwise kernels per token row), so activations and input gradients equal
the unsplit run's; a WEIGHT gradient is the sum of the sub-batches' contributions, accumulated by autograd in the parameter's
16-bit dtype, i.e. summed in a different order than the unsplit run's single GEMM (tests/test_tp2_gpu.py allows 2^-6 of the
largest value there; tests/test_mpu_cpu.py checks the schedule over gloo).  The collectives stay outside any HIP-graph capture.
Lazily built device caches shared by the sub-batches (transposed frozen weights, rotary tables, the rng tracker's device
state) carry an event of the stream that filled them and every other stream waits for it (multimodal_amd/_streams.py):
the sub-batch streams are otherwise only ordered at entry and exit.
"""
import torch

_STREAMS = {}


def _chunk_streams(device, n):
    """Stream 0 is the caller's current stream; the others are per-device side streams."""
    if device.type != "cuda":
        return [None] * n
    side = _STREAMS.setdefault(device, [])
    while len(side) < n - 1:
        side.append(torch.cuda.Stream(device=device))
    return [torch.cuda.current_stream(device)] + side[: n - 1]


def split_batch(x, chunks, dim):
    """Contiguous sub-batches of `x` along `dim` (the [s, b, h] activation is split along b = dim 1: a strided view, so copy)."""
    if x.size(dim) % chunks:
        raise ValueError("batch dimension %d is not divisible into %d chunks" % (x.size(dim), chunks))
    return [c.contiguous() for c in torch.chunk(x, chunks, dim=dim)]


def run_interleaved(stages, xs, finish=None):
    """Run `stages` (callables x -> x, e.g. transformer layers) over the sub-batches `xs` stage by stage, sub-batch i on
    stream i, so that a collective inside a stage of one sub-batch overlaps the other sub-batches' compute.  `finish(i, x)`
    (optional) runs on sub-batch i's stream after the last stage.  Returns the list of results; the caller's stream has
    waited for all of them."""
    n = len(xs)
    dev = xs[0].device
    streams = _chunk_streams(dev, n)
    main = streams[0]
    if main is not None:
        for i in range(1, n):
            streams[i].wait_stream(main)
            xs[i].record_stream(streams[i])

    def on(i):
        return torch.cuda.stream(streams[i]) if streams[i] is not None else _Null()

    xs = list(xs)
    for st in stages:
        for i in range(n):
            with on(i):
                xs[i] = st(xs[i])
    outs = []
    for i in range(n):
        with on(i):
            outs.append(finish(i, xs[i]) if finish is not None else xs[i])
    if main is not None:
        for i in range(1, n):
            main.wait_stream(streams[i])
            for t in (outs[i] if isinstance(outs[i], (tuple, list)) else (outs[i],)):
                if torch.is_tensor(t):
                    t.record_stream(main)
    return outs


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
