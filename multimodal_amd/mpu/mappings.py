"""Autograd collectives of the tensor-parallel region (mirrors megatron/mpu/mappings.py:29-192).

The collectives are ``torch.distributed`` calls on the model-parallel group: RCCL over xGMI with
backend "nccl" on MI355X.  One all-reduce of the whole [s, b, h] activation per call - a single large
message, as xGMI's point-to-point links want - never chunked.
"""
import torch

from .initialize import (get_fp32_allreduce, get_model_parallel_group, get_model_parallel_rank,
                         get_model_parallel_world_size)
from .utils import split_tensor_along_last_dim


def _all_reduce(t, group, op=None):
    """RCCL ("nccl") takes the device tensor as it is.  gloo - the transport of the CPU tests and of one-GPU rehearsals of the
    multi-rank path - stalls on device tensors on these boxes (seen with a 55 MB bucket and with [T] vectors issued from a side
    stream), so with gloo device tensors are staged through host memory."""
    op = torch.distributed.ReduceOp.SUM if op is None else op
    if t.is_cuda and torch.distributed.get_backend(group) == "gloo":
        host = t.cpu()
        torch.distributed.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        torch.distributed.all_reduce(t, op=op, group=group)


def _reduce(input_):
    """All-reduce across the model parallel group (mappings.py:29-48), optional fp32 up-cast for bf16."""
    if get_model_parallel_world_size() == 1:
        return input_
    dt = input_.dtype
    if dt == torch.bfloat16 and get_fp32_allreduce():
        input_ = input_.float()
    _all_reduce(input_, get_model_parallel_group())
    if dt == torch.bfloat16 and get_fp32_allreduce():
        input_ = input_.bfloat16()
    return input_


def _split(input_):
    """Keep this rank's slice of the last dimension (mappings.py:51-76)."""
    world_size = get_model_parallel_world_size()
    if world_size == 1:
        return input_
    return split_tensor_along_last_dim(input_, world_size)[get_model_parallel_rank()].contiguous()


def _gather(input_):
    """All-gather and concatenate along the last dimension (mappings.py:79-107)."""
    world_size = get_model_parallel_world_size()
    if world_size == 1:
        return input_
    dt = input_.dtype
    if dt == torch.bfloat16 and get_fp32_allreduce():
        input_ = input_.float()
    rank = get_model_parallel_rank()
    tensor_list = [torch.empty_like(input_) for _ in range(world_size)]
    tensor_list[rank] = input_
    torch.distributed.all_gather(tensor_list, input_.contiguous(), group=get_model_parallel_group())
    output = torch.cat(tensor_list, dim=input_.dim() - 1).contiguous()
    if dt == torch.bfloat16 and get_fp32_allreduce():
        output = output.bfloat16()
    return output


class _CopyToModelParallelRegion(torch.autograd.Function):
    """identity forward / all-reduce backward (mappings.py:110-123)."""

    @staticmethod
    def forward(ctx, input_):
        return input_

    @staticmethod
    def backward(ctx, grad_output):
        return _reduce(grad_output.contiguous())


class _ReduceFromModelParallelRegion(torch.autograd.Function):
    """all-reduce forward / identity backward (mappings.py:126-139)."""

    @staticmethod
    def forward(ctx, input_):
        return _reduce(input_)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


class _ScatterToModelParallelRegion(torch.autograd.Function):
    """split forward / gather backward (mappings.py:142-155)."""

    @staticmethod
    def forward(ctx, input_):
        return _split(input_)

    @staticmethod
    def backward(ctx, grad_output):
        return _gather(grad_output)


class _GatherFromModelParallelRegion(torch.autograd.Function):
    """gather forward / split backward (mappings.py:158-171)."""

    @staticmethod
    def forward(ctx, input_):
        return _gather(input_)

    @staticmethod
    def backward(ctx, grad_output):
        return _split(grad_output)


def copy_to_model_parallel_region(input_):
    return _CopyToModelParallelRegion.apply(input_)


def reduce_from_model_parallel_region(input_):
    return _ReduceFromModelParallelRegion.apply(input_)


def scatter_to_model_parallel_region(input_):
    return _ScatterToModelParallelRegion.apply(input_)


def gather_from_model_parallel_region(input_):
    return _GatherFromModelParallelRegion.apply(input_)
