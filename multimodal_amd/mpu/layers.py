"""Tensor-parallel linears with the reference's constructor / forward signatures, parameter names and
partitioning (megatron/mpu/layers.py:42-93, 381-757), computing with the HIP MFMA GEMM.

Layout contract kept from the reference: ColumnParallelLinear.weight is [out/p, in] (partition_dim 0),
RowParallelLinear.weight is [out, in/p] (partition_dim 1), Row biases are NOT sharded, forward returns
``(output, bias_or_None)`` and honours ``skip_bias_add`` / ``gather_output`` / ``parallel_output``.
"""
import torch
import torch.nn.init as init
from torch.nn.parameter import Parameter

from .. import functional as Fn
from .initialize import get_model_parallel_rank, get_model_parallel_world_size
from .mappings import (copy_to_model_parallel_region, gather_from_model_parallel_region,
                       reduce_from_model_parallel_region, scatter_to_model_parallel_region)
from .random import get_cuda_rng_tracker
from .utils import divide


def _device(neox_args):
    if getattr(neox_args, "use_cpu_initialization", False) or not torch.cuda.is_available():
        return torch.device("cpu")
    return torch.device("cuda", torch.cuda.current_device())


def _initialize_affine_weight_gpu(weight, init_method, partition_dim, stride=1):
    """megatron/mpu/layers.py:42-50."""
    weight.model_parallel = True
    weight.partition_dim = partition_dim
    weight.partition_stride = stride
    with get_cuda_rng_tracker().fork():
        init_method(weight)


def _initialize_affine_weight_cpu(neox_args, weight, output_size, input_size, per_partition_size, partition_dim,
                                  init_method, stride=1, return_master_weight=False):
    """Build the master weight on every rank and keep this rank's strided slices (mpu/layers.py:53-93)."""
    weight.model_parallel = True
    weight.partition_dim = partition_dim
    weight.partition_stride = stride
    master_weight = torch.empty(output_size, input_size, dtype=torch.float, requires_grad=False)
    init_method(master_weight)
    master_weight = master_weight.to(dtype=neox_args.params_dtype)
    per_partition_per_stride_size = divide(per_partition_size, stride)
    weight_list = torch.split(master_weight, per_partition_per_stride_size, dim=partition_dim)
    rank, world_size = get_model_parallel_rank(), get_model_parallel_world_size()
    with torch.no_grad():
        weight.copy_(torch.cat(weight_list[rank::world_size], dim=partition_dim))
    return master_weight if return_master_weight else None


class ColumnParallelLinear(torch.nn.Module):
    """Y = X A^T + b with A split along its output dimension (megatron/mpu/layers.py:381-564)."""

    def __init__(self, neox_args, input_size, output_size, bias=True, gather_output=True,
                 init_method=init.xavier_normal_, stride=1, keep_master_weight_for_test=False,
                 skip_bias_add=False, mup_rescale_parameters=False):
        super().__init__()
        if getattr(neox_args, "use_mup", False):
            raise NotImplementedError("muP is outside this path's scope (no listed config uses it)")
        self.input_size = input_size
        self.output_size = output_size
        self.gather_output = gather_output
        world_size = get_model_parallel_world_size()
        self.output_size_per_partition = divide(output_size, world_size)
        self.skip_bias_add = skip_bias_add
        self.init_method = init_method
        self.stride = stride
        dev = _device(neox_args)
        self.weight = Parameter(torch.empty(self.output_size_per_partition, self.input_size, device=dev,
                                            dtype=neox_args.params_dtype))
        self.master_weight = None
        if dev.type == "cpu":
            self.master_weight = _initialize_affine_weight_cpu(
                neox_args, self.weight, self.output_size, self.input_size, self.output_size_per_partition, 0,
                init_method, stride=stride, return_master_weight=keep_master_weight_for_test)
        else:
            _initialize_affine_weight_gpu(self.weight, init_method, partition_dim=0, stride=stride)
        if bias:
            self.bias = Parameter(torch.zeros(self.output_size_per_partition, device=dev, dtype=neox_args.params_dtype))
            self.bias.model_parallel = True
            self.bias.partition_dim = 0
            self.bias.stride = stride
        else:
            self.register_parameter("bias", None)

    def set_parallel_output(self, value: bool):
        assert isinstance(value, bool)
        self.gather_output = not value  # layers.py:543-547

    def forward(self, input_):
        input_parallel = copy_to_model_parallel_region(input_)
        bias = self.bias if not self.skip_bias_add else None
        output_parallel = Fn.linear(input_parallel, self.weight, bias)
        output = gather_from_model_parallel_region(output_parallel) if self.gather_output else output_parallel
        return output, (self.bias if self.skip_bias_add else None)


class RowParallelLinear(torch.nn.Module):
    """Y = X A^T + b with A split along its input dimension (megatron/mpu/layers.py:567-757)."""

    def __init__(self, neox_args, input_size, output_size, bias=True, input_is_parallel=False,
                 init_method=init.xavier_normal_, stride=1, keep_master_weight_for_test=False,
                 skip_bias_add=False, parallel_output=False, mup_rescale_parameters=False):
        super().__init__()
        if getattr(neox_args, "use_mup", False):
            raise NotImplementedError("muP is outside this path's scope (no listed config uses it)")
        self.input_size = input_size
        self.output_size = output_size
        self.input_is_parallel = input_is_parallel
        world_size = get_model_parallel_world_size()
        self.input_size_per_partition = divide(input_size, world_size)
        self.skip_bias_add = skip_bias_add
        self.parallel_output = parallel_output
        self.init_method = init_method
        self.stride = stride
        dev = _device(neox_args)
        self.weight = Parameter(torch.empty(self.output_size, self.input_size_per_partition, device=dev,
                                            dtype=neox_args.params_dtype))
        self.master_weight = None
        if dev.type == "cpu":
            self.master_weight = _initialize_affine_weight_cpu(
                neox_args, self.weight, self.output_size, self.input_size, self.input_size_per_partition, 1,
                init_method, stride=stride, return_master_weight=keep_master_weight_for_test)
        else:
            _initialize_affine_weight_gpu(self.weight, init_method, partition_dim=1, stride=stride)
        if bias:
            self.bias = Parameter(torch.zeros(self.output_size, device=dev, dtype=neox_args.params_dtype))
        else:
            self.register_parameter("bias", None)

    def set_parallel_output(self, parallel_output: bool):
        assert isinstance(parallel_output, bool)
        self.parallel_output = parallel_output

    def forward(self, input_):
        input_parallel = input_ if self.input_is_parallel else scatter_to_model_parallel_region(input_)
        fuse_bias = (not self.skip_bias_add) and self.bias is not None and \
            (self.parallel_output or get_model_parallel_world_size() == 1)
        output_parallel = Fn.linear(input_parallel, self.weight, self.bias if fuse_bias else None)
        output_ = output_parallel if self.parallel_output else reduce_from_model_parallel_region(output_parallel)
        if not self.skip_bias_add:
            if self.bias is not None and not fuse_bias:
                output_ = Fn.bias_dropout_add(output_, self.bias, None, 0.0, False)
            return output_, None
        return output_, self.bias
