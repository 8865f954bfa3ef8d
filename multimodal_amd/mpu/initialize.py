"""Process-group construction for tensor (model) and data parallelism.

Same rank layout and query API as megatron/mpu/initialize.py:53-293 (model-parallel ranks are adjacent
global ranks: TP group {0..p-1}, {p..2p-1}, ...; DP group = ranks with equal rank % p), built on
``torch.distributed`` - backend "nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests.
Pipeline parallelism is out of scope; a `topology` with pipe > 1 is rejected.
"""
import torch

from .utils import ensure_divisibility

_MODEL_PARALLEL_GROUP = None
_DATA_PARALLEL_GROUP = None
_MODEL_PARALLEL_RANKS = None
_DATA_PARALLEL_RANKS = None
_MPU_WORLD_SIZE = None
_MPU_RANK = None
_FP32_ALLREDUCE = None


def is_unitialized():
    """megatron/mpu/initialize.py:48-50 (sic)."""
    return _DATA_PARALLEL_GROUP is None


def group_layout(world_size, model_parallel_size):
    """Pure function: (list of TP rank lists, list of DP rank lists) - initialize.py:60-66,97-101,151-155."""
    ensure_divisibility(world_size, model_parallel_size)
    tp = [list(range(i * model_parallel_size, (i + 1) * model_parallel_size)) for i in range(world_size // model_parallel_size)]
    dp = [list(range(i, world_size, model_parallel_size)) for i in range(model_parallel_size)]
    return tp, dp


def initialize_model_parallel(model_parallel_size, topology=None, fp32_allreduce=False):
    """megatron/mpu/initialize.py:53-159.  Works without torch.distributed when the world is one process."""
    global _MODEL_PARALLEL_GROUP, _DATA_PARALLEL_GROUP, _MODEL_PARALLEL_RANKS, _DATA_PARALLEL_RANKS, _FP32_ALLREDUCE
    if topology is not None and getattr(topology, "get_dim", lambda _: 1)("pipe") > 1:
        raise NotImplementedError("pipeline parallelism is outside this path's scope (single-node TP x DP only)")
    assert _DATA_PARALLEL_GROUP is None, "data parallel group is already initialized"
    assert _MODEL_PARALLEL_GROUP is None, "model parallel group is already initialized"
    assert _FP32_ALLREDUCE is None, "fp32_allreduce is already initialized"
    have_pg = torch.distributed.is_available() and torch.distributed.is_initialized()
    if have_pg:
        world_size, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
    else:
        world_size, rank = 1, 0
    if world_size < model_parallel_size:
        raise ValueError("world size cannot be smaller than model parallel size")
    tp_lists, dp_lists = group_layout(world_size, model_parallel_size)
    if world_size == 1 and not have_pg:  # (a one-rank process group gets real one-rank groups below)
        _MODEL_PARALLEL_GROUP = _DATA_PARALLEL_GROUP = "single-process"
        _MODEL_PARALLEL_RANKS = _DATA_PARALLEL_RANKS = [0]
    else:
        for ranks in dp_lists:  # every process must create every group, in the same order
            g = torch.distributed.new_group(ranks)
            if rank in ranks:
                _DATA_PARALLEL_GROUP, _DATA_PARALLEL_RANKS = g, ranks
        for ranks in tp_lists:
            g = torch.distributed.new_group(ranks)
            if rank in ranks:
                _MODEL_PARALLEL_GROUP, _MODEL_PARALLEL_RANKS = g, ranks
    _FP32_ALLREDUCE = fp32_allreduce


def model_parallel_is_initialized():
    return _MODEL_PARALLEL_GROUP is not None and _DATA_PARALLEL_GROUP is not None


def get_model_parallel_group():
    assert _MODEL_PARALLEL_GROUP is not None, "model parallel group is not initialized"
    return _MODEL_PARALLEL_GROUP


def get_data_parallel_group():
    assert _DATA_PARALLEL_GROUP is not None, "data parallel group is not initialized"
    return _DATA_PARALLEL_GROUP


def set_model_parallel_world_size(world_size):
    global _MPU_WORLD_SIZE
    _MPU_WORLD_SIZE = world_size


def get_model_parallel_world_size():
    """initialize.py:193-198 (overridable)."""
    if _MPU_WORLD_SIZE is not None:
        return _MPU_WORLD_SIZE
    get_model_parallel_group()
    return len(_MODEL_PARALLEL_RANKS)


def set_model_parallel_rank(rank):
    global _MPU_RANK
    _MPU_RANK = rank


def get_model_parallel_rank():
    """initialize.py:207-212 (overridable)."""
    if _MPU_RANK is not None:
        return _MPU_RANK
    get_model_parallel_group()
    if len(_MODEL_PARALLEL_RANKS) == 1:
        return 0
    return _MODEL_PARALLEL_RANKS.index(torch.distributed.get_rank())


def get_model_parallel_src_rank():
    """initialize.py:215-220."""
    global_rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    p = get_model_parallel_world_size()
    return (global_rank // p) * p


def get_data_parallel_src_rank():
    """initialize.py:223-236 (no pipeline topology)."""
    global_rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    return global_rank % get_model_parallel_world_size()


def get_data_parallel_world_size():
    get_data_parallel_group()
    return len(_DATA_PARALLEL_RANKS)


def get_data_parallel_rank():
    get_data_parallel_group()
    if len(_DATA_PARALLEL_RANKS) == 1:
        return 0
    return _DATA_PARALLEL_RANKS.index(torch.distributed.get_rank())


def get_topology():
    return None


def destroy_model_parallel():
    """initialize.py:269-286."""
    global _MODEL_PARALLEL_GROUP, _DATA_PARALLEL_GROUP, _MODEL_PARALLEL_RANKS, _DATA_PARALLEL_RANKS
    global _MPU_WORLD_SIZE, _MPU_RANK, _FP32_ALLREDUCE
    _MODEL_PARALLEL_GROUP = _DATA_PARALLEL_GROUP = _MODEL_PARALLEL_RANKS = _DATA_PARALLEL_RANKS = None
    _MPU_WORLD_SIZE = _MPU_RANK = _FP32_ALLREDUCE = None


def get_fp32_allreduce():
    assert _FP32_ALLREDUCE is not None, "fp32_allreduce is not Initialized"
    return _FP32_ALLREDUCE
