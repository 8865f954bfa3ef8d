"""mpu utilities (mirrors megatron/mpu/utils.py:22-74)."""
import torch


def ensure_divisibility(numerator, denominator):
    """megatron/mpu/utils.py:22-26."""
    assert numerator % denominator == 0, "{} is not divisible by {}".format(numerator, denominator)


def divide(numerator, denominator):
    """megatron/mpu/utils.py:29-33."""
    ensure_divisibility(numerator, denominator)
    return numerator // denominator


def split_tensor_along_last_dim(tensor, num_partitions, contiguous_split_chunks=False):
    """megatron/mpu/utils.py:36-53: views unless contiguous_split_chunks."""
    last_dim = tensor.dim() - 1
    last_dim_size = divide(tensor.size()[last_dim], num_partitions)
    tensor_list = torch.split(tensor, last_dim_size, dim=last_dim)
    if contiguous_split_chunks:
        return tuple(chunk.contiguous() for chunk in tensor_list)
    return tensor_list


class VocabUtility:
    """megatron/mpu/utils.py:56-74."""

    @staticmethod
    def vocab_range_from_per_partition_vocab_size(per_partition_vocab_size, rank, world_size):
        index_f = rank * per_partition_vocab_size
        return index_f, index_f + per_partition_vocab_size

    @staticmethod
    def vocab_range_from_global_vocab_size(global_vocab_size, rank, world_size):
        return VocabUtility.vocab_range_from_per_partition_vocab_size(divide(global_vocab_size, world_size), rank, world_size)
