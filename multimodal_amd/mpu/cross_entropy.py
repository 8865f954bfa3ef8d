"""vocab_parallel_cross_entropy (megatron/mpu/cross_entropy.py:27-117) on the HIP cross-entropy kernels.

The reference materialises softmax over its vocabulary partition in fp32 (1.6 GB at V = 50304, T = 8192) and exchanges
three [T] vectors (max, predicted logit, sum-exp).  Here each rank runs the online max / sum-exp kernel on its 16-bit
logits partition (fp32 statistics, nothing materialised), and the same three [T] exchanges combine the partitions:

    lse_r = log sum_{v in partition r} exp(x_v)           (kernel, per rank)
    M = max_r lse_r (all-reduce MAX);  lse = M + log sum_r exp(lse_r - M) (all-reduce SUM)
    loss = lse - sum_r x_target[r]  (all-reduce SUM; the rank that owns the target contributes, the others 0)

Backward: d logits = (exp(x - lse) - onehot_local) * g, written over the logits partition (they are dead afterwards).
With one rank the exchanges disappear and this is functional.CrossEntropyFn.
"""
import torch

from .. import ops
from .initialize import get_model_parallel_group, get_model_parallel_rank, get_model_parallel_world_size
from .mappings import _all_reduce
from .utils import VocabUtility


class _VocabParallelCrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vocab_parallel_logits, target):
        vp = vocab_parallel_logits.size(-1)
        logits2d = vocab_parallel_logits.contiguous().view(-1, vp)
        world = get_model_parallel_world_size()
        start, _ = VocabUtility.vocab_range_from_per_partition_vocab_size(vp, get_model_parallel_rank(), world)
        # labels relative to this partition; out-of-partition targets fall outside [0, vp) and contribute neither a
        # predicted logit nor a one-hot term (the kernels test the range)
        local = (target.reshape(-1) - start).contiguous()
        loss_local, lse_local = ops.cross_entropy_fwd(logits2d, local)
        if world == 1:
            lse, loss = lse_local, loss_local
        else:
            group = get_model_parallel_group()
            predicted = lse_local - loss_local          # x[target] on the owning rank, 0 elsewhere
            m = lse_local.clone()
            _all_reduce(m, group, torch.distributed.ReduceOp.MAX)
            sum_exp = torch.exp(lse_local - m)
            _all_reduce(sum_exp, group)
            _all_reduce(predicted, group)
            lse = m + torch.log(sum_exp)
            loss = lse - predicted
        ctx.save_for_backward(logits2d, local, lse)
        ctx.shape = vocab_parallel_logits.shape
        return loss.view(target.shape)

    @staticmethod
    def backward(ctx, grad_output):
        logits2d, local, lse = ctx.saved_tensors
        g = grad_output.reshape(-1).contiguous().float()
        grad = ops.cross_entropy_bwd(logits2d, local, lse, g, out=logits2d)
        return grad.view(ctx.shape), None


def vocab_parallel_cross_entropy(vocab_parallel_logits, target):
    """Per-token loss (fp32, target's shape) from [..., V/p] 16-bit logits split over the model-parallel group."""
    return _VocabParallelCrossEntropy.apply(vocab_parallel_logits, target)
