"""bias_dropout_add and its getters (megatron/model/fused_bias_dropout.py:28-55) on the HIP kernel."""
from .. import functional as Fn
from ..mpu.random import get_cuda_rng_tracker


def bias_dropout_add(x, bias, residual, prob, training):
    if training and prob > 0:
        # device-resident Philox state: the launch carries only its position inside the step (mpu/random.py)
        state, pos = get_cuda_rng_tracker().draw_device(x.device)
        return Fn.bias_dropout_add(x, bias, residual, prob, training, 0, pos, rng_state=state)
    return Fn.bias_dropout_add(x, bias, residual, prob, training)


def get_bias_dropout_add(training):
    def _bias_dropout_add(x, bias, residual, prob):
        return bias_dropout_add(x, bias, residual, prob, training)

    return _bias_dropout_add


def bias_dropout_add_fused_train(x, bias, residual, prob):
    return bias_dropout_add(x, bias, residual, prob, True)


def bias_dropout_add_fused_inference(x, bias, residual, prob):
    return bias_dropout_add(x, bias, residual, prob, False)
