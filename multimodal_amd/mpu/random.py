"""RNG tracker for the tensor-parallel dropout streams.

The reference re-exports DeepSpeed's tracker (megatron/mpu/random.py:18-37).  This one keeps the same
surface (`get_cuda_rng_tracker().fork()`, `model_parallel_cuda_manual_seed`) but holds a (seed, offset)
Philox counter per named stream instead of swapping the device generator state: the HIP dropout kernels
take (seed, offset) as arguments (include/mm_hip.h, mm_bias_dropout_add), so forking costs nothing and a
recomputed forward replays exactly when the caller restores the offsets (`get_states` / `set_states`).
"""
import contextlib

_MODEL_PARALLEL_RNG_TRACKER_NAME = "model-parallel-rng"


class RngStatesTracker:
    def __init__(self):
        self.reset()

    def reset(self):
        self.states_ = {}
        self.active_ = "default"
        self.states_["default"] = [0, 0]

    def add(self, name, seed):
        if name in self.states_:
            raise Exception("cuda rng state {} already exists".format(name))
        self.states_[name] = [int(seed), 0]

    def get_states(self):
        return {k: list(v) for k, v in self.states_.items()}

    def set_states(self, states):
        self.states_ = {k: list(v) for k, v in states.items()}

    @contextlib.contextmanager
    def fork(self, name=_MODEL_PARALLEL_RNG_TRACKER_NAME):
        if name not in self.states_:
            raise Exception("cuda rng state {} is not added".format(name))
        prev, self.active_ = self.active_, name
        try:
            yield
        finally:
            self.active_ = prev

    def draw(self, n=1):
        """(seed, offset) for the next dropout launch on the active stream; advances the offset."""
        st = self.states_[self.active_]
        seed, off = st[0], st[1]
        st[1] += n
        return seed, off


_TRACKER = RngStatesTracker()


def get_cuda_rng_tracker():
    return _TRACKER


def model_parallel_cuda_manual_seed(seed):
    """Default stream: same seed on every TP rank; model-parallel stream: seed + 2718 + tp_rank
    (the DeepSpeed/Megatron convention the reference inherits, megatron/mpu/random.py:30-37)."""
    from .initialize import get_model_parallel_rank
    _TRACKER.reset()
    _TRACKER.states_["default"] = [int(seed), 0]
    _TRACKER.add(_MODEL_PARALLEL_RNG_TRACKER_NAME, int(seed) + 2718 + get_model_parallel_rank())


def checkpoint(function, *args):
    """Activation recompute (the reference's deepspeed.checkpointing.checkpoint, gpt2_model.py:123-132):
    the tracker's counters are rewound for the recompute so dropout masks replay, then put back."""
    import torch.utils.checkpoint as tuc
    start = _TRACKER.get_states()
    calls = [0]

    def run(*a):
        calls[0] += 1
        if calls[0] == 1:
            return function(*a)
        now = _TRACKER.get_states()
        _TRACKER.set_states(start)
        try:
            return function(*a)
        finally:
            _TRACKER.set_states(now)

    return tuc.checkpoint(run, *args, use_reentrant=True, preserve_rng_state=False)
