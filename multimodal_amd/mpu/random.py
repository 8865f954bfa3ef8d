"""RNG tracker for the tensor-parallel dropout streams.

The reference re-exports DeepSpeed's tracker (megatron/mpu/random.py:18-37).  This one keeps the same
surface (`get_cuda_rng_tracker().fork()`, `model_parallel_cuda_manual_seed`) but holds a (seed, offset)
Philox counter per named stream instead of swapping the device generator state: the HIP dropout kernels
take (seed, offset) as arguments (include/mm_hip.h, mm_bias_dropout_add), so forking costs nothing and a
recomputed forward replays exactly when the caller restores the offsets (`get_states` / `set_states`).

Device-resident form (`draw_device` / `commit`): the stream's {seed, base offset} also lives in a device uint64[2] that
the dropout kernel reads at run time (mm_bias_dropout_add_devrng); a launch only carries its position inside the step.
`commit()` - one tiny kernel at the end of a training step - moves the base by the number of draws of that step.  A step
captured in a HIP graph therefore draws FRESH masks at every replay, like the eager step, instead of replaying the
(seed, offset) integers that were baked into the captured launches.
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
        self.dev_ = {}      # (name, device) -> uint64[2] device tensor {seed, base offset} (stored as int64)
        self.dev_stamp_ = {}  # (name, device) -> event behind the upload that created it (multimodal_amd/_streams.py)
        self.pending_ = {}  # (name, device) -> draws since the last commit (host side: positions inside the step)

    def add(self, name, seed):
        if name in self.states_:
            raise Exception("cuda rng state {} already exists".format(name))
        self.states_[name] = [int(seed), 0]

    def get_states(self):
        return {"host": {k: list(v) for k, v in self.states_.items()}, "pending": dict(self.pending_)}

    def set_states(self, states):
        self.states_ = {k: list(v) for k, v in states["host"].items()}
        self.pending_ = dict(states["pending"])

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


    def draw_device(self, device, n=1):
        """(state tensor, offset within the step) for the next dropout launch on the active stream: the kernel draws from
        (state[0], state[1] + offset).  The device state is created on first use from the stream's host (seed, offset)."""
        import torch
        from .. import _streams
        key = (self.active_, torch.device(device))
        st = self.dev_.get(key)
        if st is None:
            seed, off = self.states_[self.active_]
            st = self.dev_[key] = torch.tensor([seed, off], dtype=torch.int64, device=device)
            self.dev_stamp_[key] = _streams.stamp(device) if st.is_cuda else None
        else:
            _streams.wait(self.dev_stamp_.get(key), device)  # created on another stream (mpu/overlap.py sub-batches)
        pos = self.pending_.get(key, 0)  # (absent after set_states() of a snapshot taken before the stream's first draw)
        self.pending_[key] = pos + n
        return st, pos

    def commit(self):
        """End of a training step (after the backward: a recompute replays the step's positions): base offset += draws of
        this step, on the device (capturable), and the per-step positions restart at 0."""
        import torch
        from .._lib import check, get_lib
        for key, n in self.pending_.items():
            if n:
                st = self.dev_[key]
                with torch.cuda.device(st.device):
                    check(get_lib().mm_rng_advance(st.data_ptr(), n, torch.cuda.current_stream().cuda_stream), "rng_advance")
                self.pending_[key] = 0


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
