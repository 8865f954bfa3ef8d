"""Optimizer step of the training recipe on the HIP path: mixed-precision Adam + global-norm clipping in two launches.

Mirrors what the reference builds for `magma_pythia_410M`: apex / DeepSpeed ``FusedAdam(params, lr, betas, eps,
weight_decay, adam_w_mode=True, bias_correction=True)`` (megatron/training.py:571-580) wrapped by DeepSpeed's 16-bit
optimizer (fp32 master weights, fp32 moments, 16-bit model weights) with ``gradient_clipping`` on the global L2 norm
(configs/magma_pythia_410M.yml:57).  Those libraries are not part of the reference tree, so this states their published
update rule (parity unpinned); ``tests/test_optim_gpu.py`` checks it against an fp32 torch restatement.
"""
import numpy as np
import torch

from ._lib import check, dtype_code, get_lib


class FusedAdam:
    """Constructor arguments follow apex.optimizers.FusedAdam; ``max_grad_norm`` (> 0) adds the engine's gradient
    clipping to the same two launches.  ``param_groups`` is a one-entry list so that lr schedulers can set ``lr``."""

    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-8, adam_w_mode=True,
                 weight_decay=0.0, max_grad_norm=0.0):
        params = [p for p in params]
        if not params:
            raise ValueError("FusedAdam: empty parameter list")
        dt, dev = params[0].dtype, params[0].device
        if dev.type != "cuda":
            raise RuntimeError("multimodal_amd.optim.FusedAdam: parameters must live on the GPU (the HIP path has no CPU fallback)")
        for p in params:
            if p.dtype != dt or p.device != dev or not p.is_contiguous():
                raise RuntimeError("FusedAdam: parameters must share one 16-bit dtype and device and be contiguous")
        if weight_decay != 0.0 and not adam_w_mode:
            raise NotImplementedError("FusedAdam: L2-mode weight decay (adam_w_mode=False) is not implemented; the recipe uses 0")
        self._dtype = dtype_code(dt)
        self.param_groups = [dict(params=params, lr=lr, bias_correction=bias_correction, betas=tuple(betas), eps=eps,
                                  weight_decay=weight_decay, max_grad_norm=max_grad_norm)]
        lib = get_lib()
        chunk = lib.mm_adam_chunk_elems()
        # flat fp32 state: every tensor starts at a multiple of 8 elements
        rec = np.dtype([("tensor", "<i4"), ("count", "<i4"), ("toff", "<i8"), ("foff", "<i8")])
        rows, flat = [], 0
        for ti, p in enumerate(params):
            n = p.numel()
            for off in range(0, n, chunk):
                rows.append((ti, min(chunk, n - off), off, flat + off))
            flat += (n + 7) // 8 * 8
        table = np.array(rows, dtype=rec)
        self._nchunks = len(rows)
        self._chunks = torch.from_numpy(table.view(np.uint8).copy()).to(dev)
        self.master = torch.zeros(flat, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(flat, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(flat, dtype=torch.float32, device=dev)
        off = 0
        for p in params:  # fp32 master copy of the 16-bit weights
            n = p.numel()
            self.master[off:off + n].copy_(p.detach().reshape(-1).float())
            off += (n + 7) // 8 * 8
        self._param_ptrs = torch.tensor([p.data_ptr() for p in params], dtype=torch.int64).to(dev)
        # pinned staging for the per-step gradient pointer table: two buffers, each reused only after the copy that read
        # it has executed (the host can run a whole step ahead of the GPU)
        self._grad_ptrs_host = [torch.zeros(len(params), dtype=torch.int64).pin_memory() for _ in range(2)]
        self._capture_host = torch.zeros(len(params), dtype=torch.int64).pin_memory()  # see step(): HIP-graph capture
        self._host_events = [None, None]
        self._flip = 0
        self._grad_ptrs = torch.zeros(len(params), dtype=torch.int64, device=dev)
        self._partial = torch.empty(self._nchunks, dtype=torch.float32, device=dev)
        self._step = torch.zeros(1, dtype=torch.int32, device=dev)
        # the learning rate lives in device memory (the update kernel reads it at run time), so that a step captured in a HIP
        # graph follows a scheduler: step() uploads param_groups[0]["lr"] when it changed; before a graph REPLAY call sync_lr()
        self._lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        # ring of pinned slots for the uploads: a slot is rewritten only after the copy that last read it has completed (its
        # event), so a scheduler that changes the rate EVERY step never makes the host wait for the step in flight
        self._lr_host = torch.full((8,), float(lr), dtype=torch.float32).pin_memory()
        self._lr_events = [None] * 8
        self._lr_slot = 0
        self._lr_uploaded = float(lr)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)  # pre-clip global norm of the last step

    @property
    def params(self):
        return self.param_groups[0]["params"]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def sync_lr(self):
        """Upload param_groups[0]["lr"] to the device scalar the update kernel reads (no-op when unchanged).  step() calls it;
        call it yourself before replaying a captured step after a scheduler changed the learning rate."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_uploaded:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdam: the learning rate changed inside a graph capture; call sync_lr() before capturing")
            i = self._lr_slot = (self._lr_slot + 1) % len(self._lr_events)
            if self._lr_events[i] is not None:
                self._lr_events[i].synchronize()  # the upload from this slot eight changes ago: long done, returns at once
            self._lr_host[i] = lr
            self._lr_dev.copy_(self._lr_host[i:i + 1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._lr_events[i] = ev
            self._lr_uploaded = lr

    def refresh_master_from_params(self):
        """master <- the parameters' CURRENT values.  The fp32 master copy is taken at construction; when weights are loaded
        into the model afterwards (the reference builds the optimizer before load_checkpoint, megatron/training.py:646,690, and
        relies on the DeepSpeed engine to refresh its fp32 copies) call this, or the first step() writes the stale master back
        over the loaded weights.  `checkpointing.load_checkpoint(..., optimizer=opt)` does it."""
        off = 0
        for p in self.params:
            n = p.numel()
            self.master[off:off + n].copy_(p.detach().reshape(-1).float())
            off += (n + 7) // 8 * 8

    def step(self):
        self.sync_lr()
        g = self.param_groups[0]
        params = g["params"]
        self._held = []  # keep this step's gradient tensors alive until the next step (the launches are asynchronous)
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            # the copy below becomes a memcpy node that re-reads this pinned buffer at every replay; the gradients live in
            # the graph's private pool (same addresses every replay), so the table stays valid: a buffer of its own
            host, ev = self._capture_host, None
        else:
            self._flip ^= 1
            host, ev = self._grad_ptrs_host[self._flip], self._host_events[self._flip]
        if ev is not None:
            ev.synchronize()
        ptrs = []
        for p in params:
            gr = p.grad
            if gr is None:
                ptrs.append(0)
                continue
            if gr.dtype != p.dtype or not gr.is_contiguous():
                gr = gr.to(p.dtype).contiguous()
            self._held.append(gr)
            ptrs.append(gr.data_ptr())
        host.numpy()[:] = ptrs
        self._grad_ptrs.copy_(host, non_blocking=True)
        if not capturing:
            ev = torch.cuda.Event()
            ev.record()
            self._host_events[self._flip] = ev
        if self._param_ptrs.numel() != len(params):
            raise RuntimeError("FusedAdam: parameter list changed after construction")
        b1, b2 = g["betas"]
        check(get_lib().mm_adam_step(self._param_ptrs.data_ptr(), self._grad_ptrs.data_ptr(), self.master.data_ptr(),
                                     self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self._chunks.data_ptr(), self._nchunks,
                                     self._partial.data_ptr(), self._step.data_ptr(), self.grad_norm.data_ptr(), float(g["lr"]),
                                     self._lr_dev.data_ptr(), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), float(g["max_grad_norm"]),
                                     int(bool(g["bias_correction"])), self._dtype, torch.cuda.current_stream().cuda_stream),
              "adam_step")

    def state_dict(self):
        return dict(step=int(self._step.item()), master=self.master.clone(), exp_avg=self.exp_avg.clone(),
                    exp_avg_sq=self.exp_avg_sq.clone(), param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self._step.fill_(int(sd["step"]))
        self.master.copy_(sd["master"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0].update(sd["param_groups"][0])
        off = 0
        for p in self.params:  # the 16-bit weights follow the restored master copy
            n = p.numel()
            p.data.copy_(self.master[off:off + n].view_as(p))
            off += (n + 7) // 8 * 8
