"""Tensor-level wrappers over the C ABI (``include/mm_hip.h``).

torch is used here for device memory and the current HIP stream only; every computation is a
hand-written gfx950 kernel in ``libmm_hip.so``.  All functions require CUDA(HIP) tensors and raise
``RuntimeError`` on misuse.
"""
import torch

from . import _lib
from ._lib import check, dtype_code, get_lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("multimodal_amd: tensors must live on the GPU (the HIP path has no CPU fallback)")


# ---------------------------------------------------------------------------- softmax
def softmax_causal_fwd(x, scale):
    """[attn_batches, s, s] -> softmax(scale*x) under the implicit causal mask (allocates the output,
    like scaled_upper_triang_masked_softmax_cuda.forward, *_cuda.cu:52-54)."""
    _need_cuda(x)
    if x.dim() != 3 or x.size(1) != x.size(2):
        raise RuntimeError("expected 3D tensor [attn_batches, s, s]")
    x = x.contiguous()
    out = torch.empty_like(x)
    check(get_lib().mm_softmax_causal_fwd(out.data_ptr(), x.data_ptr(), float(scale), x.size(0), x.size(1),
                                          dtype_code(x.dtype), _stream()), "softmax_causal_fwd")
    return out


def softmax_causal_bwd(grad, y, scale):
    """In place into ``grad`` (a contiguous copy is made first if needed, as the reference does)."""
    _need_cuda(grad, y)
    if grad.dim() != 3 or y.shape != grad.shape:
        raise RuntimeError("expected matching 3D tensors")
    grad = grad.contiguous()
    y = y.contiguous()
    check(get_lib().mm_softmax_causal_bwd(grad.data_ptr(), y.data_ptr(), float(scale), grad.size(0), grad.size(1),
                                          dtype_code(grad.dtype), _stream()), "softmax_causal_bwd")
    return grad


def softmax_masked_fwd(x, mask, scale):
    _need_cuda(x, mask)
    if x.dim() != 4 or mask.dim() != 4:
        raise RuntimeError("expected 4D input and mask")
    b, np_, sq, sk = x.shape
    if mask.size(1) != 1 or mask.size(2) != sq or mask.size(3) != sk or mask.size(0) not in (1, b):
        raise RuntimeError(f"mask shape {tuple(mask.shape)} does not match input {tuple(x.shape)}")
    x = x.contiguous()
    m = mask.contiguous()
    if m.dtype == torch.bool:
        m = m.view(torch.uint8)
    elif m.dtype != torch.uint8:
        m = m.to(torch.uint8)
    out = torch.empty_like(x)
    check(get_lib().mm_softmax_masked_fwd(out.data_ptr(), x.data_ptr(), m.data_ptr(), float(scale), b, np_, sq, sk,
                                          m.size(0), dtype_code(x.dtype), _stream()), "softmax_masked_fwd")
    return out


def softmax_masked_bwd(grad, y, scale):
    _need_cuda(grad, y)
    if grad.dim() != 4 or y.shape != grad.shape:
        raise RuntimeError("expected matching 4D tensors")
    grad = grad.contiguous()
    y = y.contiguous()
    b, np_, sq, sk = grad.shape
    check(get_lib().mm_softmax_masked_bwd(grad.data_ptr(), y.data_ptr(), float(scale), b, np_, sq, sk,
                                          dtype_code(grad.dtype), _stream()), "softmax_masked_bwd")
    return grad


def softmax_batch_per_block(sq, sk, b, np_):
    return get_lib().mm_softmax_batch_per_block(sq, sk, b, np_)


# ---------------------------------------------------------------------------- layernorm
def layernorm_fwd(x, w, b, eps, w2=None, b2=None):
    """Returns (y, y2|None, mean, rstd).  x: [..., h]."""
    _need_cuda(x, w, b)
    h = x.size(-1)
    x2 = x.contiguous().view(-1, h)
    rows = x2.size(0)
    y = torch.empty_like(x2)
    y2 = torch.empty_like(x2) if w2 is not None else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(get_lib().mm_layernorm_fwd(y.data_ptr(), _ptr(y2), mean.data_ptr(), rstd.data_ptr(), x2.data_ptr(),
                                     w.data_ptr(), b.data_ptr(), _ptr(w2), _ptr(b2), rows, h, float(eps),
                                     dtype_code(x.dtype), _stream()), "layernorm_fwd")
    return y.view(x.shape), (y2.view(x.shape) if y2 is not None else None), mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, dy2=None, w2=None, dresidual=None, param_grads=True):
    """Returns (dx, [dw, db, (dw2, db2)] or None)."""
    _need_cuda(dy, x, w)
    h = x.size(-1)
    x2 = x.contiguous().view(-1, h)
    rows = x2.size(0)
    dy = dy.contiguous()
    dy2 = dy2.contiguous() if dy2 is not None else None
    dres = dresidual.contiguous() if dresidual is not None else None
    dx = torch.empty_like(x2)
    lib = get_lib()
    nvec = 4 if dy2 is not None else 2
    partial = None
    if param_grads:
        nblk = lib.mm_layernorm_bwd_nblk(rows)
        partial = torch.empty(nblk * nvec * h, dtype=torch.float32, device=x.device)
    check(lib.mm_layernorm_bwd(dx.data_ptr(), _ptr(partial), dy.data_ptr(), _ptr(dy2), _ptr(dres), x2.data_ptr(),
                               w.data_ptr(), _ptr(w2), mean.data_ptr(), rstd.data_ptr(), rows, h, dtype_code(x.dtype),
                               _stream()), "layernorm_bwd")
    grads = _ln_finalize(partial, nblk, nvec, h, x) if param_grads else None
    return dx.view(x.shape), grads


def layernorm_fwd_sb(x, w, b, eps):
    """Final LayerNorm fused with the [s, b, h] -> [b, s, h] transpose (mm_layernorm_fwd_sb): x [s, b, h] contiguous ->
    (y [b, s, h], mean [b*s], rstd [b*s])."""
    _need_cuda(x, w, b)
    if x.dim() != 3 or not x.is_contiguous():
        raise RuntimeError("layernorm_fwd_sb: x must be a contiguous [s, b, h] tensor")
    s, bsz, h = x.shape
    y = torch.empty(bsz, s, h, dtype=x.dtype, device=x.device)
    mean = torch.empty(s * bsz, dtype=torch.float32, device=x.device)
    rstd = torch.empty(s * bsz, dtype=torch.float32, device=x.device)
    check(get_lib().mm_layernorm_fwd_sb(y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), x.data_ptr(), w.data_ptr(), b.data_ptr(),
                                        s, bsz, h, float(eps), dtype_code(x.dtype), _stream()), "layernorm_fwd_sb")
    return y, mean, rstd


def layernorm_bwd_sb(dy, x, w, mean, rstd, dy_scale=None, param_grads=True):
    """dy [b, s, h] (optionally times the device fp32 scalar dy_scale), x [s, b, h] -> (dx [s, b, h], [dw, db] | None)."""
    _need_cuda(dy, x, w, dy_scale)
    s, bsz, h = x.shape
    dy = dy.contiguous()
    if dy.numel() != x.numel() or not x.is_contiguous():
        raise RuntimeError("layernorm_bwd_sb: dy [b, s, h] and a contiguous x [s, b, h] of the same size expected")
    if dy_scale is not None and (dy_scale.dtype != torch.float32 or dy_scale.numel() != 1):
        raise RuntimeError("layernorm_bwd_sb: dy_scale must be a one-element fp32 device tensor")
    dx = torch.empty_like(x)
    lib = get_lib()
    partial, nblk = None, 0
    if param_grads:
        nblk = lib.mm_layernorm_bwd_nblk(s * bsz)
        partial = torch.empty(nblk * 2 * h, dtype=torch.float32, device=x.device)
    check(lib.mm_layernorm_bwd_sb(dx.data_ptr(), _ptr(partial), dy.data_ptr(), _ptr(dy_scale), x.data_ptr(), w.data_ptr(),
                                  mean.data_ptr(), rstd.data_ptr(), s, bsz, h, dtype_code(x.dtype), _stream()), "layernorm_bwd_sb")
    grads = None
    if param_grads:
        grads = _ln_finalize(partial, nblk, 2, h, x)
    return dx, grads


def _ln_finalize(partial, nblk, nvec, h, like):
    import ctypes
    outs = [torch.empty(h, dtype=like.dtype, device=like.device) for _ in range(nvec)]
    arr = (ctypes.c_void_p * 4)(*[o.data_ptr() for o in outs] + [None] * (4 - nvec))
    check(get_lib().mm_layernorm_bwd_finalize(ctypes.cast(arr, ctypes.c_void_p), partial.data_ptr(), nblk, nvec, h,
                                              dtype_code(like.dtype), _stream()), "layernorm_bwd_finalize")
    return outs


# ---------------------------------------------------------------------------- rotary / gelu / bias-dropout-add
def rotary_qk_inplace(qkv, cos, sin, s, b, np_, hn, d, offset=0, inverse=False):
    """qkv: contiguous [s, b, np, 3*hn]; cos/sin: [>=offset+s, d] tables in qkv's dtype."""
    _need_cuda(qkv, cos, sin)
    if not qkv.is_contiguous() or qkv.numel() != s * b * np_ * 3 * hn:
        raise RuntimeError("rotary_qk_inplace: qkv must be contiguous [s, b, np, 3*hn]")
    if cos.dtype != qkv.dtype or sin.dtype != qkv.dtype or cos.size(-1) != d or cos.numel() < (offset + s) * d:
        raise RuntimeError("rotary_qk_inplace: cos/sin tables must be [>=offset+s, d] in the activation dtype")
    check(get_lib().mm_rotary_qk_inplace(qkv.data_ptr(), cos.contiguous().data_ptr(), sin.contiguous().data_ptr(), s, b,
                                         np_, hn, d, offset, int(bool(inverse)), dtype_code(qkv.dtype), _stream()),
          "rotary_qk_inplace")
    return qkv


def bias_gelu_fwd(y, bias, mode):
    _need_cuda(y, bias)
    y = y.contiguous()
    cols = y.size(-1)
    out = torch.empty_like(y)
    check(get_lib().mm_bias_gelu_fwd(out.data_ptr(), y.data_ptr(), bias.data_ptr(), y.numel() // cols, cols, mode,
                                     dtype_code(y.dtype), _stream()), "bias_gelu_fwd")
    return out


def bias_gelu_bwd(g, y, bias, mode, out=None):
    _need_cuda(g, y, bias)
    g = g.contiguous()
    y = y.contiguous()
    cols = y.size(-1)
    out = torch.empty_like(y) if out is None else out
    check(get_lib().mm_bias_gelu_bwd(out.data_ptr(), g.data_ptr(), y.data_ptr(), bias.data_ptr(), y.numel() // cols, cols,
                                     mode, dtype_code(y.dtype), _stream()), "bias_gelu_bwd")
    return out


def bias_dropout_add(x, bias, residual, p, training, seed=0, offset=0, want_mask=False, rng_state=None):
    """rng_state: device int64[2] {seed, base offset} (mpu.get_cuda_rng_tracker().draw_device): the mask is then drawn from
    (rng_state[0], rng_state[1] + offset) read at run time, and `seed` is ignored."""
    _need_cuda(x, bias, residual, rng_state)
    x = x.contiguous()
    cols = x.size(-1)
    if bias is not None:
        if bias.numel() != cols:
            raise RuntimeError("bias_dropout_add: bias must be a [cols] vector (an expanded bias is taken by its last dim)")
        bias = bias.contiguous()
    residual = residual.contiguous() if residual is not None else None
    out = torch.empty_like(x)
    keep = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if (want_mask and training and p > 0) else None
    if rng_state is not None:
        if rng_state.dtype != torch.int64 or rng_state.numel() != 2:
            raise RuntimeError("bias_dropout_add: rng_state must be a device int64[2] tensor {seed, base offset}")
        check(get_lib().mm_bias_dropout_add_devrng(out.data_ptr(), x.data_ptr(), _ptr(bias), _ptr(residual), _ptr(keep),
                                                   x.numel() // cols, cols, float(p), int(bool(training)), rng_state.data_ptr(),
                                                   int(offset), dtype_code(x.dtype), _stream()), "bias_dropout_add_devrng")
    else:
        check(get_lib().mm_bias_dropout_add(out.data_ptr(), x.data_ptr(), _ptr(bias), _ptr(residual), _ptr(keep),
                                            x.numel() // cols, cols, float(p), int(bool(training)), int(seed), int(offset),
                                            dtype_code(x.dtype), _stream()), "bias_dropout_add")
    return (out, keep) if want_mask else out


def colsum(x):
    """[rows, cols] -> [cols] (bias gradient)."""
    _need_cuda(x)
    cols = x.size(-1)
    x2 = x.contiguous().view(-1, cols)
    lib = get_lib()
    partial = torch.empty(lib.mm_colsum_nblk(x2.size(0)) * cols, dtype=torch.float32, device=x.device)
    out = torch.empty(cols, dtype=x.dtype, device=x.device)
    check(lib.mm_colsum(out.data_ptr(), partial.data_ptr(), x2.data_ptr(), x2.size(0), cols, dtype_code(x.dtype), _stream()),
          "colsum")
    return out


def transpose2d(x):
    _need_cuda(x)
    if x.dim() != 2 or x.element_size() != 2:
        raise RuntimeError("transpose2d: expected a 2D tensor of 2-byte elements")
    x = x.contiguous()
    out = torch.empty(x.size(1), x.size(0), dtype=x.dtype, device=x.device)
    check(get_lib().mm_transpose2d(out.data_ptr(), x.data_ptr(), x.size(0), x.size(1), _stream()), "transpose2d")
    return out


def add(a, b, c=None, out=None):
    _need_cuda(a, b, c)
    a, b = a.contiguous(), b.contiguous()
    c = c.contiguous() if c is not None else None
    out = torch.empty_like(a) if out is None else out
    check(get_lib().mm_add(out.data_ptr(), a.data_ptr(), b.data_ptr(), _ptr(c), a.numel(), dtype_code(a.dtype), _stream()),
          "add")
    return out


def relu_bwd(g, act):
    _need_cuda(g, act)
    g, act = g.contiguous(), act.contiguous()
    dx = torch.empty_like(g)
    check(get_lib().mm_relu_bwd(dx.data_ptr(), g.data_ptr(), act.data_ptr(), g.numel(), dtype_code(g.dtype), _stream()), "relu_bwd")
    return dx


def gptj_combine(x, attn, attn_bias, mlp, mlp_bias):
    """[x +] ((attn + attn_bias) + (mlp + mlp_bias)) with the eager ops' roundings; x may be None."""
    _need_cuda(x, attn, attn_bias, mlp, mlp_bias)
    attn, mlp = attn.contiguous(), mlp.contiguous()
    cols = attn.size(-1)
    out = torch.empty_like(attn)
    xx = x.contiguous() if x is not None else None
    check(get_lib().mm_gptj_combine(out.data_ptr(), _ptr(xx), attn.data_ptr(), attn_bias.contiguous().data_ptr(), mlp.data_ptr(),
                                    mlp_bias.contiguous().data_ptr(), attn.numel() // cols, cols, dtype_code(attn.dtype), _stream()),
          "gptj_combine")
    return out


# ---------------------------------------------------------------------------- GEMM
class cu_share:
    """`with ops.cu_share(n):` - scheduling hint for the launches enqueued inside (mm_set_cu_share, include/mm_hip.h): a caller that
    runs two GEMM chains on two HIP streams gives each n = half the CUs.  Thread-local (the autograd engine runs backward nodes
    on its own thread: a node re-applies the value it saw in its forward).  n = 0: no limit."""

    def __init__(self, n):
        self.n = int(n or 0)

    def __enter__(self):
        self.prev = get_lib().mm_set_cu_share(self.n)
        return self

    def __exit__(self, *exc):
        get_lib().mm_set_cu_share(self.prev)
        return False


def current_cu_share():
    lib = get_lib()
    v = lib.mm_set_cu_share(0)
    lib.mm_set_cu_share(v)
    return v


GEMM_SPLITK = True        # plain GEMMs of split-K shapes go to mm_gemm_nt_splitk (False: always mm_gemm_nt; A/B tools flip it)
GEMM_EVENT_PROBE = None  # set by bench.py to {"epilogues": {codes}, "events": []} to time kernel instantiations in situ


def gemm_nt(a, b, bias=None, epilogue=_lib.EPI_NONE, residual=None, out=None, want_aux=False):
    """C[M,N] = A[M,K] @ B[N,K]^T with a fused epilogue.  a: [..., K] (flattened to 2D), b: [N, K].
    Returns C (and the pre-bias product ``aux`` for the GeLU epilogues)."""
    _need_cuda(a, b, bias, residual)
    K = a.size(-1)
    a2 = a.contiguous().view(-1, K)
    b = b.contiguous()
    M, N = a2.size(0), b.size(0)
    if b.size(1) != K:
        raise RuntimeError(f"gemm_nt: inner dimensions differ ({K} vs {b.size(1)})")
    if bias is None and epilogue not in (_lib.EPI_NONE, _lib.EPI_RELU_BWD):
        raise RuntimeError("gemm_nt: this epilogue needs a bias")
    if epilogue in (_lib.EPI_RELU_BWD, _lib.EPI_GELU_TANH_BWD, _lib.EPI_GELU_ERF_BWD) and residual is None:
        raise RuntimeError("gemm_nt: backward epilogues take the saved forward tensor as `residual`")
    if bias is not None and epilogue == _lib.EPI_NONE:
        epilogue = _lib.EPI_BIAS
    c = torch.empty(M, N, dtype=a.dtype, device=a.device) if out is None else out
    if GEMM_SPLITK and epilogue == _lib.EPI_NONE and bias is None and residual is None and not want_aux and c.is_contiguous():
        # too few 256 x 256 tiles for the chip (the N = h class at M = 8192, the LM head's data gradient): two K-slices per
        # tile exchange partials inside the launch (csrc/gemm_pp.hip)
        ws_bytes = get_lib().mm_gemm_nt_splitk_workspace_bytes(M, N, K)
        if ws_bytes:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
            check(get_lib().mm_gemm_nt_splitk(c.data_ptr(), a2.data_ptr(), b.data_ptr(), None, ws.data_ptr(), M, N, K, K, K, N,
                                              epilogue, dtype_code(a.dtype), _stream()), "gemm_nt_splitk")
            return c.view(*a.shape[:-1], N)
    aux = None
    if epilogue in (_lib.EPI_BIAS_GELU_TANH, _lib.EPI_BIAS_GELU_ERF):
        aux = torch.empty(M, N, dtype=a.dtype, device=a.device)
    if residual is not None:
        residual = residual.contiguous()
    probe = GEMM_EVENT_PROBE if (GEMM_EVENT_PROBE is not None and epilogue in GEMM_EVENT_PROBE["epilogues"]) else None
    if probe is not None:  # bench.py: HIP events around this kernel, on the stream it is launched on
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(get_lib().mm_gemm_nt(c.data_ptr(), _ptr(aux), a2.data_ptr(), b.data_ptr(), _ptr(bias), _ptr(residual), M, N, K,
                               K, K, N, epilogue, dtype_code(a.dtype), _stream()), "gemm_nt")
    if probe is not None:
        e1.record()
        probe["events"].append((e0, e1, M, N, K, epilogue))
    c = c.view(*a.shape[:-1], N)
    if aux is not None:
        return c, aux.view(*a.shape[:-1], N)
    return (c, None) if want_aux else c


def gemm_nt_kernel_name(M, N, K, epilogue=_lib.EPI_NONE):
    """Name of the kernel mm_gemm_nt dispatches for a contiguous 16-byte aligned problem of this shape (host-only query)."""
    return get_lib().mm_gemm_nt_kernel_name(M, N, K, epilogue).decode()


def qkv_rotary_supported(hn, d, n_out, k_in, rows, b):
    """Shapes the fused QKV-projection + rotary epilogue takes (mm_gemm_nt_qkv_rotary)."""
    return hn == 64 and d in (16, 32, 64) and n_out % 192 == 0 and k_in % 64 == 0 and rows * b < (1 << 32)


def gemm_nt_qkv_rotary(x, w, bias, cos, sin, b, hn, d, offset=0):
    """rotary(x W^T + bias) on the q / k slices of the fused QKV activation, in the GEMM's epilogue.
    x: [..., K] with rows ordered [s, b]; cos/sin: [>= offset + s, d] tables in x's dtype.  Bit-identical to
    gemm_nt(x, w, bias) followed by rotary_qk_inplace."""
    _need_cuda(x, w, bias, cos, sin)
    K = x.size(-1)
    x2 = x.contiguous().view(-1, K)
    w = w.contiguous()
    M, N = x2.size(0), w.size(0)
    if w.size(1) != K or bias.numel() != N:
        raise RuntimeError("gemm_nt_qkv_rotary: shape mismatch")
    if cos.dtype != x.dtype or sin.dtype != x.dtype or cos.size(-1) != d or cos.numel() < (offset + M // b) * d:
        raise RuntimeError("gemm_nt_qkv_rotary: cos/sin tables must be [>=offset+s, d] in the activation dtype")
    c = torch.empty(M, N, dtype=x.dtype, device=x.device)
    check(get_lib().mm_gemm_nt_qkv_rotary(c.data_ptr(), x2.data_ptr(), w.data_ptr(), bias.contiguous().data_ptr(),
                                          cos.contiguous().data_ptr(), sin.contiguous().data_ptr(), M, N, K, K, K, N, b, hn, d,
                                          offset, dtype_code(x.dtype), _stream()), "gemm_nt_qkv_rotary")
    return c.view(*x.shape[:-1], N)


def gemm_tn(a, b, with_colsum=False):
    """C[N, Kp] = A[M, N]^T @ B[M, Kp] (weight gradient); ``with_colsum`` also returns sum_m A[m, :] (the bias gradient of
    the same backward) computed from the tiles the GEMM stages anyway.  Uses the transposed-read split-K kernel when the
    shape allows, otherwise two HBM transposes + gemm_nt (+ the column-sum kernels)."""
    _need_cuda(a, b)
    a, b = a.contiguous(), b.contiguous()
    M, N = a.shape
    Kp = b.size(1)
    if b.size(0) != M:
        raise RuntimeError("gemm_tn: row counts differ")
    lib = get_lib()
    splits = lib.mm_gemm_tn_splits(M, N, Kp)
    # Model-sized weight gradients (an LM that is NOT frozen: [9216, 2304] at summit-1-3B) are GEMMs of hundreds of GFLOP: two
    # HBM transposes (16-byte-access kernel, ~0.1 ms for the [8192, 9216] activation) and the ping-pong NT kernel at 1.2-1.3 PF
    # beat the split-K kernel's ~0.5 PF there; the adapters' [128, 1024] gradients stay on the split-K kernel.
    if splits == 0 or (N * Kp >= (1 << 21) and M >= 512):
        c = gemm_nt(transpose2d(a), transpose2d(b))
        return (c, colsum(a)) if with_colsum else c
    c = torch.empty(N, Kp, dtype=a.dtype, device=a.device)
    cs = torch.empty(N, dtype=a.dtype, device=a.device) if with_colsum else None
    ws = torch.empty(splits * (N * Kp + N), dtype=torch.float32, device=a.device) if splits > 1 else None
    check(lib.mm_gemm_tn(c.data_ptr(), _ptr(cs), _ptr(ws), a.data_ptr(), b.data_ptr(), M, N, Kp, N, Kp, Kp, splits,
                         dtype_code(a.dtype), _stream()), "gemm_tn")
    return (c, cs) if with_colsum else c


def gemm_tn_pair(a0, b0, a1, b1):
    """(A0^T B0, colsum A0, A1^T B1, colsum A1) for two operand pairs over the same rows, in one launch pair."""
    _need_cuda(a0, b0, a1, b1)
    a0, b0, a1, b1 = a0.contiguous(), b0.contiguous(), a1.contiguous(), b1.contiguous()
    M = a0.size(0)
    if not (b0.size(0) == M and a1.size(0) == M and b1.size(0) == M):
        raise RuntimeError("gemm_tn_pair: row counts differ")
    lib = get_lib()
    N0, K0, N1, K1 = a0.size(1), b0.size(1), a1.size(1), b1.size(1)
    splits = lib.mm_gemm_tn_pair_splits(M, N0, K0, N1, K1)
    if splits == 0:
        c0, s0 = gemm_tn(a0, b0, with_colsum=True)
        c1, s1 = gemm_tn(a1, b1, with_colsum=True)
        return c0, s0, c1, s1
    mk = lambda *sh: torch.empty(*sh, dtype=a0.dtype, device=a0.device)
    c0, s0, c1, s1 = mk(N0, K0), mk(N0), mk(N1, K1), mk(N1)
    ws = torch.empty(splits * (N0 * K0 + N0 + N1 * K1 + N1), dtype=torch.float32, device=a0.device) if splits > 1 else None
    check(lib.mm_gemm_tn_pair(c0.data_ptr(), s0.data_ptr(), a0.data_ptr(), b0.data_ptr(), N0, K0, c1.data_ptr(), s1.data_ptr(),
                              a1.data_ptr(), b1.data_ptr(), N1, K1, M, _ptr(ws), splits, dtype_code(a0.dtype), _stream()),
          "gemm_tn_pair")
    return c0, s0, c1, s1


# ---------------------------------------------------------------------------- adapter (one kernel per direction)
def adapter_supported(h, r):
    return bool(get_lib().mm_adapter_supported(h, r))


def adapter_fwd(o, w1, b1, w2, b2):
    """out = o + relu(o W1^T + b1) W2^T + b2; returns (out, h1) with h1 the saved ReLU output [rows, r]."""
    _need_cuda(o, w1, b1, w2, b2)
    h, r = w1.size(1), w1.size(0)
    o2 = o.contiguous().view(-1, h)
    out = torch.empty_like(o2)
    h1 = torch.empty(o2.size(0), r, dtype=o.dtype, device=o.device)
    check(get_lib().mm_adapter_fwd(out.data_ptr(), h1.data_ptr(), o2.data_ptr(), w1.contiguous().data_ptr(), b1.data_ptr(),
                                   w2.contiguous().data_ptr(), b2.data_ptr(), o2.size(0), h, r, dtype_code(o.dtype), _stream()),
          "adapter_fwd")
    return out.view(o.shape), h1


def adapter_fwd_combine(o, w1, b1, w2, b2, residual, attn, attn_bias, mlp_bias):
    """[residual +] ((attn + attn_bias) + (adapter(o) + mlp_bias)) - the adapter forward with the GPT-J residual tail on its
    output store; returns (out, h1).  Bit-identical to adapter_fwd + gptj_combine."""
    _need_cuda(o, w1, b1, w2, b2, residual, attn, attn_bias, mlp_bias)
    h, r = w1.size(1), w1.size(0)
    o2 = o.contiguous().view(-1, h)
    attn2 = attn.contiguous().view(-1, h)
    res2 = residual.contiguous().view(-1, h) if residual is not None else None
    if attn2.shape != o2.shape or (res2 is not None and res2.shape != o2.shape) or attn_bias.numel() != h or mlp_bias.numel() != h:
        raise RuntimeError("adapter_fwd_combine: shape mismatch")
    out = torch.empty_like(o2)
    h1 = torch.empty(o2.size(0), r, dtype=o.dtype, device=o.device)
    check(get_lib().mm_adapter_fwd_combine(out.data_ptr(), h1.data_ptr(), o2.data_ptr(), w1.contiguous().data_ptr(), b1.data_ptr(),
                                           w2.contiguous().data_ptr(), b2.data_ptr(), _ptr(res2), attn2.data_ptr(),
                                           attn_bias.contiguous().data_ptr(), mlp_bias.contiguous().data_ptr(), o2.size(0), h, r,
                                           dtype_code(o.dtype), _stream()), "adapter_fwd_combine")
    return out.view(o.shape), h1


def adapter_bwd_dx(dy, w2, w1, h1):
    """(dx, dpre) of the adapter: dpre = (dy W2) * (h1 > 0), dx = dy + dpre W1; W2 [h, r] and W1 [r, h] as stored."""
    _need_cuda(dy, w2, w1, h1)
    r, h = w1.shape
    dy2 = dy.contiguous().view(-1, h)
    dx = torch.empty_like(dy2)
    dpre = torch.empty(dy2.size(0), r, dtype=dy.dtype, device=dy.device)
    check(get_lib().mm_adapter_bwd_dx(dx.data_ptr(), dpre.data_ptr(), dy2.data_ptr(), w2.contiguous().data_ptr(),
                                      w1.contiguous().data_ptr(), h1.data_ptr(), dy2.size(0), h, r, dtype_code(dy.dtype),
                                      _stream()), "adapter_bwd_dx")
    return dx, dpre


# ---------------------------------------------------------------------------- attention
def _dropout_args(dropout):
    """dropout = (p, rng_state, pos): rng_state the tracker's device int64[2] {seed, base offset} (mpu/random.py draw_device),
    pos the launch's position inside the step."""
    p, state, pos = dropout
    _need_cuda(state)
    if state.dtype != torch.int64 or state.numel() != 2:
        raise RuntimeError("attention dropout: rng_state must be a device int64[2] {seed, offset}")
    return float(p), state, int(pos)


def attn_dropout_mask(s, b, np_, dropout):
    """The keep mask [b, np, s, s] (uint8, 1 = kept) that attn_fwd / attn_bwd apply for `dropout` = (p, rng_state, pos)."""
    p, state, pos = _dropout_args(dropout)
    keep = torch.empty(b, np_, s, s, dtype=torch.uint8, device=state.device)
    check(get_lib().mm_attn_dropout_mask(keep.data_ptr(), s, b, np_, p, state.data_ptr(), pos, _stream()), "attn_dropout_mask")
    return keep


def attn_fwd(qkv, s, b, np_, hn, softmax_scale, mask=None, dropout=None):
    """qkv: contiguous [s, b, np, 3*hn] (post-rotary).  mask None => implicit causal; otherwise a
    uint8/bool [1|b, 1, s, s] tensor (1 = masked, filled with -10000).  dropout = (p, rng_state, pos): attention dropout
    (see _dropout_args).  Returns (ctx [s, b, np*hn], lse [b, np, s])."""
    _need_cuda(qkv, mask)
    if not qkv.is_contiguous() or qkv.numel() != s * b * np_ * 3 * hn:
        raise RuntimeError("attn_fwd: qkv must be contiguous [s, b, np, 3*hn]")
    ctx = torch.empty(s, b, np_ * hn, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(b, np_, s, dtype=torch.float32, device=qkv.device)
    m, mode, mb = _mask_args(mask, s, b)
    if dropout is not None:
        p, state, pos = _dropout_args(dropout)
        check(get_lib().mm_attn_fwd_dropout(ctx.data_ptr(), lse.data_ptr(), qkv.data_ptr(), _ptr(m), mode, mb, s, b, np_, hn,
                                            float(softmax_scale), p, state.data_ptr(), pos, dtype_code(qkv.dtype), _stream()),
              "attn_fwd_dropout")
        return ctx, lse
    check(get_lib().mm_attn_fwd(ctx.data_ptr(), lse.data_ptr(), qkv.data_ptr(), _ptr(m), mode, mb, s, b, np_, hn,
                                float(softmax_scale), dtype_code(qkv.dtype), _stream()), "attn_fwd")
    return ctx, lse


def attn_bwd(dctx, ctx, lse, qkv, s, b, np_, hn, softmax_scale, mask=None, rotary=None, dropout=None):
    """Gradient wrt qkv [s, b, np, 3*hn].  rotary = (cos, sin, d, offset): also apply the rotary embedding's backward to
    the q / k gradients inside the kernels' stores (the result is then the gradient wrt the PRE-rotary activation)."""
    _need_cuda(dctx, ctx, lse, qkv, mask)
    dctx = dctx.contiguous()
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(b, np_, s, dtype=torch.float32, device=qkv.device)
    m, mode, mb = _mask_args(mask, s, b)
    cos = sin = None
    d = off = 0
    if rotary is not None:
        cos, sin, d, off = rotary
        _need_cuda(cos, sin)
        if cos.dtype != qkv.dtype or sin.dtype != qkv.dtype or cos.size(-1) != d or cos.numel() < (off + s) * d:
            raise RuntimeError("attn_bwd: cos/sin tables must be [>=offset+s, d] in the activation dtype")
        cos, sin = cos.contiguous(), sin.contiguous()
    if dropout is not None:  # the forward's (p, rng_state, pos): the kernels regenerate its mask
        p, state, pos = _dropout_args(dropout)
        check(get_lib().mm_attn_bwd_dropout(dqkv.data_ptr(), delta.data_ptr(), dctx.data_ptr(), ctx.data_ptr(), lse.data_ptr(),
                                            qkv.data_ptr(), _ptr(m), mode, mb, s, b, np_, hn, float(softmax_scale), _ptr(cos),
                                            _ptr(sin), d, off, p, state.data_ptr(), pos, dtype_code(qkv.dtype), _stream()),
              "attn_bwd_dropout")
        return dqkv
    check(get_lib().mm_attn_bwd_rotary(dqkv.data_ptr(), delta.data_ptr(), dctx.data_ptr(), ctx.data_ptr(), lse.data_ptr(),
                                       qkv.data_ptr(), _ptr(m), mode, mb, s, b, np_, hn, float(softmax_scale), _ptr(cos),
                                       _ptr(sin), d, off, dtype_code(qkv.dtype), _stream()), "attn_bwd")
    return dqkv


def _mask_args(mask, s, b):
    if mask is None:
        return None, 0, 1
    if mask.dim() != 4 or mask.size(1) != 1 or mask.size(2) != s or mask.size(3) != s or mask.size(0) not in (1, b):
        raise RuntimeError(f"attention mask must be [1|b, 1, s, s], got {tuple(mask.shape)}")
    m = mask.contiguous()
    m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
    return m, 1, m.size(0)


# ---------------------------------------------------------------------------- cross entropy
def cross_entropy_fwd(logits, labels):
    """logits [T, V] 16-bit, labels [T] int64 -> (loss [T] fp32, lse [T] fp32)."""
    _need_cuda(logits, labels)
    T, V = logits.shape
    loss = torch.empty(T, dtype=torch.float32, device=logits.device)
    lse = torch.empty(T, dtype=torch.float32, device=logits.device)
    check(get_lib().mm_cross_entropy_fwd(loss.data_ptr(), lse.data_ptr(), logits.data_ptr(), labels.data_ptr(), T, V,
                                         dtype_code(logits.dtype), _stream()), "cross_entropy_fwd")
    return loss, lse


def cross_entropy_bwd(logits, labels, lse, row_scale, out=None):
    _need_cuda(logits, labels, lse, row_scale)
    T, V = logits.shape
    out = torch.empty_like(logits) if out is None else out
    check(get_lib().mm_cross_entropy_bwd(out.data_ptr(), logits.data_ptr(), labels.data_ptr(), lse.data_ptr(),
                                         row_scale.data_ptr(), T, V, dtype_code(logits.dtype), _stream()), "cross_entropy_bwd")
    return out


def cross_entropy_fwd_grad_supported(V):
    return V % 8 == 0 and V <= get_lib().mm_cross_entropy_fwd_grad_max_v()


def cross_entropy_fwd_grad(logits, labels, row_weight):
    """One pass: (loss [T], lse [T]) and `logits` OVERWRITTEN with (softmax - onehot) * row_weight[t]."""
    _need_cuda(logits, labels, row_weight)
    T, V = logits.shape
    if not logits.is_contiguous() or row_weight.dtype != torch.float32 or row_weight.numel() != T or labels.numel() != T:
        raise ValueError("cross_entropy_fwd_grad: logits [T, V] contiguous, labels [T], row_weight [T] fp32")
    loss = torch.empty(T, dtype=torch.float32, device=logits.device)
    lse = torch.empty(T, dtype=torch.float32, device=logits.device)
    check(get_lib().mm_cross_entropy_fwd_grad(loss.data_ptr(), lse.data_ptr(), logits.data_ptr(), labels.data_ptr(),
                                              row_weight.data_ptr(), T, V, dtype_code(logits.dtype), _stream()),
          "cross_entropy_fwd_grad")
    return loss, lse


def loss_mask_weights(loss_mask):
    """fp32 [T] loss mask -> (row_weight = mask / sum(mask) [T], den = sum(mask) [1]), one launch (gpt2_model.py:64-70)."""
    _need_cuda(loss_mask)
    lm = loss_mask.reshape(-1)
    lm = lm if lm.dtype == torch.float32 else lm.float()
    lm = lm.contiguous()
    w = torch.empty_like(lm)
    den = torch.empty(1, dtype=torch.float32, device=lm.device)
    check(get_lib().mm_loss_mask_weights(w.data_ptr(), den.data_ptr(), lm.data_ptr(), lm.numel(), _stream()), "loss_mask_weights")
    return w, den, lm


def masked_mean(losses, lm, den):
    """sum(losses * lm) / den as a 0-dim fp32 tensor, one launch."""
    _need_cuda(losses, lm, den)
    out = torch.empty((), dtype=torch.float32, device=losses.device)
    check(get_lib().mm_masked_mean(out.data_ptr(), losses.data_ptr(), lm.data_ptr(), den.data_ptr(), losses.numel(), _stream()),
          "masked_mean")
    return out


def scale_dev(x, scalar, out=None):
    """x * scalar with `scalar` a one-element fp32 DEVICE tensor (no host sync); 16-bit x, numel % 8 == 0."""
    _need_cuda(x, scalar)
    x = x.contiguous()
    if scalar.dtype != torch.float32 or scalar.numel() != 1:
        raise RuntimeError("scale_dev: scalar must be a one-element fp32 device tensor")
    out = torch.empty_like(x) if out is None else out
    check(get_lib().mm_scale_dev(out.data_ptr(), x.data_ptr(), scalar.data_ptr(), x.numel(), dtype_code(x.dtype), _stream()),
          "scale_dev")
    return out


# ---------------------------------------------------------------------------- embedding + image prefix
def prefix_tail_supported(rows, enc_dim, out_dim):
    return bool(get_lib().mm_prefix_tail_supported(rows, enc_dim, out_dim))


def prefix_tail_fwd(x, w, bias, ln_w, ln_b, eps, p=0.0, training=False, rng_state=None, offset=0):
    """LayerNorm(dropout(x W^T + bias)) in one launch (mm_prefix_tail_fwd).  x [rows, enc]; returns
    (out [rows, h], ln_in [rows, h], keep uint8 [rows, h] | None, mean [rows], rstd [rows])."""
    _need_cuda(x, w, bias, ln_w, ln_b, rng_state)
    x = x.contiguous()
    rows, enc = x.shape
    h = w.size(0)
    lib = get_lib()
    drop = bool(training) and p > 0
    if drop and (rng_state is None or rng_state.dtype != torch.int64 or rng_state.numel() != 2):
        raise RuntimeError("prefix_tail_fwd: training with p > 0 needs rng_state, a device int64[2] tensor {seed, base offset}")
    out = torch.empty(rows, h, dtype=x.dtype, device=x.device)
    ln_in = torch.empty(rows, h, dtype=x.dtype, device=x.device)
    keep = torch.empty(rows, h, dtype=torch.uint8, device=x.device) if drop else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.mm_prefix_tail_workspace_bytes(rows, h), dtype=torch.uint8, device=x.device)
    check(lib.mm_prefix_tail_fwd(out.data_ptr(), ln_in.data_ptr(), _ptr(keep), mean.data_ptr(), rstd.data_ptr(), x.data_ptr(),
                                 w.contiguous().data_ptr(), _ptr(bias), ln_w.data_ptr(), ln_b.data_ptr(), ws.data_ptr(), rows, enc, h,
                                 float(eps), float(p), int(bool(training)), _ptr(rng_state), int(offset), dtype_code(x.dtype),
                                 _stream()), "prefix_tail_fwd")
    return out, ln_in, keep, mean, rstd


def dropout_bwd(g, keep, p):
    """keep ? g / (1 - p) : 0."""
    _need_cuda(g, keep)
    g = g.contiguous()
    dx = torch.empty_like(g)
    check(get_lib().mm_dropout_bwd(dx.data_ptr(), g.data_ptr(), keep.data_ptr(), g.numel(), float(p), dtype_code(g.dtype), _stream()),
          "dropout_bwd")
    return dx



def embed_prefix(table, tokens, prefix=None):
    """[s, b, h] input of the first layer: out[si, bi] = prefix[bi, si] for si < P else table[tokens[bi, si - P]]
    (word-embedding gather + image-prefix concat + [b,s,h]->[s,b,h] transpose in one kernel)."""
    _need_cuda(table, tokens, prefix)
    if table.dim() != 2 or table.element_size() != 2 or tokens.dim() != 2 or tokens.dtype != torch.int64:
        raise RuntimeError("embed_prefix: table must be a 16-bit [V, h] tensor and tokens an int64 [b, s] tensor")
    b, s = tokens.shape
    V, h = table.shape
    P = 0
    if prefix is not None:
        if prefix.dim() != 3 or prefix.size(0) != b or prefix.size(2) != h or prefix.dtype != table.dtype:
            raise RuntimeError("embed_prefix: prefix must be [b, P, h] in the table's dtype")
        P = prefix.size(1)
        prefix = prefix.contiguous()
    out = torch.empty(s, b, h, dtype=table.dtype, device=table.device)
    check(get_lib().mm_embed_prefix(out.data_ptr(), table.contiguous().data_ptr(), tokens.contiguous().data_ptr(), _ptr(prefix),
                                    s, b, h, P, V, _stream()), "embed_prefix")
    return out
