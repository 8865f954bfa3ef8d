"""Both drop-in levels of the fused-softmax boundary.

Level 1: after `load_fused_kernels()` the modules the reference imports BY NAME exist with its function signatures and
conventions (megatron/fused_kernels/__init__.py:32-45, scaled_*_softmax.cpp, *_cuda.cu).  Level 2: `FusedScaleMaskSoftmax`
with the reference's three fusion types and eligibility rule (megatron/model/fused_softmax.py:92-205).  The bar is the
reference's own fused-vs-eager test (tests/model/test_fused_kernels.py:122-142): mean |diff| <= 1e-3 against the eager
path - here against outputs the REFERENCE's eager path produced (tests/golden/op_softmax_eager_*.npz) - plus ulp-closeness
to the oracle's restatement of the CUDA kernels."""
import glob
import os

import pytest
import torch

import golden_io
from oracle import neox_layer_oracle as O

pytestmark = pytest.mark.gpu

EAGER = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(golden_io.HERE, "op_softmax_eager_*16_*.npz")))


def _mask_func(scores, mask):  # gpt2_attention_mask_func (megatron/model/gpt2_model.py:47-49)
    scores.masked_fill_(mask, -10000.0)
    return scores


def _ulp_close(a, b, dtype, ulps=2):
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    a, b = a.float().cpu(), b.float().cpu()
    assert ((a - b).abs() <= ulps * eps * b.abs() + 1e-6).all(), float((a - b).abs().max())


def test_level1_modules_by_import_name(cuda_device):
    from multimodal_amd.fused_kernels import load_fused_kernels
    load_fused_kernels()
    import scaled_masked_softmax_cuda
    import scaled_upper_triang_masked_softmax_cuda
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(8, 64, 64, generator=g) * 3).half().to(cuda_device)
    scale = torch.tensor([0.5])  # the reference passes scale_t[0], a 0-dim CPU tensor (fused_softmax.py:36-38)
    y = scaled_upper_triang_masked_softmax_cuda.forward(x, scale[0])
    assert y.data_ptr() != x.data_ptr() and y.dtype == x.dtype and not y.requires_grad   # allocates (…_cuda.cu:52-54)
    _ulp_close(y, O.softmax_causal_fused_fwd(x.cpu(), 0.5), torch.float16)
    dy = torch.randn(8, 64, 64, generator=g).half().to(cuda_device)
    want = O.softmax_fused_bwd(dy.cpu(), y.cpu(), 0.5, causal=True)
    got = scaled_upper_triang_masked_softmax_cuda.backward(dy, y, scale[0])
    assert got.data_ptr() == dy.data_ptr()            # in place into output_grads, which is returned (…_cuda.cu:77-87)
    _ulp_close(got, want, torch.float16, ulps=4)
    # a strided gradient is made contiguous first (the reference's .contiguous()), then overwritten
    dy2 = torch.randn(8, 64, 128, generator=g).half().to(cuda_device)[:, :, ::2]
    got2 = scaled_upper_triang_masked_softmax_cuda.backward(dy2, y, scale[0])
    assert got2.is_contiguous() and got2.shape == y.shape

    xm = (torch.randn(2, 4, 16, 96, generator=g) * 3).bfloat16().to(cuda_device)
    mask = (torch.rand(2, 1, 16, 96, generator=g) < 0.3).to(cuda_device)
    ym = scaled_masked_softmax_cuda.forward(xm, mask, scale[0])
    _ulp_close(ym, O.softmax_masked_fused_fwd(xm.cpu(), mask.cpu(), 0.5), torch.bfloat16)
    ym1 = scaled_masked_softmax_cuda.forward(xm, mask[:1], scale[0])          # pad_batches == 1: mask broadcast over the batch
    _ulp_close(ym1, O.softmax_masked_fused_fwd(xm.cpu(), mask[:1].cpu(), 0.5), torch.bfloat16)
    dm = torch.randn(2, 4, 16, 96, generator=g).bfloat16().to(cuda_device)
    wantm = O.softmax_fused_bwd(dm.cpu(), ym.cpu(), 0.5, causal=False)
    gotm = scaled_masked_softmax_cuda.backward(dm, ym, scale[0])
    assert gotm.data_ptr() == dm.data_ptr()
    _ulp_close(gotm, wantm, torch.bfloat16, ulps=4)
    # wave64 value of the reference's formula (scaled_masked_softmax.h:338-352)
    for sq, sk, b, np_ in ((2048, 2048, 4, 16), (64, 64, 4, 16), (32, 32, 4, 16), (128, 17, 1, 4), (4, 129, 2, 2)):
        assert scaled_masked_softmax_cuda.get_batch_per_block(sq, sk, b, np_) == O.softmax_batch_per_block(sq, sk, b, np_)
    # dtype / shape violations are RuntimeError, as AT_ASSERTM -> c10::Error is (scaled_masked_softmax.cpp:34-37)
    with pytest.raises(RuntimeError):
        scaled_upper_triang_masked_softmax_cuda.forward(x.float(), scale[0])
    with pytest.raises(RuntimeError):
        scaled_masked_softmax_cuda.forward(xm.float(), mask, scale[0])
    with pytest.raises(RuntimeError):
        scaled_upper_triang_masked_softmax_cuda.forward(x[:, :, :32], scale[0])        # not [attn_batches, s, s]
    with pytest.raises(RuntimeError):
        scaled_masked_softmax_cuda.forward(xm, mask[:, :, :8], scale[0])               # mask rows != sq
    with pytest.raises(RuntimeError):
        big = torch.zeros(1, 4, 4100, dtype=torch.half, device=cuda_device)
        scaled_masked_softmax_cuda.forward(big.view(1, 1, 4, 4100), torch.zeros(1, 1, 4, 4100, dtype=torch.bool, device=cuda_device), scale[0])


@pytest.mark.parametrize("name", EAGER)
def test_level2_class_three_fusion_types_on_reference_eager_fixtures(cuda_device, name):
    from multimodal_amd.model.fused_softmax import FusedScaleMaskSoftmax, SoftmaxFusionTypes
    t, meta = golden_io.load(name)
    x = t["x"]
    dt = x.dtype
    b, np_, sk = x.size(0), x.size(1), x.size(-1)
    scale, in32 = meta["scale"], meta["softmax_in_fp32"]

    def module(kind):
        return FusedScaleMaskSoftmax(input_in_fp16=dt == torch.float16, input_in_bf16=dt == torch.bfloat16, fusion_type=kind,
                                     mask_func=_mask_func, softmax_in_fp32=in32, scale=scale)

    causal = O.causal_mask(sk).to(cuda_device)
    eligible = O.fused_softmax_available(True, True, True, b, np_, sk, sk, True)
    # upper_triang: fused when eligible (the mask tensor is ignored), eager otherwise - either way the reference's bar
    m = module(SoftmaxFusionTypes.upper_triang)
    assert m.is_kernel_available(causal, b, np_, sk, sk) == eligible
    xin = x.to(cuda_device).clone().requires_grad_(True)
    p = m(xin * 1.0, causal)
    p.backward(t["dy"].to(cuda_device))
    assert (p.detach().float().cpu() - t["probs_causal"].float()).abs().mean() <= 1e-3
    assert (xin.grad.float().cpu() - t["dx_causal"].float()).abs().mean() <= 2e-3
    if eligible:
        s_eff = 1.0 if scale is None else scale
        _ulp_close(p.detach(), O.softmax_causal_fused_fwd(x.view(-1, sk, sk), s_eff).view_as(x), dt)
    # general: the explicit-mask kernel on the reference's random mask
    mg = module(SoftmaxFusionTypes.general)
    gm = t["mask_general"].to(cuda_device)
    assert mg.is_kernel_available(gm, b, np_, sk, sk) == O.fused_softmax_available(True, True, True, b, np_, sk, sk, False)
    pg = mg(x.to(cuda_device).clone(), gm)
    assert (pg.float().cpu() - t["probs_general"].float()).abs().mean() <= 1e-3
    # none: the eager sequence, bit for bit what the reference computed on CPU up to the device's exp
    mn = module(SoftmaxFusionTypes.none)
    assert not mn.is_kernel_available(gm, b, np_, sk, sk)
    pn = mn(x.to(cuda_device).clone(), gm)
    assert (pn.float().cpu() - t["probs_general"].float()).abs().mean() <= 1e-3
    # no mask -> never fused (fused_softmax.py:152)
    assert not m.is_kernel_available(None, b, np_, sk, sk)


def test_eligibility_table_matches_the_oracle(cuda_device):
    from multimodal_amd.model.fused_softmax import FusedScaleMaskSoftmax, SoftmaxFusionTypes
    mods = {k: FusedScaleMaskSoftmax(True, False, k, _mask_func, False, None) for k in SoftmaxFusionTypes}
    m32 = FusedScaleMaskSoftmax(False, False, SoftmaxFusionTypes.general, _mask_func, False, None)  # fp32 input: never fused
    mask = object()
    for b in (1, 2, 4):
        for np_ in (1, 2, 4, 16):
            for sq in (4, 6, 16, 64, 2048, 2049):
                for sk in (16, 17, 64, 128, 129, 2048, 2049):
                    for kind, mod in mods.items():
                        want = O.fused_softmax_available(kind != SoftmaxFusionTypes.none, True, True, b, np_, sq, sk,
                                                         kind == SoftmaxFusionTypes.upper_triang)
                        assert mod.is_kernel_available(mask, b, np_, sq, sk) == want, (kind, b, np_, sq, sk)
                    assert not m32.is_kernel_available(mask, b, np_, sq, sk)
    with pytest.raises(AssertionError):
        FusedScaleMaskSoftmax(True, False, SoftmaxFusionTypes.general, _mask_func, False, 0.5)  # scaled softmax must be fp32
    with pytest.raises(AssertionError):
        FusedScaleMaskSoftmax(True, False, "general", _mask_func, False, None)


def test_ineligible_shapes_run_on_the_hip_kernel_or_raise(cuda_device):
    """FusedScaleMaskSoftmax.forward_torch_softmax (the reference's eager branch, fused_softmax.py:184-199) computes on the HIP
    masked kernel for every 16-bit shape it can take - odd sizes, no mask, fusion type none - and raises for the rest:
    nothing in the package computes outside libmm_hip.so."""
    from multimodal_amd.model.fused_softmax import FusedScaleMaskSoftmax, SoftmaxFusionTypes
    torch.manual_seed(0)
    for dt in (torch.bfloat16, torch.float16):
        for (b, np_, sq, sk, scale, in32) in ((1, 3, 5, 19, None, False), (2, 1, 7, 130, 0.5, True), (1, 1, 2, 2048, None, False)):
            m = FusedScaleMaskSoftmax(dt == torch.float16, dt == torch.bfloat16, SoftmaxFusionTypes.none, _mask_func, in32, scale)
            x = (torch.randn(b, np_, sq, sk) * 3).to(dt)
            mask = torch.rand(b, 1, sq, sk) < 0.3
            for mk in (mask, None):
                xin = x.to(cuda_device).clone().requires_grad_(True)
                before = xin.detach().clone()
                p = m(xin, mk.to(cuda_device) if mk is not None else None)
                assert torch.equal(xin.detach(), before), "scores are not modified"
                ref = O.softmax_eager(x.clone(), mk, scale, in32)
                assert (p.detach().float().cpu() - ref.float()).abs().mean() <= 1e-3
                g = torch.randn(b, np_, sq, sk).to(dt)
                p.backward(g.to(cuda_device))
                x32 = x.float().requires_grad_(True)
                z = x32 * (1.0 if scale is None else scale)
                if mk is not None:
                    z = z.masked_fill(mk, -10000.0)
                torch.softmax(z, -1).backward(g.float())
                assert (xin.grad.float().cpu() - x32.grad).abs().mean() <= 2e-3
    m = FusedScaleMaskSoftmax(True, False, SoftmaxFusionTypes.none, _mask_func, False, None)
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 1, 4, 4100, device=cuda_device).half(), None)      # more keys than a row kernel holds
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 1, 1, 64, device=cuda_device).half(), None)        # single query row (kv-cache decode)
    m32 = FusedScaleMaskSoftmax(False, False, SoftmaxFusionTypes.none, _mask_func, False, None)
    with pytest.raises(RuntimeError):
        m32(torch.randn(1, 1, 4, 64, device=cuda_device), None)             # fp32 scores
