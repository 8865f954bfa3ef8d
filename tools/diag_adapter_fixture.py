"""Developer diagnostic: error statistics of the HIP adapter-wrapped layer against the reference fixture and the fp32 oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import golden_io, golden_params, layer_harness as LH
from oracle import neox_layer_oracle as O
from multimodal_amd import mpu
from multimodal_amd.model import add_adapters
mpu.initialize_model_parallel(1)
dev = torch.device("cuda:0")
for name in sys.argv[1:]:
    t, meta = golden_io.load(name)
    dtype = torch.bfloat16
    rng = np.random.default_rng(meta["seed"])
    master = golden_params.master_params(rng, meta["h"])
    golden_params.rng_t(rng, (meta["s"], meta["b"], meta["h"])); golden_params.rng_t(rng, (meta["s"], meta["b"], meta["h"]))
    ad = golden_params.adapter_params(rng, meta["h"], meta["h"] // 8)
    for fusion in (True, False):
        for streams in ("1", "0"):
            os.environ["MM_BRANCH_STREAMS"] = streams
            args = LH.make_neox_args(meta["h"], meta["n"], dtype, meta["rotary_pct"], meta["gpt_j_residual"], meta["bias_gelu_fusion"], softmax_fusion=fusion)
            layer = LH.build_hip_layer(args, master, dev)
            holder = torch.nn.ModuleList([layer])
            add_adapters(args, holder, downsample_factor=8, location="mlp"); add_adapters(args, holder, downsample_factor=8, location="attention")
            sd = dict(layer.named_parameters())
            with torch.no_grad():
                for k in golden_params.ADAPTER_KEYS: sd[k].copy_(ad[k].to(dtype))
            for nme, p in layer.named_parameters(): p.requires_grad = "adapter" in nme
            y, dx, grads = LH.run_hip_layer(layer, t["x"], t["dy"], dev)
            cfg = O.LayerConfig(meta["h"], meta["n"], rotary_pct=meta["rotary_pct"], gpt_j_residual=meta["gpt_j_residual"], bias_gelu_fusion=meta["bias_gelu_fusion"])
            y32, dx32, shards = LH.oracle_layer_fp32(cfg, master, t["x"], t["dy"], dtype, adapters=ad)
            for what, got, r16, r32 in (("y", y, t["y"], y32), ("dx", dx, t["dx"], dx32)):
                g, r = got.float(), r16.float()
                eh, er = (g - r32).abs(), (r - r32).abs()
                print(name[-40:], "fusion", fusion, "streams", streams, what, "max hip %.4f ref %.4f | l2 hip %.5f ref %.5f | scale %.2f | p99.9 hip %.4f ref %.4f | argmax hip %s val32 %.3f"
                      % (eh.max(), er.max(), (g - r32).norm() / r32.norm(), (r - r32).norm() / r32.norm(), r32.abs().max(),
                         eh.flatten().kthvalue(int(eh.numel() * 0.999)).values, er.flatten().kthvalue(int(er.numel() * 0.999)).values,
                         tuple(np.unravel_index(int(eh.argmax()), eh.shape)), r32.flatten()[int(eh.argmax())]))
