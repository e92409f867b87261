"""bench.py section: one sparse-query spatial transformer at the Stable Diffusion v1 level-1 shape (BASELINE.json configs[3])."""
import torch

from .common import _replay_ms


def sd_transformer_section(dev):
    """BASELINE.json configs[3], the part that is specific to Stable Diffusion: one sparse-query spatial transformer at the SD
    v1 level-1 shape (320 channels, 8 heads, text context 768), 64 x 64 latent, CFG batch 2, 15 % edit."""
    from sige_amd.nn import SIGEModel
    from sige_amd.utils import downsample_mask
    from sige_amd.workloads.sd_transformer import SpatialTransformer

    class Wrap(SIGEModel):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x, **kw):
            return self.m(x, **kw)

    res = {}
    cl = lambda t_: t_.contiguous(memory_format=torch.channels_last)  # noqa: E731
    gen = torch.Generator().manual_seed(5)
    x0 = cl(torch.randn(2, 320, 64, 64, generator=gen).to(dev))
    noise = cl(torch.randn(2, 320, 64, 64, generator=gen).to(dev))
    ctx = torch.randn(2, 77, 768, generator=gen).to(dev)
    mask512 = torch.zeros(512, 512, dtype=torch.bool, device=dev)
    mask512[150:348, 120:318] = True
    masks = downsample_mask(mask512, min_res=8, dilation=1)
    x1 = cl(x0 + noise * masks[(64, 64)])
    outs = {}
    with torch.no_grad():
        for name, kv in (("sparse_queries_kv_scattered", True), ("sparse_queries_kv_reprojected", False)):
            torch.manual_seed(0)
            model = Wrap(SpatialTransformer(320, 8, 40, depth=1, context_dim=768, block_size=4, sparse_kv=kv)).to(dev).eval()
            for p_ in model.parameters():
                if p_.dim() >= 2:
                    p_.data.normal_(0, 1.0 / float(p_[0].numel()) ** 0.5)
            model = model.to(memory_format=torch.channels_last)
            model.set_scatter_inplace(True)
            model.set_mode("full")
            if "dense_forward_ms" not in res:
                res["dense_forward_ms"] = round(_replay_ms(lambda: model(x1, context=ctx))[0], 3)
            model(x0, context=ctx)
            model.set_masks(masks)
            model.set_mode("sparse")
            ms, out, g = _replay_ms(lambda: model(x1, context=ctx))
            outs[name] = out.float().cpu()
            res[name] = {"forward_ms": round(ms, 3), "speedup_vs_dense": round(res["dense_forward_ms"] / ms, 2)}
            del g, model
    res["kv_scattered_vs_reprojected_max_abs"] = round(float((outs["sparse_queries_kv_scattered"] - outs["sparse_queries_kv_reprojected"]).abs().max()), 8)
    res["active_token_ratio"] = round(float(masks[(64, 64)].float().mean()), 4)
    res["workload"] = "SD v1 spatial transformer (320 ch, 8 heads x 40, context 768), latent [2,320,64,64] (CFG batch 2), fp32 NHWC, hipGraph replay"
    return res
