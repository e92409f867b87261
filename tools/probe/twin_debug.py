"""Debug aid: the flow of tests/test_gpu_channels_last.py::test_ddpm_unet_channels_last_equals_nchw with per-module outputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sige_amd.utils import dilate_mask, downsample_mask
from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet, ResBlock, AttnBlock, Upsample, Downsample

DEV = "cuda"
for nchw_first in (True, False):
    torch.manual_seed(0)
    model = DDPMSparseUNet(DDPMConfig(ch=32)).to(DEV).eval()
    x0 = torch.randn(1, 3, 256, 256, device=DEV)
    mask = torch.zeros(256, 256, dtype=torch.bool, device=DEV)
    mask[100:140, 90:150] = True
    edits = [x0 + torch.randn(1, 3, 256, 256, device=DEV) * mask for _ in range(2)]
    t = torch.zeros(1, device=DEV)
    masks = downsample_mask(dilate_mask(mask, 5), 8)
    mods = [(n, b) for n, b in model.named_modules() if isinstance(b, (ResBlock, AttnBlock, Upsample, Downsample))]
    with torch.no_grad():
        for layout in (("nchw", "nhwc") if nchw_first else ("nhwc",)):
            fmt = torch.channels_last if layout == "nhwc" else torch.contiguous_format
            model.to(memory_format=fmt)
            model.clear_cache()
            model.set_scatter_inplace(layout == "nhwc")
            model.set_mode("full")
            model(x0.contiguous(memory_format=fmt), t)
            model.set_masks(masks)
            model.set_mode("sparse")
            recs = []
            for i, e in enumerate((edits[1], edits[0], edits[1], edits[0])):
                feats = {}
                hooks = [b.register_forward_hook(lambda m, a, o, n=n: feats.__setitem__(n, o.detach().contiguous().clone())) for n, b in mods]
                out = model(e.contiguous(memory_format=fmt), t).contiguous().clone()
                for h in hooks: h.remove()
                recs.append((out, feats))
            d = float((recs[3][0] - recs[1][0]).abs().max())
            first = next(((n, float((recs[3][1][n] - recs[1][1][n]).abs().max())) for n, _ in mods if float((recs[3][1][n] - recs[1][1][n]).abs().max()) > 0), None)
            links = sum(len(b._twin_links) for _, b in mods if isinstance(b, ResBlock))
            print("nchw_first=%s layout=%s: forward 4 vs forward 2 (same input): %.3e, first module that differs %s, links %d" % (nchw_first, layout, d, first, links), flush=True)
