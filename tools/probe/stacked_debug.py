"""One-question probe (round 4): where does a stacked forward first differ from the single-edit forwards?  Forward hooks on every
block of the DDPM workload record the outputs of E single-edit forwards and of one stacked forward; the first module whose
per-edit slice differs by more than 1e-4 is printed with the rows where it differs."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from sige_amd import stacked  # noqa: E402
from sige_amd.utils import dilate_mask, downsample_mask  # noqa: E402
from sige_amd.workloads.ddpm_unet import AttnBlock, DDPMConfig, DDPMSparseUNet, Downsample, ResBlock, Upsample  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
model = DDPMSparseUNet(DDPMConfig()).eval().to(dev).to(memory_format=torch.channels_last)
model.set_scatter_inplace(True)
x0, noise = bench.make_inputs()
cl = lambda a: a.to(dev).contiguous(memory_format=torch.channels_last)  # noqa: E731
x0, noise, t = cl(x0), cl(noise), torch.zeros(1, device=dev)
places = [(0.012, 100, 90), (0.02, 0, 40), (0.03, 256 - 44, 150)]
masks = [bench.square_mask(r, top=tp, left=lf).to(dev) for r, tp, lf in places]
E = len(masks)
build = lambda m: downsample_mask(dilate_mask(m, 5), 8)  # noqa: E731

names = {m: n for n, m in model.named_modules()}
rec = []


def hook(mod, inp, out):
    if isinstance(out, torch.Tensor):
        rec.append((names[mod], out.detach().clone()))


for m in model.modules():
    if isinstance(m, (ResBlock, AttnBlock, Upsample, Downsample)):
        m.register_forward_hook(hook)

with torch.no_grad():
    model.set_mode("full")
    model(x0, t)
    singles, single_out = [], []
    for mk in masks:
        model.set_masks(build(mk))
        model.set_mode("sparse")
        x1 = x0 + noise * mk
        model(x1, t)
        rec.clear()
        single_out.append(model(x1, t).clone())
        singles.append(list(rec))
    xe = cl(torch.cat([x0 + noise * mk for mk in masks], 0))
    stacked.stack_caches(model, E)
    model.set_masks(stacked.stack_masks([build(mk) for mk in masks]))
    with stacked.edit_batch(model, E):
        model(xe, t)
        model(xe, t)
        rec.clear()
        out = model(xe, t).clone()
        st = list(rec)
    print("modules recorded", len(st), "final max err per edit",
          [float((out[e] - single_out[e][0]).abs().max()) for e in range(E)])
    shown = 0
    for i, (name, ts) in enumerate(st):
        per = stacked.untall(ts, E) if ts.shape[0] == 1 else ts
        for e in range(E):
            ref = singles[e][i][1][0]
            d = (per[e] - ref).abs()
            err = float(d.max())
            if err > 1e-4 and shown < 12:
                rows = d.amax(dim=(0, 2)).nonzero().flatten()
                cols = d.amax(dim=(0, 1)).nonzero().flatten()
                print("%-28s edit %d err %.3e rows %s..%s (%d) cols %s..%s (%d) shape %s" % (
                    name, e, err, int(rows.min()), int(rows.max()), rows.numel(), int(cols.min()), int(cols.max()), cols.numel(),
                    tuple(per.shape)))
                shown += 1
