"""Deterministic, name-keyed initialisation shared by the fixture generator (which applies it to the REFERENCE's model
classes in the build container) and the GPU tests (which apply it to sige_amd's workload models): both sides end up with
bit-identical weights because every tensor is drawn from a CPU generator seeded by a hash of its state-dict name."""
import zlib

import torch


def init_by_name(model: torch.nn.Module, seed: int = 0) -> None:
    with torch.no_grad():
        for name, t in list(model.named_parameters()) + list(model.named_buffers()):
            if name.endswith("num_batches_tracked"):
                continue
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
            if name.endswith("running_var"):
                v = torch.rand(t.shape, generator=g) + 0.5
            elif name.endswith("running_mean"):
                v = torch.randn(t.shape, generator=g) * 0.3
            elif t.dim() >= 2:
                v = torch.randn(t.shape, generator=g) / float(t[0].numel()) ** 0.5
            else:
                v = torch.randn(t.shape, generator=g) * 0.1
            t.copy_(v.to(t.dtype))


def gaugan_labels(H=256, W=512, nc=36, seed=3):
    """(original, edited) one-hot label maps [1,nc,H,W]: blocky random labels, a ~5 % rectangle relabelled."""
    import numpy as np

    rs = np.random.RandomState(seed)
    coarse = rs.randint(0, nc, size=(H // 8, W // 8))
    lab0 = np.kron(coarse, np.ones((8, 8), dtype=np.int64))
    lab1 = lab0.copy()
    h0, w0, hh, ww = H // 3, W // 4, int(H * 0.2), int(W * 0.25)
    lab1[h0:h0 + hh, w0:w0 + ww] = (lab0[h0:h0 + hh, w0:w0 + ww] + 5) % nc
    onehot = lambda l: torch.nn.functional.one_hot(torch.from_numpy(l), nc).permute(2, 0, 1)[None].float().contiguous()  # noqa: E731
    return onehot(lab0), onehot(lab1)


def summarize(t: torch.Tensor, step: int = 4, cstep: int = 1):
    """What a fixture keeps of a model output: every `step`-th pixel of every `cstep`-th channel, plus sums that depend on
    every value."""
    t = t.detach().float().cpu()
    return {"sub": t[:, ::cstep, ::step, ::step].contiguous().numpy(), "sum": float(t.double().sum()),
            "abs_sum": float(t.double().abs().sum()), "shape": list(t.shape)}


def sd_transformer_inputs(C=320, H=64, W=64, B=2, ctx_dim=768, seed=5):
    """(original, edited, context, mask512) of the SD spatial-transformer fixture: CFG batch 2, a 15 % square edit of the
    512 x 512 image seen at the 64 x 64 latent resolution."""
    import numpy as np

    rs = np.random.RandomState(seed)
    x0 = torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32))
    noise = torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32))
    ctx = torch.from_numpy(rs.standard_normal((B, 77, ctx_dim)).astype(np.float32))
    mask512 = torch.zeros(512, 512, dtype=torch.bool)
    mask512[150:348, 120:318] = True  # 198^2 / 512^2 = 15 %
    return x0, noise, ctx, mask512


def sd_unet_inputs(B=2, ctx_dim=768, seed=9):
    """(original latent, noise, context, timesteps, mask512) of the SD U-Net fixture: CFG batch 2, 64 x 64 latent, 15 % edit."""
    import numpy as np

    rs = np.random.RandomState(seed)
    x0 = torch.from_numpy(rs.standard_normal((B, 4, 64, 64)).astype(np.float32))
    noise = torch.from_numpy(rs.standard_normal((B, 4, 64, 64)).astype(np.float32))
    ctx = torch.from_numpy(rs.standard_normal((B, 77, ctx_dim)).astype(np.float32))
    ts = torch.full((B,), 500.0)
    mask512 = torch.zeros(512, 512, dtype=torch.bool)
    mask512[150:348, 120:318] = True
    return x0, noise, ctx, ts, mask512
