"""Round 4, on the MI355X: launch plans (a sparse forward that survives a mask change: sige_amd/plan.py, csrc/plan.hpp),
the dense-layer conv paired with a residual block's shortcut, fp16 cache storage, stress tests of the in-launch K-split finish
and of the device guard."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from sige_amd import hip as h

    h.lib()
    return h


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.fixture(scope="module")
def ddpm_pair():
    """Two DDPM-256 U-Nets (BASELINE.json configs[1] size) with the same weights and the same full-pass caches: one is driven
    by a launch plan, the other by the module-level (Python) path -- the reference of every plan test."""
    import bench
    from sige_amd.workloads.ddpm_unet import DDPMConfig, DDPMSparseUNet

    torch.manual_seed(0)
    a = DDPMSparseUNet(DDPMConfig()).eval()
    b = DDPMSparseUNet(DDPMConfig()).eval()
    b.load_state_dict(a.state_dict())
    x0, noise = bench.make_inputs()
    x0, noise = _cl(x0.to(DEV)), _cl(noise.to(DEV))
    t = torch.zeros(1, device=DEV)
    models = []
    with torch.no_grad():
        for m in (a, b):
            m = m.to(DEV).to(memory_format=torch.channels_last)
            m.set_scatter_inplace(True)
            m.set_mode("full")
            m(x0, t)
            models.append(m)
    return models[0], models[1], x0, noise, t


def _build_masks(mask):
    from sige_amd.utils import dilate_mask, downsample_mask

    return downsample_mask(dilate_mask(mask, 5), 8)


def _mask(ratio, top, left):
    import bench

    return bench.square_mask(ratio, top=top, left=left).to(DEV)


def test_launch_plan_follows_mask_changes(hip, ddpm_pair):
    """VERDICT r3 missing #1: ONE recording, then masks of different sizes and places in turn -- the forward issued from C
    (plan.run), the same calls replayed from a hipGraph (plan.replay) and the module-level forward of an identical second model
    must agree BIT FOR BIT, for every mask, including a return to the first one.  The reference sizes each launch from
    activeIndices.size(0) at call time (sige/cuda/gather_kernel.cu:78-84,111; sige/nn/gather.py:101-107)."""
    from sige_amd.plan import FORWARD, MASKS, LaunchPlan

    model, ref, x0, noise, t = ddpm_pair
    masks = [_mask(0.012, 100, 90), _mask(0.05, 60, 40), _mask(0.02, 150, 120), _mask(0.002, 8, 200), _mask(0.012, 100, 90),
             _mask(0.15, 30, 30)]
    xs = (x0 + noise * masks[0]).clone()
    with torch.no_grad():
        plan = LaunchPlan(model)
        out = plan.record(masks[0], _build_masks, lambda: model(xs, t))
        assert not plan.shape_bound
        assert plan.calls(MASKS) > 10 and 60 < plan.calls(FORWARD) < 200
        launches_per_run = None
        seen = set()
        for k, m in enumerate(masks):
            x1 = x0 + noise * m
            xs.copy_(x1)
            plan.bind_mask(m)
            n0 = hip.launch_count()
            got = plan.run().clone()
            launches_per_run = hip.launch_count() - n0
            rep = plan.replay().clone()
            # the module-level path on the second model (twins warm after its first forward under any mask)
            ref.set_masks(_build_masks(m))
            ref.set_mode("sparse")
            if k == 0:
                ref(x1, t)
            want = ref(x1, t)
            assert torch.equal(got, want), (k, float((got - want).abs().max()))
            assert torch.equal(rep, got), k
            # the plan's model itself, module-level, after adopting the plan's index lists
            assert torch.equal(model(xs, t), want), k
            seen.add(tuple(plan.counts))
        assert len(seen) >= 5  # (the masks really had different tile counts)
        assert launches_per_run is not None and launches_per_run <= 110
        # an empty edit: every count 0, the output is the forward of the cached original outside... nothing to compare tile by
        # tile, but it must run and equal the module-level result under the same (empty) mask
        empty = torch.zeros_like(masks[0])
        xs.copy_(x0)
        plan.bind_mask(empty)
        assert sum(plan.counts) == 0
        got = plan.run().clone()
        ref.set_masks(_build_masks(empty))
        assert torch.equal(got, ref(x0, t))
    del plan


def test_launch_plan_refuses_what_it_cannot_follow(hip):
    """A plan that recorded an NCHW (two-kernel) call is shape bound: it replays the recorded mask and refuses another."""
    from sige_amd import nn as snn
    from sige_amd.plan import LaunchPlan
    from sige_amd.nn.base import SIGEModel

    class Net(SIGEModel):
        def __init__(self):
            super().__init__()
            self.conv = snn.SIGEConv2d(16, 16, 3, padding=1)
            self.gather = snn.Gather(self.conv, 6)
            self.scatter = snn.Scatter(self.gather)

        def forward(self, x):
            return self.scatter(self.conv(self.gather(x)))

    torch.manual_seed(0)
    net = Net().to(DEV).eval()
    x = torch.randn(1, 16, 32, 32, device=DEV)  # NCHW: the reference's layout
    m0 = torch.zeros(32, 32, dtype=torch.bool, device=DEV)
    m0[4:12, 6:20] = True
    m1 = torch.zeros_like(m0)
    m1[20:24, 3:7] = True
    with torch.no_grad():
        net.set_mode("full")
        net(x)
        plan = LaunchPlan(net)
        xs = x.clone()
        xs[:, :, 4:12, 6:20] += 1.0
        out = plan.record(m0, lambda mk: {(32, 32): mk}, lambda: net(xs)).clone()
        assert plan.shape_bound
        plan.bind_mask(m0)  # (the recorded mask again: allowed)
        assert torch.equal(plan.run(), out)
        with pytest.raises(RuntimeError, match="cannot follow a new mask"):
            plan.bind_mask(m1)
