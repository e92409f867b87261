"""One-question probe (round 4): which launches of the two-stream K-split stress test (tests/test_gpu_round4.py) differ from
their reference bits -- tile kernel or dense-layer kernel, eager or graph, one stream or two -- and by how much."""
import sys

import torch

sys.path.insert(0, ".")
import os as _os; _os.environ.setdefault("SIGE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "..", "sige_amd", "lib", "libsige_hip_tuning.so"))  # noqa: E702 -- dispatch knobs exist only in the measurement build (python -m sige_amd.build --tuning)
from sige_amd import hip  # noqa: E402
from tests.test_gpu_round2 import _pair_case  # noqa: E402

DEV = "cuda"
cl = lambda t: t.contiguous(memory_format=torch.channels_last)  # noqa: E731
res, c1, c2, cout = 8, 512, 512, 512
convs = [_pair_case(hip, res, c1, c2, cout, 0, True, seed=11 + i, residual=True)[1] for i in range(4)]
g = torch.Generator().manual_seed(5)
r = lambda *s: torch.randn(*s, generator=g).to(DEV)  # noqa: E731
xs = [cl(r(1, 1024, 16, 16)) for _ in range(4)]
w = r(512, 1024, 3, 3) / (3 * 32.0)
sc, sh, bias = r(1, 1024, 1, 1), r(1, 1024, 1, 1), r(512)
pw = hip.wide_conv_pack_weights(w, "f32")
wide = [lambda x=x: hip.wide_conv_cl(x, None, sc, sh, "swish", pw, bias, 512, (3, 3)) for x in xs]
hip.conv_force_ksplit(4)
hip.conv_force_ksplit_pass(True)
want = [c().clone() for c in convs]
hip.conv_force_ksplit_pass(False)
want_w = [f().clone() for f in wide]
again = [f().clone() for f in wide]
torch.cuda.synchronize()
print("wide deterministic:", [bool(torch.equal(a, b)) for a, b in zip(want_w, again)])
stats = {}


def check(tag, outs):
    for i, kind, o in outs:
        ref = (want_w if kind else want)[i]
        d = (o - ref).abs()
        nbad = int((d > 0).sum())
        if nbad:
            k = (tag, "wide" if kind else "tile")
            s = stats.setdefault(k, [0, 0, 0.0])
            s[0] += 1
            s[1] += nbad
            s[2] = max(s[2], float(d.max()))


def eager(tag, streams, rounds):
    for rnd in range(rounds):
        outs = []
        for k in range(100):
            for si, st in enumerate(streams):
                with torch.cuda.stream(st):
                    i = (k + 2 * si + rnd) % 4
                    outs.append((i, 0, convs[i]()))
                    outs.append((i, 1, wide[i]()))
        torch.cuda.synchronize()
        check(tag, outs)


def graphs(tag, streams, reps, kinds=(0, 1)):
    gs = []
    for si, st in enumerate(streams):
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            torch.cuda.synchronize()
            with torch.cuda.graph(gph, stream=st):
                outs = []
                for k in range(100):
                    i = (k + 2 * si) % 4
                    if 0 in kinds:
                        outs.append((i, 0, convs[i]()))
                    if 1 in kinds:
                        outs.append((i, 1, wide[i]()))
        gs.append((gph, outs))
    for rep in range(reps):
        for (gph, _), st in zip(gs, streams):
            with torch.cuda.stream(st):
                gph.replay()
        torch.cuda.synchronize()
        for _, outs in gs:
            check(tag, outs)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
eager("eager 1 stream", [s1], 5)
eager("eager 2 streams", [s1, s2], 5)
graphs("graph 1 stream", [s1], 20)
graphs("graph 2 streams", [s1, s2], 20)
graphs("graph 2 streams tile only", [s1, s2], 20, kinds=(0,))
graphs("graph 2 streams wide only", [s1, s2], 20, kinds=(1,))
print("mismatching launches (launches, elements, max |d|):")
for k, v in sorted(stats.items()):
    print("  ", k, v)
if not stats:
    print("   none")
