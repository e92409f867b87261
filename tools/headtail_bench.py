#!/usr/bin/env python3
"""The dense head / tail of the sparse forward (conv_in, norm_out, conv_out) at the DDPM-256 shapes:
us per launch from a hipGraph of back-to-back launches (HIP events on the launch stream)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os; _os.environ.setdefault("SIGE_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "sige_amd", "lib", "libsige_hip_tuning.so"))  # noqa: E702 -- dispatch knobs exist only in the measurement build (python -m sige_amd.build --tuning)
from sige_amd import hip
from tools.conv_bench import graph_time

dev = torch.device("cuda")
torch.manual_seed(0)
CL = torch.channels_last
xs = [torch.randn(1, 128, 256, 256, device=dev).contiguous(memory_format=CL) for _ in range(4)]
w = torch.randn(3, 128, 3, 3, device=dev) / 34
b = torch.randn(3, device=dev)
sc, sh = torch.randn(1, 128, 1, 1, device=dev), torch.randn(1, 128, 1, 1, device=dev)
gamma, beta = torch.randn(128, device=dev), torch.randn(128, device=dev)
imgs = [torch.randn(1, 3, 256, 256, device=dev).contiguous(memory_format=CL) for _ in range(4)]
win = torch.randn(128, 3, 3, 3, device=dev) / 5
bin_ = torch.randn(128, device=dev)

def rec(name, us, mb):
    print(json.dumps(dict(kernel=name, us=round(us, 2), alg_MB=mb, GBps=round(mb * 1e3 / us, 1))), flush=True)

rec("conv_out tap-GEMM (128->3, affine+swish)", graph_time(lambda i: hip.conv3x3_small_cout_cl(xs[i], w, b, sc, sh, "swish"), 4), 34.3)
hip.conv3x3_small_cout_force_scalar(True)
rec("conv_out scalar-weight kernel", graph_time(lambda i: hip.conv3x3_small_cout_cl(xs[i], w, b, sc, sh, "swish"), 4), 34.3)
hip.conv3x3_small_cout_force_scalar(False)
rec("conv_in thin GEMM (3->128)", graph_time(lambda i: hip.conv3x3_small_cin_cl(imgs[i], win, bin_), 4), 34.3)
conv = torch.nn.Conv2d(3, 128, 3, 1, 1).to(dev).to(memory_format=CL)
with torch.no_grad():
    rec("conv_in torch/MIOpen (+ layout fix-up)", graph_time(lambda i: conv(imgs[i]).contiguous(memory_format=CL), 4), 34.3)
rec("norm_out GroupNorm affine [1,128,256,256]", graph_time(lambda i: hip.group_norm_affine_cl(xs[i], 32, 1e-6, gamma, beta), 4), 33.5)
