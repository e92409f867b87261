#!/usr/bin/env python3
"""Compile the REAL reference CPU backend (sige/cpu/*.cpp) into oracle/_ref/.

TEST INFRASTRUCTURE ONLY.  The sources are compiled from where they lie under
/root/reference (nothing is copied into this repo); the output is a pybind11 /
libtorch extension module ``oracle/_ref/sige_ref_cpu.so`` exporting the
reference's five functions (sige/cpu/pybind_cpu.cpp:5-12):

    gather, scatter, scatter_with_block_residual, scatter_gather, get_scatter_map

We do not run the reference's setup.py; this is a direct g++ invocation with the
flags setup.py:153-164 uses (-O3 -fopenmp) plus the libtorch include/link lines.
The three .cpp files #include "common_cpu.cpp" / "../common.cpp" themselves.

The .so only depends on libtorch (present on the GPU box's identical image), so
it travels with the gpurun snapshot and serves as the `"kind": "reference"` CPU
baseline and as a live cross-check of oracle/sige_oracle.c.
"""
import argparse
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
NAME = "sige_ref_cpu"


def build(ref_root: str = "/root/reference", verbose: bool = True) -> str:
    import torch
    from torch.utils import cpp_extension

    src_dir = os.path.join(ref_root, "sige", "cpu")
    sources = [os.path.join(src_dir, f) for f in ("gather.cpp", "scatter.cpp", "scatter_gather.cpp", "pybind_cpu.cpp")]
    for s in sources:
        if not os.path.isfile(s):
            raise FileNotFoundError(s)
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, NAME + ".so")
    newest_src = max(os.path.getmtime(s) for s in sources + [os.path.abspath(__file__)])
    if os.path.isfile(out) and os.path.getmtime(out) >= newest_src:
        return out

    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    incs = cpp_extension.include_paths() + [sysconfig.get_paths()["include"]]
    cmd = ["g++", "-O3", "-fopenmp", "-std=c++17", "-fPIC", "-shared", "-w",
           "-DTORCH_EXTENSION_NAME=" + NAME, "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for i in incs:
        cmd += ["-isystem", i]
    cmd += sources
    cmd += ["-L" + torch_lib, "-Wl,-rpath," + torch_lib,
            "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python", "-lgomp", "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def load():
    """Import the prebuilt oracle/_ref/sige_ref_cpu.so (None if absent)."""
    import importlib.util

    import torch  # noqa: F401  (libtorch symbols must be loaded first)

    path = os.path.join(HERE, "_ref", NAME + ".so")
    if not os.path.isfile(path):
        return None
    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    a = ap.parse_args()
    if not os.path.isdir(a.ref):
        print("reference tree %s not present: skipping oracle/_ref build" % a.ref)
        sys.exit(0)
    print(build(a.ref))
