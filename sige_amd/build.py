"""Build libsige_hip.so (gfx950) in-tree with hipcc.

`python -m sige_amd.build` or `sige_amd.build.build()`; __graft_entry__.build()
calls this.  hipcc cross-compiles without a GPU.  The .so is git-ignored but
travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libsige_hip.so")
SOURCES = ["api.hip", "plan.hip", "gather.hip", "scatter.hip", "reduce_mask.hip", "mask_pipeline.hip", "block_conv.hip", "conv_k3s1.hip", "conv_k1.hip",
           "conv_k3s2.hip", "conv_k3s1_nhwc.hip", "conv_k1_nhwc.hip", "conv_k3s2_nhwc.hip", "conv_k3s1_nhwc_w8.hip", "conv_k1_nhwc_w8.hip", "conv_k3s1_nhwc_h.hip", "conv_k1_nhwc_h.hip", "conv_k3s1_nhwc_x.hip", "conv_k1_nhwc_x.hip", "conv_k3s1_nhwc_c16.hip", "conv_k3s1_nhwc_h_c16.hip", "conv_k3s1_nhwc_x_c16.hip", "conv_pair_nhwc_x_t4.hip", "conv_pair_nhwc_x_f4.hip",
           "conv_pair_nhwc_t4.hip", "conv_pair_nhwc_t8.hip", "conv_pair_nhwc_f4.hip", "conv_pair_nhwc_f8.hip", "conv_pair_nhwc_h_t4.hip", "conv_pair_nhwc_h_f4.hip", "conv_wide.hip", "conv_wide_k3_p8.hip", "conv_wide_k3_f32.hip", "conv_wide_k1_p8.hip", "conv_wide_pair_f32.hip", "conv_wide_pair_x3.hip", "conv_wide_pair_f16.hip", "group_norm.hip", "attention.hip", "attention_fused.hip", "attention_tokens.hip", "nhwc_ops.hip", "conv_out.hip", "conv_in.hip"]
# -ffp-contract=off: the reference applies scale then shift as two separately
# rounded fp32 ops (sige/cpu/gather.cpp:33-53); an fma would differ in the last bit.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]


def source_hash() -> str:
    """sha256 over the kernel sources (csrc/*, include/sige_hip.h) and the compile flags: identifies the build a
    committed profile was taken on (bench.py prints `roofline.traffic` only for a matching hash; the GPU box has no .git)."""
    import hashlib

    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(REPO, "include", "sige_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.isfile(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build() -> bool:
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO, "include", "sige_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _stale(obj: str, src: str, headers) -> bool:
    if not os.path.isfile(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src, *headers])


def build_probe(verbose: bool = True) -> str:
    """The measurement build of tools/conv_phase_probe.py: the conv translation units with -DSIGE_CONV_PROBE (phase
    timestamps inside the kernel) linked with the product's other objects into lib/libsige_hip_probe.so."""
    build(verbose=verbose)
    tag = os.environ.get("SIGE_PROBE_TAG", "")  # several measurement builds side by side (ablations): lib/libsige_hip_probe<tag>.so
    only = os.environ.get("SIGE_PROBE_ONLY", "").split()  # restrict the -D build to these translation units
    pdir = os.path.join(LIB_DIR, "probe" + tag)
    os.makedirs(pdir, exist_ok=True)
    objs, procs = [], []
    wide = bool(os.environ.get("SIGE_PROBE_WIDE"))  # the dense-layer conv's stamps instead (tools/probe/wide_phase_probe.py)
    for src in SOURCES:
        if (src.startswith("conv_wide") if wide else
                src.startswith(("conv_k", "conv_pair", "block_conv")) and (not only or src in only or src.startswith("block_conv"))):
            obj = os.path.join(pdir, src.replace(".hip", ".o"))
            # (SIGE_VARIANT_ONLY=1: an experimental variant of the product kernels -- the defines only, no phase stamps)
            probe_def = [] if os.environ.get("SIGE_VARIANT_ONLY") else ["-DSIGE_WIDE_PROBE" if wide else "-DSIGE_CONV_PROBE"]
            cmd = [_hipcc(), *FLAGS, *probe_def, *os.environ.get("SIGE_PROBE_DEFS", "").split(),
                   "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-c",
                   os.path.join(CSRC, src), "-o", obj]
            procs.append((cmd, subprocess.Popen(cmd)))
        else:
            obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    out = os.path.join(LIB_DIR, "libsige_hip_probe%s.so" % tag)
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(REPO, "include", "sige_hip.h"))
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        # only translation units whose source (or any header) changed are recompiled
        hdrs = (headers if src.startswith("conv_wide") else [h for h in headers if not h.endswith("conv_wide.hpp")] if src.startswith(("conv_k", "conv_pair", "block_conv"))
                else [h for h in headers if not h.endswith(("conv_mfma.hpp", "conv_wide.hpp"))])
        if not force and not _stale(obj, os.path.join(CSRC, src), hdrs):
            continue
        cmd = [_hipcc(), *FLAGS, "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-c",
               os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    print(build_probe() if "--probe" in sys.argv else build(force="--force" in sys.argv))
