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
           "conv_pair_nhwc_t4.hip", "conv_pair_nhwc_t8.hip", "conv_pair_nhwc_f4.hip", "conv_pair_nhwc_f8.hip", "conv_pair_nhwc_h_t4.hip", "conv_pair_nhwc_h_f4.hip", "conv_wide.hip", "conv_wide_k3_p8.hip", "conv_wide_k3_f32.hip", "conv_wide_k1_p8.hip", "conv_wide_pair_f32.hip", "conv_wide_pair_x3.hip", "conv_wide_pair_f16.hip", "group_norm.hip", "attention.hip", "attention_fused.hip", "attention_tokens.hip", "nhwc_ops.hip", "conv_out.hip", "conv_in.hip", "spade_ops.hip", "conv_tile3.hip", "conv_tile3_f32.hip", "token_ops.hip", "conv_tile3_f16.hip"]
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


# translation units that read a dispatch knob (csrc/tuning.hpp): compiled a second time with -DSIGE_HIP_TUNING for the
# measurement library lib/libsige_hip_tuning.so; every other object is shared with the product build
TUNING_UNITS = ("api.hip", "block_conv.hip", "gather.hip", "conv_wide.hip", "attention_tokens.hip", "conv_out.hip", "conv_tile3.hip")
TUNING_LIB = os.path.join(LIB_DIR, "libsige_hip_tuning.so")


def _compile_cmd(src: str, obj: str, defs=()):
    # -DSIGE_TU_ID: the unit's anchor kernel (csrc/common.hpp) that sige_hip_preload() loads the unit's code object through
    return [_hipcc(), *FLAGS, "-DSIGE_TU_ID=%d" % SOURCES.index(src), *defs, "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-c",
            os.path.join(CSRC, src), "-o", obj]


def _headers():
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(REPO, "include", "sige_hip.h"))
    return headers


def _headers_of(src: str, headers):
    return (headers if src.startswith("conv_wide") else [h for h in headers if not h.endswith("conv_wide.hpp")] if src.startswith(("conv_k", "conv_pair", "conv_tile3", "block_conv"))
            else [h for h in headers if not h.endswith(("conv_mfma.hpp", "conv_wide.hpp", "conv_tile3.hpp"))])


def _run_all(procs):
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))


def build_tuning(force: bool = False, verbose: bool = True) -> str:
    """lib/libsige_hip_tuning.so: the product's objects, with the units of TUNING_UNITS rebuilt under -DSIGE_HIP_TUNING -- the only
    library that exports sige_hip_tuning_set / _get (include/sige_hip.h).  tools/, the bench sections that compare kernel
    forms and the tests that force a form load it (sige_amd.hip.tuning_build())."""
    build(force=force, verbose=verbose)
    tdir = os.path.join(LIB_DIR, "tuning")
    os.makedirs(tdir, exist_ok=True)
    headers = _headers()
    objs, procs = [], []
    for src in SOURCES:
        if src in TUNING_UNITS:
            obj = os.path.join(tdir, src.replace(".hip", ".o"))
            if force or _stale(obj, os.path.join(CSRC, src), _headers_of(src, headers)):
                cmd = _compile_cmd(src, obj, ["-DSIGE_HIP_TUNING"])
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((cmd, subprocess.Popen(cmd)))
        else:
            obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
    _run_all(procs)
    if procs or not os.path.isfile(TUNING_LIB) or os.path.getmtime(TUNING_LIB) < os.path.getmtime(LIB):
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", TUNING_LIB])
    return TUNING_LIB


def build_probe(verbose: bool = True) -> str:
    """The measurement build of tools/conv_phase_probe.py: the conv translation units with -DSIGE_CONV_PROBE (phase
    timestamps inside the kernel) linked with the tuning build's other objects into lib/libsige_hip_probe.so."""
    build_tuning(verbose=verbose)
    tag = os.environ.get("SIGE_PROBE_TAG", "")  # several measurement builds side by side (ablations): lib/libsige_hip_probe<tag>.so
    only = os.environ.get("SIGE_PROBE_ONLY", "").split()  # restrict the -D build to these translation units
    pdir = os.path.join(LIB_DIR, "probe" + tag)
    os.makedirs(pdir, exist_ok=True)
    objs, procs = [], []
    wide = bool(os.environ.get("SIGE_PROBE_WIDE"))  # the dense-layer conv's stamps instead (tools/probe/wide_phase_probe.py)
    for src in SOURCES:
        if (src.startswith("conv_wide") if wide else
                src.startswith(("conv_k", "conv_pair", "conv_tile3", "block_conv")) and (not only or src in only or src.startswith("block_conv"))):
            obj = os.path.join(pdir, src.replace(".hip", ".o"))
            # (SIGE_VARIANT_ONLY=1: an experimental variant of the product kernels -- the defines only, no phase stamps)
            probe_def = [] if os.environ.get("SIGE_VARIANT_ONLY") else ["-DSIGE_WIDE_PROBE" if wide else "-DSIGE_CONV_PROBE"]
            cmd = _compile_cmd(src, obj, ["-DSIGE_HIP_TUNING", *probe_def, *os.environ.get("SIGE_PROBE_DEFS", "").split()])
            procs.append((cmd, subprocess.Popen(cmd)))
        elif src in TUNING_UNITS:
            obj = os.path.join(LIB_DIR, "tuning", src.replace(".hip", ".o"))
        else:
            obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
    _run_all(procs)
    out = os.path.join(LIB_DIR, "libsige_hip_probe%s.so" % tag)
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = _headers()
    objs = []
    procs = []
    jobs = max(1, int(os.environ.get("SIGE_BUILD_JOBS", str(os.cpu_count() or 8))))
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        # only translation units whose source (or any header) changed are recompiled
        if not force and not _stale(obj, os.path.join(CSRC, src), _headers_of(src, headers)):
            continue
        cmd = _compile_cmd(src, obj)
        if verbose:
            print(" ".join(cmd), flush=True)
        while sum(1 for _, q in procs if q.poll() is None) >= jobs:
            import time

            time.sleep(0.2)
        procs.append((cmd, subprocess.Popen(cmd)))
    _run_all(procs)
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    if "--probe" in sys.argv:
        print(build_probe())
    elif "--tuning" in sys.argv:
        print(build_tuning(force="--force" in sys.argv))
    else:
        print(build(force="--force" in sys.argv))
