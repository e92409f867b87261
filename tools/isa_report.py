#!/usr/bin/env python3
"""Static look at what hipcc made of a kernel (no GPU needed): registers, scratch, and the memory skeleton.

    python tools/isa_report.py sige_amd/csrc/conv_k3s1_nhwc.hip                      # resource table of every kernel
    python tools/isa_report.py sige_amd/csrc/conv_k3s1_nhwc.hip --kernel 'Li16EEELi1ELi1ELi1ELi0ELi1ELi4'
                                                                                      # + load / wait / barrier skeleton

The skeleton is the sequence of global / buffer / scalar loads, `s_waitcnt`s, barriers, MFMA runs and stores of one
kernel with their line numbers in the ISA.  Two patterns cost the fused conv ~1 us per launch each until they were
found this way (profiles/README.md, r1o / r1p):
  * a load immediately followed by `s_waitcnt vmcnt(0)`, repeated per loop iteration / per staging slot: dependent
    memory round trips in series (marked `<-- serial` below);
  * kernel arguments fetched lazily (`s_load` + `s_waitcnt lgkmcnt(0)` pairs spread over the prologue).
Scratch (`private_segment_fixed_size` > 0) means register arrays indexed dynamically or spilled: every scratch access
drains `vmcnt` to 0.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def compile_to_asm(src: str, workdir: str) -> str:
    from sige_amd import build

    cmd = [build._hipcc(), *build.FLAGS, "-I" + os.path.join(REPO, "include"), "-I" + build.CSRC, "-c", os.path.abspath(src),
           "-o", os.path.join(workdir, "out.o"), "-save-temps=obj"]
    subprocess.run(cmd, check=True, cwd=workdir, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(workdir):
        if f.endswith("gfx950.s"):
            return open(os.path.join(workdir, f)).read()
    raise RuntimeError("no gfx950 assembly produced")


def resources(asm: str):
    """[(name, {vgpr, agpr, sgpr, spill, scratch, kernarg})] from the code-object metadata."""
    out = []
    for block in asm.split("  - .agpr_count:")[1:]:
        block = ".agpr_count:" + block

        def field(key, cast=int):
            m = re.search(r"\.%s:\s+(\S+)" % key, block)
            return cast(m.group(1)) if m else None

        out.append((field("name", str), dict(vgpr=field("vgpr_count"), agpr=field("agpr_count"), sgpr=field("sgpr_count"),
                                               spill=field("vgpr_spill_count"), scratch=field("private_segment_fixed_size"),
                                               kernarg=field("kernarg_segment_size"))))
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.split("\n")[:len(names)]
    except Exception:
        return list(names)


INTERESTING = re.compile(r"\b(global_load|buffer_load|scratch_load|scratch_store|global_store|buffer_store|s_load|s_buffer_load|"
                         r"s_waitcnt|s_barrier|v_mfma|s_endpgm)\w*")


def skeleton(asm: str, mangled: str):
    m = re.search(r"^%s:.*?s_endpgm" % re.escape(mangled), asm, flags=re.S | re.M)
    if not m:
        raise SystemExit("kernel body not found: " + mangled)
    lines = m.group(0).split("\n")
    events = []
    for i, ln in enumerate(lines):
        mm = INTERESTING.search(ln)
        if mm and not ln.lstrip().startswith(";"):
            events.append((i, ln.strip().split(";")[0].strip()))
    out, run = [], None
    for k, (i, ins) in enumerate(events):
        op = ins.split()[0]
        if op.startswith("v_mfma"):
            if run is None:
                run = [i, i, 0]
            run[1], run[2] = i, run[2] + 1
            continue
        if run is not None:
            out.append("%6d-%-6d %d x v_mfma" % tuple(run))
            run = None
        note = ""
        if op == "s_waitcnt" and "vmcnt(0)" in ins and k > 0:
            prev = events[k - 1][1].split()[0]
            if prev.startswith(("global_load", "buffer_load", "scratch_load")) and (k < 2 or not events[k - 2][1].split()[0].startswith(("global_load", "buffer_load"))):
                note = "   <-- serial: one load, then wait for everything"
        out.append("%6d        %s%s" % (i, ins[:110], note))
    if run is not None:
        out.append("%6d-%-6d %d x v_mfma" % tuple(run))
    return len(lines), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source", help="a .hip translation unit of sige_amd/csrc")
    ap.add_argument("--kernel", help="regex on the MANGLED name: print the skeleton of the first match")
    ap.add_argument("--all", action="store_true", help="resource table: all kernels (default: only those with scratch / spills, plus the 10 largest)")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as d:
        asm = compile_to_asm(a.source, d)
    res = resources(asm)
    names = demangle([n for n, _ in res])
    rows = sorted(zip(names, res), key=lambda r: -(r[1][1]["vgpr"] or 0))
    print("%-5s %-5s %-5s %-6s %-8s %-8s kernel" % ("vgpr", "agpr", "sgpr", "spill", "scratch", "kernarg"))
    shown = 0
    for nm, (_, r) in rows:
        flagged = (r["scratch"] or 0) > 0 or (r["spill"] or 0) > 0
        if a.all or flagged or shown < 10:
            print("%-5s %-5s %-5s %-6s %-8s %-8s %s%s" % (r["vgpr"], r["agpr"], r["sgpr"], r["spill"], r["scratch"], r["kernarg"],
                                                           nm[:150], "   <-- scratch" if flagged else ""))
            shown += 1
    print("%d kernels, %d with scratch or spills" % (len(res), sum(1 for _, r in res if (r["scratch"] or 0) > 0 or (r["spill"] or 0) > 0)))
    if a.kernel:
        hit = [n for n, _ in res if re.search(a.kernel, n)]
        if not hit:
            raise SystemExit("no kernel matches " + a.kernel)
        n_lines, sk = skeleton(asm, hit[0])
        print("\n== %s\n   (%d ISA lines)" % (demangle([hit[0]])[0], n_lines))
        print("\n".join(sk))


if __name__ == "__main__":
    main()
