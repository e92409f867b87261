#!/usr/bin/env python3
"""Static scan (no GPU) of the kernels' ISA for the three prologue patterns of DESIGN.md 5.7:

  guarded   a global / buffer load within ten instructions behind `s_cbranch_execz` / `s_and_saveexec` -- `ok ? load : 0` compiled
            to an exec-masked branch around the load; with a use inside the branch every such load is a dependent round trip
  serial    `load; ... s_waitcnt vmcnt(0)` inside a loop body of <= 12 instructions (a rolled copy loop: one round trip per trip)
  div64     64-bit integer divisions by run-time values (the expansion's `v_mul_hi_u32` / `v_rcp_iflag_f32` pairs next to
            `v_addc` carries: counted as rcp_iflag instructions in kernels that also carry 64-bit adds around them)

    python tools/isa_guarded_loads.py [file.hip ...] > profiles/<tag>_isa_guarded_loads.txt      (default: every csrc/*.hip)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from sige_amd import build  # noqa: E402


def asm_of(src, tmp):
    out = os.path.join(tmp, os.path.basename(src) + ".s")
    cmd = [build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(REPO, "include"),
           "-I" + os.path.join(REPO, "sige_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def scan(lines):
    rows, cur, body = {}, None, []
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if t.startswith(".LBB"):
                body.append(t)
            if "-- End function" in t or t.startswith(".Lfunc_end"):
                rows[cur] = body
                cur = None
            continue
        body.append(t)
    res = {}
    for k, b in rows.items():
        loads = guarded = serial = rcp = 0
        for i, t in enumerate(b):
            if t.startswith(("global_load", "buffer_load")):
                loads += 1
                back = b[max(0, i - 10):i]
                if any(x.startswith(("s_cbranch_execz", "s_and_saveexec")) for x in back):
                    guarded += 1
                fwd = b[i + 1:i + 13]
                for j, x in enumerate(fwd):
                    if x.startswith(".LBB"):
                        break
                    if "vmcnt(0)" in x and any(y.startswith(("s_cbranch_execnz", "s_cbranch_scc", "s_cbranch_vcc")) for y in fwd[j:]):
                        serial += 1
                        break
            if t.startswith("v_rcp_iflag_f32"):
                rcp += 1
        if guarded or serial or rcp >= 3:
            res[k] = (loads, guarded, serial, rcp)
    return res


def main():
    srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(REPO, "sige_amd", "csrc", "*.hip")))
    print("%-26s %5s %7s %6s %5s  kernel" % ("file", "loads", "guarded", "serial", "rcp"))
    with tempfile.TemporaryDirectory() as tmp:
        for src in srcs:
            try:
                res = scan(asm_of(src, tmp))
            except subprocess.CalledProcessError:
                print("%-26s (does not compile standalone)" % os.path.basename(src))
                continue
            if not res:
                continue
            names = subprocess.run(["c++filt"] + list(res), capture_output=True, text=True).stdout.split("\n")
            for (k, v), n in zip(res.items(), names):
                print("%-26s %5d %7d %6d %5d  %s" % (os.path.basename(src), *v, n[:150]))


if __name__ == "__main__":
    main()
