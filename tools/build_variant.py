#!/usr/bin/env python3
"""An A/B build of the whole library with extra -D flags: lib/libsige_hip_<name>.so (select it with SIGE_HIP_LIB).
   python tools/build_variant.py plain -DSIGE_PLAIN_STORES"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sige_amd import build  # noqa: E402

name, defs = sys.argv[1], sys.argv[2:]
vdir = os.path.join(build.LIB_DIR, "variant_" + name)
os.makedirs(vdir, exist_ok=True)
objs, procs = [], []
for src in build.SOURCES:
    obj = os.path.join(vdir, src.replace(".hip", ".o"))
    objs.append(obj)
    cmd = build._compile_cmd(src, obj, ["-DSIGE_HIP_TUNING", *defs] if src in build.TUNING_UNITS else list(defs))
    while sum(1 for _, q in procs if q.poll() is None) >= 8:
        time.sleep(0.2)
    procs.append((cmd, subprocess.Popen(cmd)))
build._run_all(procs)
out = os.path.join(build.LIB_DIR, "libsige_hip_%s.so" % name)
subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
print(out)
