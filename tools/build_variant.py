#!/usr/bin/env python
"""Build an experimental variant of the library: tools/build_variant.py NAME [-DFLAG=V ...] [--only mlp_fwd,mlp_bwd]
-> spin-nerf_amd/lib/ablate/libspinnerf_hip_NAME.so (objects of sources not listed in --only are reused from lib/).
Select it at run time with SNR_LIB=<path>."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
B = importlib.import_module("spin-nerf_amd.build")

name = sys.argv[1]
flags = [a for a in sys.argv[2:] if a.startswith("-D")]
only = None
for a in sys.argv[2:]:
    if a.startswith("--only="):
        only = a.split("=", 1)[1].split(",")
out_dir = os.path.join(B.LIB_DIR, "ablate")
os.makedirs(out_dir, exist_ok=True)
objs, procs = [], []
for src, extra in B.SOURCES:
    stem = os.path.splitext(src)[0]
    if only is not None and stem not in only:
        objs.append(os.path.join(B.LIB_DIR, stem + ".o"))
        continue
    o = os.path.join(out_dir, f"{stem}_{name}.o")
    objs.append(o)
    extra = [e for e in extra if not e.startswith("-save-temps")]
    cmd = [B._hipcc()] + B.COMMON + extra + flags + ["-c", os.path.join(B.CSRC, src), "-o", o]
    procs.append((cmd, subprocess.Popen(cmd)))
for cmd, p in procs:
    if p.wait() != 0:
        raise SystemExit("hipcc failed: " + " ".join(cmd))
lib = os.path.join(out_dir, f"libspinnerf_hip_{name}.so")
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
