"""Build recipe for libspinnerf_hip.so (gfx950 only, hipcc cross-compiles without a GPU).

Used by ``__graft_entry__.build()`` and by ``python -m`` style manual builds:

    python spin-nerf_amd/build.py
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libspinnerf_hip.so")

# (source, extra flags).  render_ops is built without FMA contraction so its elementwise fp32
# arithmetic rounds like the reference's separate torch ops.
SOURCES = [
    ("mlp_fwd.hip", []),
    ("mlp_bwd.hip", []),
    ("render_ops.hip", ["-ffp-contract=off"]),
    ("prof.cpp", ["-x", "hip"]),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "spinnerf_hip.h"))
    objs = []
    procs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(LIB_DIR, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [_hipcc()] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or _stale(LIB_PATH, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
