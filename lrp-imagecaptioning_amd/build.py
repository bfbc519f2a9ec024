"""Build liblrp_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblrp_hip.so")
SOURCES = ["engine.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "lrp_hip.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS
               if os.path.exists(os.path.join(CSRC, f)))


def build_library(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # build into a private file and rename: several ranks of one job may get here at once, and a reader must never see
    # a half-written library
    tmp = "%s.%d.tmp" % (LIB, os.getpid())
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-o", tmp] + os.environ.get("LRP_HIPCC_FLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[build]", " ".join(cmd).replace(tmp, LIB), file=sys.stderr)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
