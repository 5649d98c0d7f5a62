"""Build libddz_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the
library is a plain C-ABI .so (include/ddz_env.h) that the host mirror binds with ctypes."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libddz_hip.so")
SOURCES = ["ddz_engine.hip"]
DEPS = ["ddz_device.h", "ddz_build_table.h", os.path.join("..", "..", "include", "ddz_env.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build libddz_hip.so")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + DEPS)


def build(force=False, verbose=False):
    if not (force or stale()):
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall",
           "-Wno-unused-function", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
