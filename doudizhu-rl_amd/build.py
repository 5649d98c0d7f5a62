"""Build libddz_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the
library is a plain C-ABI .so (include/ddz_env.h) that the host mirror binds with ctypes.
A second build of the same source, libddz_hip_jk.so (-DDDZ_NATIVE_JOKER_KICKERS=1), is the optional
rule-set extension with the 24 joker-kicker rows (default off; include/ddz_env.h)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libddz_hip.so")
LIB_JK = os.path.join(CSRC, "libddz_hip_jk.so")
SOURCES = ["ddz_engine.hip"]
DEPS = ["ddz_device.h", "ddz_build_table.h", "ddz_auto.h", "ddz_auto2.h", "ddz_qnet.h", os.path.join("..", "..", "include", "ddz_env.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build libddz_hip.so")


def stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + DEPS)


def _cmd(lib, extra):
    return [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall",
            "-Wno-unused-function", "-o", lib] + extra + [os.path.join(CSRC, s) for s in SOURCES]


def build(force=False, verbose=False):
    """Compile both libraries (in parallel) when stale; returns the default one."""
    jobs = []
    for lib, extra in ((LIB, []), (LIB_JK, ["-DDDZ_NATIVE_JOKER_KICKERS=1"])):
        if force or stale(lib):
            cmd = _cmd(lib, extra)
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
