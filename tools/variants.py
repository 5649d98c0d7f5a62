"""Kernel experiments: build patched copies of ddz_engine.hip into build_variants/<name>.so and time
them in one GPU call (`python tools/variants.py run ITERS T1,T2`); DDZ_HIP_LIB selects the library.
  build:  python tools/variants.py build variants_file.py   (file defines VARIANTS = {name: [(old, new), ...]})
Not part of the product or the tests."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "doudizhu-rl_amd", "csrc")
OUT = os.path.join(ROOT, "build_variants")


def build(vfile):
    ns = {}
    exec(open(vfile).read(), ns)
    src = open(os.path.join(CSRC, "ddz_engine.hip")).read()
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    procs = []
    for name, reps in ns["VARIANTS"].items():
        s = src
        for old, new in reps:
            assert s.count(old) >= 1, (name, old[:70])
            s = s.replace(old, new)
        d = os.path.join(OUT, name, "a", "b")  # keeps the ../../include path valid
        os.makedirs(d)
        os.makedirs(os.path.join(OUT, name, "include"), exist_ok=True)
        shutil.copy(os.path.join(ROOT, "include", "ddz_env.h"), os.path.join(OUT, name, "include"))
        for h in glob.glob(os.path.join(CSRC, "*.h")):
            shutil.copy(h, d)
        open(os.path.join(d, "ddz_engine.hip"), "w").write(s)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o",
               os.path.join(OUT, name + ".so"), "ddz_engine.hip"] + ns.get("FLAGS", {}).get(name, [])
        procs.append((name, subprocess.Popen(cmd, cwd=d, stderr=subprocess.PIPE)))
    for name, p in procs:
        _, err = p.communicate()
        print(name, "ok" if p.returncode == 0 else "FAILED\n" + err.decode()[-2000:])
    for name in ns["VARIANTS"]:
        shutil.rmtree(os.path.join(OUT, name), ignore_errors=True)


def run(iters, tables):
    libs = [None] + sorted(glob.glob(os.path.join(OUT, "*.so")))
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["DDZ_HIP_LIB"] = lib
        print("==", os.path.basename(lib) if lib else "baseline (in-tree)", flush=True)
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep.py"), iters, tables], env=env,
                       timeout=300)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2])
    else:
        run(sys.argv[2], sys.argv[3])
