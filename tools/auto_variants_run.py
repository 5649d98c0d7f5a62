"""Diagnostic: time examples/config4_rule_opponent.py against every library in build_variants/ (DDZ_HIP_LIB)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sorted(glob.glob(os.path.join(ROOT, "build_variants", "*.so"))):
    env = dict(os.environ, DDZ_HIP_LIB=lib)
    for T, it in ((65536, 40), (4096, 80)):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "config4_rule_opponent.py"), "--tables", str(T), "--iters", str(it)],
                             env=env, capture_output=True, text=True, timeout=300).stdout.strip().split("\n")[-1]
        print(os.path.basename(lib), out[:90], flush=True)
