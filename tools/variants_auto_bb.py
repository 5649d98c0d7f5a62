"""Diagnostic: build k_auto2 variants (frontier passes / target) into build_variants/ for tools/auto_variants_run.py."""
import os, subprocess, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "build_variants")
shutil.rmtree(out, ignore_errors=True)
os.makedirs(out)
src = os.path.join(ROOT, "doudizhu-rl_amd", "csrc", "ddz_engine.hip")
for name, flags in (("base", []), ("a16_2", ["-DA2_ADAPT_N=16", "-DA2_ADAPT_P=2"]), ("a32_2", ["-DA2_ADAPT_N=32", "-DA2_ADAPT_P=2"]),
                    ("a32_3", ["-DA2_ADAPT_N=32", "-DA2_ADAPT_P=3"]), ("a64_3", ["-DA2_ADAPT_N=64", "-DA2_ADAPT_P=3"]),
                    ("a24_1", ["-DA2_ADAPT_N=24", "-DA2_ADAPT_P=1"])):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", *flags,
                           "-o", os.path.join(out, f"auto_{name}.so"), src])
    print(name, flush=True)
