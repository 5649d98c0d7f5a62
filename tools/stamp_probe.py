"""Diagnostic: build the engine with -DDDZ_STAMP, run a rollout, report where k_table's
waves spend their cycles (s_memtime deltas per phase).  Not part of the product."""
import ctypes as C
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "doudizhu-rl_amd", "csrc")
lib = os.environ.get("DDZ_HIP_LIB") or os.path.join(csrc, "libddz_hip.so")
if os.environ.get("STAMP_BUILD", "1") == "1" and not os.environ.get("DDZ_HIP_LIB"):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                           "-DDDZ_STAMP=1", "-o", lib, os.path.join(csrc, "ddz_engine.hip")])
import torch  # noqa: E402

pkg = importlib.import_module("doudizhu-rl_amd")
L = importlib.import_module("doudizhu-rl_amd._lib").lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = pkg.BatchedEnv(T, seed=0, want_ids=False)
env.reset()
env.rollout_random(100)
buf = torch.zeros((T, 16), dtype=torch.int64, device="cuda")
raw = C.CDLL(lib)
raw.ddz_debug_set_stamps.argtypes = [C.c_void_p]
assert raw.ddz_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
N_IT = int(sys.argv[2]) if len(sys.argv) > 2 else 200
env.rollout_random(N_IT)
torch.cuda.synchronize()
import numpy as np
s = buf.cpu().numpy().astype(np.float64)
names = ["prologue", "iter setup", "scan (generic)", "flush (generic)", "trajectory + loop tail", "fast path list+pick", "row updates", "carried scalars", "deal / turn change", "state store", "pick (generic)", "-"]
tot = s[:, :12].sum(1)
print(f"T={T}, {N_IT} in-launch iterations; cycles per wave per iteration (s_memtime):")
print(f"  total/iter  mean {tot.mean()/N_IT:8.0f}  max {tot.max()/N_IT:8.0f}")
for k, nm in enumerate(names):
    per = s[:, k] / (1 if k == 0 else N_IT)
    print(f"  {nm:20s} mean {per.mean():8.0f}  p99 {np.percentile(per,99):8.0f}  max {per.max():8.0f}" + ("  (once per launch)" if k == 0 else ""))
