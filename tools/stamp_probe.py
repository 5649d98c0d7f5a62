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
lib = os.path.join(csrc, "libddz_hip.so")
if os.environ.get("STAMP_BUILD", "1") == "1":
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                           "-DDDZ_STAMP=1", "-o", lib, os.path.join(csrc, "ddz_engine.hip")])
import torch  # noqa: E402

pkg = importlib.import_module("doudizhu-rl_amd")
L = importlib.import_module("doudizhu-rl_amd._lib").lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = pkg.BatchedEnv(T, seed=0, want_ids=False)
env.reset()
env.rollout_random(100)
buf = torch.zeros((T, 8), dtype=torch.int64, device="cuda")
raw = C.CDLL(lib)
raw.ddz_debug_set_stamps.argtypes = [C.c_void_p]
assert raw.ddz_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
role_before = env.role.clone()
env.rollout_random(1)
torch.cuda.synchronize()
s = buf.cpu().numpy()
names = ["load+decode", "enumerate", "step", "(reset)", "requery", "count", "scan->end"]
import numpy as np
d = np.diff(s[:, :7], axis=1).astype(np.float64)
start = s[:, 0] - s[:, 0].min()
print(f"T={T}: wave start spread: mean {start.mean():.0f} max {start.max():.0f} cycles")
tot = (s[:, 6] - s[:, 0])
print(f"per-wave total: mean {tot.mean():.0f}  p50 {np.median(tot):.0f}  p99 {np.percentile(tot,99):.0f}  max {tot.max():.0f}")
print(f"kernel span (first start -> last end): {(s[:,6].max()-s[:,0].min())}")
for k, nm in enumerate(names[:6]):
    print(f"  {nm:12s} mean {d[:,k].mean():8.0f}  p99 {np.percentile(d[:,k],99):8.0f}  max {d[:,k].max():8.0f}")
print(f"  {'scan->end':12s} mean {d[:,5+0].mean() if False else (s[:,6]-s[:,5]).mean():8.0f}  max {(s[:,6]-s[:,5]).max():8.0f}")
