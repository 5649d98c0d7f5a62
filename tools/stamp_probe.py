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
s = buf.cpu().numpy().astype(np.int64) if False else buf.cpu().numpy()
import numpy as np
names = ["prologue(entry->table)", "decode+philox", "scan(stage)", "flush rows", "pick+apply+store"]
pts = np.stack([s[:, 7], s[:, 0], s[:, 1], s[:, 2], s[:, 3], s[:, 4]], axis=1).astype(np.float64)
d = np.diff(pts, axis=1)
tot = pts[:, -1] - pts[:, 0]
print(f"T={T}: per-wave total: mean {tot.mean():.0f}  p50 {np.median(tot):.0f}  p99 {np.percentile(tot,99):.0f}  max {tot.max():.0f}")
for k, nm in enumerate(names):
    print(f"  {nm:24s} mean {d[:,k].mean():8.0f}  p50 {np.median(d[:,k]):8.0f}  p99 {np.percentile(d[:,k],99):8.0f}  max {d[:,k].max():8.0f}")
cnt = env.counts.cpu().numpy()
heavy = np.argsort(tot)[-5:]
print("slowest waves: total / scan / apply / list size:", [(int(tot[h]), int(d[h,2]), int(d[h,4]), int(cnt[h])) for h in heavy])
