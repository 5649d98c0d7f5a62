"""Diagnostic: build the engine with -DDDZ_STAMP into build_variants/, run the slab API loop (step_slab(CHOICE)),
report where k_slab's waves spend their cycles (s_memtime deltas per phase).  Not product."""
import ctypes as C
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "doudizhu-rl_amd", "csrc")
out = os.path.join(ROOT, "build_variants")
os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "stamp.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DDDZ_STAMP=1",
                       "-o", lib, os.path.join(csrc, "ddz_engine.hip")])
importlib.import_module("doudizhu-rl_amd._lib").use_library(lib)
import numpy as np  # noqa: E402
import torch  # noqa: E402

pkg = importlib.import_module("doudizhu-rl_amd")
raw = C.CDLL(lib)
raw.ddz_debug_set_stamps.argtypes = [C.c_void_p]
for T in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096", "65536"])]:
    env = pkg.BatchedEnv(T, seed=0, want_ids=False)
    env.reset()
    env.rollout_random(60)
    env.legal_slab()
    RANDOM = len(sys.argv) > 2 and sys.argv[2] == "random"   # uniformly random legal moves instead of list entry 0
    choice = torch.zeros(T, dtype=torch.int32, device="cuda")
    for _ in range(10):
        if RANDOM:
            choice = (torch.rand(T, device="cuda") * env.counts).to(torch.int32)
        env.step_slab(choice, pkg.STEP_CHOICE, auto_reset=True)
    if RANDOM:
        choice = (torch.rand(T, device="cuda") * env.counts).to(torch.int32)
    buf = torch.zeros((T, 16), dtype=torch.int64, device="cuda")
    assert raw.ddz_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
    env.step_slab(choice, pkg.STEP_CHOICE, auto_reset=True)
    torch.cuda.synchronize()
    assert raw.ddz_debug_set_stamps(None) == 0
    s = buf.cpu().numpy().astype(np.float64)
    s = s[s[:, 5] > 0]
    ntab = s[:, 5]
    names = ["prologue loads + hot fill + barrier", "-", "decode + selection (per table)",
             "apply + outputs + state store (per table)", "list of the new state (per table)"]
    print(f"T={T}: {len(s)} waves, {ntab.mean():.1f} tables per wave; s_memtime cycles per wave")
    for k, nm in enumerate(names):
        per = s[:, k] / (ntab if k >= 2 else 1)
        print(f"  {nm:36s} mean {per.mean():8.0f}  p50 {np.percentile(per, 50):8.0f}  p99 {np.percentile(per, 99):8.0f}  max {per.max():8.0f}")
    tw = s[:, :5].sum(1)
    print(f"  total per wave mean {tw.mean():8.0f} cycles  p50 {np.percentile(tw, 50):8.0f}  p99 {np.percentile(tw, 99):8.0f}  max {tw.max():8.0f}")
    # the fused policy iteration (ddz_policy_step_slab): arg-max in the prologue, `face` between apply and lists
    q = torch.rand((T, env.slab_stride), dtype=torch.float32, device="cuda")
    face = torch.empty((T, 6, 15, 4), dtype=torch.float32, device="cuda")
    for _ in range(10):
        env.policy_step_slab(q, face_variant=3, face_out=face, auto_reset=True)
    buf.zero_()
    assert raw.ddz_debug_set_stamps(C.c_void_p(buf.data_ptr())) == 0
    env.policy_step_slab(q, face_variant=3, face_out=face, auto_reset=True)
    torch.cuda.synchronize()
    assert raw.ddz_debug_set_stamps(None) == 0
    s = buf.cpu().numpy().astype(np.float64)
    s = s[s[:, 5] > 0]
    ntab = s[:, 5]
    print(f"T={T} fused: per table: prologue+argmax {np.mean(s[:, 0] / ntab):.0f}  decode {np.mean(s[:, 2] / ntab):.0f}  "
          f"apply {np.mean(s[:, 3] / ntab):.0f}  face {np.mean(s[:, 6] / ntab):.0f}  lists {np.mean(s[:, 4] / ntab):.0f}  "
          f"total per wave {s[:, [0, 2, 3, 4, 6]].sum(1).mean():.0f}")
    del env
