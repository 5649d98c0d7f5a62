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
    if T <= 4096:
        # one table per wave: wave 0 of a block decodes / applies the block's 16 tables, then every wave deals (if its game
        # ended) and writes its table's list, then the block writes its heavy lists together
        full = buf.cpu().numpy().astype(np.float64)
        w0 = full[0::16]
        w0 = w0[w0[:, 5] > 0]
        rest = np.concatenate([full[k::16] for k in range(1, 16)])
        rest = rest[rest[:, 5] > 0]
        for nm, x in (("wave 0 of a block", w0), ("waves 1..15", rest)):
            print(f"  {nm}: {len(x)} waves")
            for k, ph in ((0, "prologue"), (2, "decode + selection (16 tables)"), (3, "apply / hot fill"), (1, "wait for wave 0"),
                          (7, "deal"), (4, "own list + outputs"), (8, "plan of a rich lead"), (9, "its rounds by the wave itself"),
                          (6, "wait for the block's lists"), (10, "block lists: pass 1"), (11, "block lists: wait, scan, pass 2")):
                v = x[:, k]
                print(f"    {ph:32s} mean {v.mean():8.0f}  p50 {np.percentile(v, 50):8.0f}  p99 {np.percentile(v, 99):8.0f}  max {v.max():8.0f}   nonzero {int((v > 0).sum())}")
            tot = x[:, [0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11]].sum(1)
            print(f"    {'total':32s} mean {tot.mean():8.0f}  p50 {np.percentile(tot, 50):8.0f}  p99 {np.percentile(tot, 99):8.0f}  max {tot.max():8.0f}")
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
