"""Diagnostic: time step_slab(CHOICE) with every library in build_variants/ (DDZ_HIP_LIB) and check the final state digest."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, os, sys, time, hashlib
sys.path.insert(0, %r)
import torch
pkg = importlib.import_module("doudizhu-rl_amd")
for T in (65536, 4096):
    env = pkg.BatchedEnv(T, seed=0, want_ids=False)
    env.reset(); env.rollout_random(60); env.legal_slab()
    g = torch.Generator(device="cuda").manual_seed(1)
    choice = torch.zeros(T, dtype=torch.int32, device="cuda")
    for _ in range(30):
        env.step_slab(choice, pkg.STEP_CHOICE, auto_reset=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300
    e0.record()
    for _ in range(n):
        env.step_slab(choice, pkg.STEP_CHOICE, auto_reset=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    dig = hashlib.sha1(env.state.cpu().numpy().tobytes() + env.counts.cpu().numpy().tobytes() + env.done.cpu().numpy().tobytes()).hexdigest()[:10]
    print(f"T={T}: {us:7.1f} us/iter {T / us / 1e3:6.3f} G steps/s  digest {dig} status {env.status()}", flush=True)
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, "build_variants", "*.so"))):
    print("==", os.path.basename(lib), flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, DDZ_HIP_LIB=lib), timeout=300)
