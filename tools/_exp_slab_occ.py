"""experiment: k_slab built for 4 / 5 / 6 waves per SIMD x tables per wave, step_slab(RANDOM) and fused at 65,536 tables"""
import importlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHILD = r'''
import importlib, sys, time, torch
sys.path.insert(0, %r)
importlib.import_module("doudizhu-rl_amd._lib").use_library(sys.argv[1])
pkg = importlib.import_module("doudizhu-rl_amd")
T = 65536
for tpw in [int(x) for x in sys.argv[2].split(",")]:
    env = pkg.BatchedEnv(T, seed=0, _debug_tables_per_wave=tpw)
    env.reset(); env.rollout_random(200); env.legal_slab()
    q = torch.rand((T, env.slab_stride), device="cuda"); face = torch.empty((T, 6, 15, 4), device="cuda")
    out = []
    for name, fn in (("random", lambda: env.step_slab(None, pkg.STEP_RANDOM)), ("fused", lambda: env.policy_step_slab(q, 0.0, face_variant=3, face_out=face))):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): fn()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 300 * 1e6)
    print(f"  tpw={tpw:2d}: random {out[0]:6.1f} us  fused {out[1]:6.1f} us  status {env.status()}", flush=True)
    del env
''' % ROOT
os.makedirs(os.path.join(ROOT, "build_variants"), exist_ok=True)
src = os.path.join(ROOT, "doudizhu-rl_amd", "csrc", "ddz_engine.hip")
for wv, tpws in ((4, "16,12"), (5, "16,13,11"), (6, "16,11,8")):
    lib = os.path.join(ROOT, "build_variants", f"slab_w{wv}.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", f"-DDDZ_SLAB_WAVES={wv}", "-o", lib, src])
    print(f"k_slab built for {wv} waves per SIMD", flush=True)
    subprocess.run([sys.executable, "-c", CHILD, lib, tpws], timeout=600)
