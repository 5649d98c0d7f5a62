"""Diagnostic: step_slab(CHOICE) time per iteration for different tables-per-wave settings (DDZ_TPW)."""
import importlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    pkg = importlib.import_module("doudizhu-rl_amd")
    T = int(sys.argv[2]); ids = sys.argv[3] == "1"
    env = pkg.BatchedEnv(T, seed=0, want_ids=ids)
    env.reset(); env.rollout_random(60); env.legal_slab()
    choice = torch.zeros(T, dtype=torch.int32, device="cuda")
    for _ in range(30):
        env.step_slab(choice, pkg.STEP_CHOICE, auto_reset=True)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        env.step_slab(choice, pkg.STEP_CHOICE, auto_reset=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"T={T} ids={int(ids)} DDZ_TPW={os.environ.get('DDZ_TPW','auto')}: {dt / n * 1e6:7.1f} us/iter {T * n / dt / 1e9:6.3f} G steps/s", flush=True)
else:
    for T, tpws in ((65536, ["auto", "4", "6", "8", "11", "12", "16", "22", "32"]), (4096, ["auto", "1", "2", "4"])):
        for ids in ("0", "1"):
            for tpw in tpws:
                env = dict(os.environ)
                if tpw != "auto":
                    env["DDZ_TPW"] = tpw
                subprocess.run([sys.executable, __file__, "child", str(T), ids], env=env, timeout=120)
