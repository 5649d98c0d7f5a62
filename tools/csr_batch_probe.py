"""The CSR rollout (ddz_rollout_random_csr_staged) by batch size: iterations staged per rollout launch.
  python tools/csr_batch_probe.py [T=65536] [batches=2,4,8,16,32,63]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
pkg = importlib.import_module("doudizhu-rl_amd")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
batches = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,4,8,16,32,63").split(",")]
for want_ids in (False, True):
    for b in batches:
        if b * T * 512 > 0x7FFFFFFF:
            continue
        env = pkg.BatchedEnv(T, seed=0, want_ids=want_ids)
        env.reset(); env.rollout_random(100)
        n = max(b, 252 // b * b)
        env.rollout_random_csr(n, batch=b)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            env.rollout_random_csr(n, batch=b)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"T={T} ids={want_ids} batch {b:3d}: {best / n * 1e6:7.2f} us per iteration = {T * n / best / 1e9:5.2f} G steps/s "
              f"(staging {env._staging.numel() / 2**30:5.1f} GiB, status {env.status()})", flush=True)
        del env
        torch.cuda.empty_cache()
