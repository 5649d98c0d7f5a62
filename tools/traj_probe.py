"""Diagnostic: rollout rate with trajectory records written (32 B per ply per table), in the default launch geometry and with
16-wave blocks forced (`_debug_tables_per_wave`)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
pkg = importlib.import_module("doudizhu-rl_amd")
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
for T in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096", "65536"])]:
    for name, kw in (("default geometry", {}), ("16-wave blocks", {"_debug_tables_per_wave": max(1, (T + 4095) // 4096)})):
        env = pkg.BatchedEnv(T, seed=0, want_ids=False, **kw)
        env.reset()
        k = min(K, (1 << 31) // (T * 32))
        traj = torch.zeros((k, T, 32), dtype=torch.uint8, device="cuda")
        env.rollout_random(50, traj=traj[:50])
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            env.rollout_random(k, traj=traj)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"T={T:7d} {name}: traj on: {best / k * 1e6:8.3f} us/iter {T * k / best / 1e6:9.1f} M steps/s  chk={int(traj[-1].to(torch.int64).sum())}", flush=True)
        del env, traj
