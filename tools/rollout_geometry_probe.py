"""k_rollout without ids / records: 12-wave blocks (two per CU, 6 waves per SIMD; the default) against 16-wave blocks (one per
CU, 4 waves per SIMD: `_debug_tables_per_wave` = the default share restores it) at 4096 and 65,536 tables."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
pkg = importlib.import_module("doudizhu-rl_amd")
for T in (4096, 65536):
    for name, kw in (("12-wave blocks", {}), ("16-wave blocks", {"_debug_tables_per_wave": max(1, (T + 4095) // 4096)})):
        env = pkg.BatchedEnv(T, seed=0, want_ids=False, **kw)
        env.reset(); env.rollout_random(300)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            ms = env.rollout_random_timed(1000)
            best = min(best, ms)
        st = env.stats()
        print(f"T={T:6d} {name}: {best * 1e3 / 1000:7.3f} us per iteration = {T * 1000 / best / 1e6:6.2f} G steps/s  (plies {st['plies']})")
        del env
