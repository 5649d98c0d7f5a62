"""Throughput of the rollout kernel vs table count (diagnostic; DDZ_TPW overrides tables/wave)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("doudizhu-rl_amd")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for T in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1024,4096,16384,65536,262144".split(","))]:
    env = pkg.BatchedEnv(T, seed=0, want_ids=False)
    env.reset()
    env.rollout_random(100)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        best = min(best, env.rollout_random_timed(iters))
    st = env.stats()
    chk = int(env.state_export().to(torch.int64).sum().item()) if hasattr(env, "state_export") else 0
    print(f"T={T:7d} tpw={os.environ.get('DDZ_TPW', 'auto'):>4s}  {best / iters * 1e3:8.3f} us/iter  "
          f"{T * iters / best / 1e3:10.1f} M steps/s  meanA={st['legal_rows'] / st['plies']:.2f} status={env.status()} chk={chk} eps={st['episodes']}",
          flush=True)
    del env
    torch.cuda.empty_cache()
