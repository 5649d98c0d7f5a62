"""Diagnostic: cost of the CSR API loop an NN policy would drive (legal -> observe -> choice -> step)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("doudizhu-rl_amd")
for T in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096", "65536"])]:
    env = pkg.BatchedEnv(T, seed=0)
    env.reset()
    choice = torch.zeros(T, dtype=torch.int32, device="cuda")
    face = torch.empty((T, 6, 15, 4), dtype=torch.float32, device="cuda")

    def loop(n, observe):
        for _ in range(n):
            env.legal()
            if observe:
                env.observe(3, out=face)
            env.step(choice, pkg.STEP_CHOICE, auto_reset=True)   # always the first legal move

    for observe in (False, True):
        loop(20, observe)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        loop(n, observe)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"T={T:6d} observe={observe}: {dt / n * 1e6:8.1f} us/iter  {T * n / dt / 1e6:8.1f} M steps/s  status={env.status()}", flush=True)
    del env
