"""CSR legal lists two ways, per iteration of a CHOICE-driven loop at T tables (GPU):
  (a) ddz_step (CSR, fused scan inside k_table)            -- the round-1 path
  (b) ddz_step_slab + ddz_slab_to_csr (two small launches) -- lists in the same CSR layout, same indices
Usage: python tools/csr_probe.py [T ...]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("doudizhu-rl_amd")
dev = torch.device("cuda:0")


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for T in [int(x) for x in sys.argv[1:]] or [4096, 65536]:
    env = pkg.BatchedEnv(T, seed=1, device=dev)
    env.reset(); env.rollout_random(30)
    zero = torch.zeros(T, dtype=torch.int32, device=dev)
    env.legal()
    a = timed(lambda: env.step(zero, pkg.STEP_CHOICE, auto_reset=True), 200)
    env.legal_slab()

    def b():
        env.step_slab(zero, pkg.STEP_CHOICE, auto_reset=True)
        env.slab_to_csr(rows_per_table=512)
    tb = timed(b, 200)
    tc = timed(lambda: env.slab_to_csr(rows_per_table=512), 200)
    print(f"T={T:7d}: step(CSR) {a * 1e6:7.1f} us = {T / a / 1e9:5.2f} G/s | step_slab+slab_to_csr {tb * 1e6:7.1f} us = "
          f"{T / tb / 1e9:5.2f} G/s | slab_to_csr alone {tc * 1e6:6.1f} us", flush=True)
