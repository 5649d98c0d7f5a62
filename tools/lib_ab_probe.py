"""A/B of engine builds: for every shared library given (default: the product build), the random rollout (k_rollout) at 4096 and
65,536 tables and step_slab(RANDOM) (k_slab) at 65,536, one child process per library.
  python tools/lib_ab_probe.py [lib.so ...]"""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        import torch
        if sys.argv[2] != "product":
            importlib.import_module("doudizhu-rl_amd._lib").use_library(sys.argv[2])
        pkg = importlib.import_module("doudizhu-rl_amd")
        out = []
        for T in (4096, 65536):
            env = pkg.BatchedEnv(T, seed=0, want_ids=False)
            env.reset(); env.rollout_random(300)
            torch.cuda.synchronize()
            n = 4000 if T == 4096 else 1000
            best = min(env.rollout_random_timed(n) for _ in range(5))
            out.append(f"rollout {T}: {best * 1e3 / n:7.3f} us = {T * n / best / 1e6:5.2f} G/s")
            if T == 65536:
                env.legal_slab()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(20):
                    env.step_slab(None, pkg.STEP_RANDOM)
                bs = 1e9
                for _ in range(5):
                    e0.record()
                    for _ in range(100):
                        env.step_slab(None, pkg.STEP_RANDOM)
                    e1.record(); torch.cuda.synchronize()
                    bs = min(bs, e0.elapsed_time(e1) / 100)
                out.append(f"step_slab {T}: {bs * 1e3:6.2f} us")
            st = env.stats()
            out.append(f"plies {st['plies']}")
            del env
        # the stress set of bench.py (65,536 plane-rich 20-card leads) through get_moves_slab: the kicker-block rounds
        import bench
        hands = torch.from_numpy(bench.plane_rich_hands(65536, 12345)).cuda()
        lasts = torch.zeros_like(hands)
        o = pkg.get_moves_slab(hands, lasts, want_ids=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        bs = 1e9
        for _ in range(5):
            e0.record()
            for _ in range(20):
                pkg.get_moves_slab(hands, lasts, want_ids=True, out=o)
            e1.record(); torch.cuda.synchronize()
            bs = min(bs, e0.elapsed_time(e1) / 20)
        out.append(f"stress get_moves_slab: {bs * 1e3:6.1f} us (rows {int(o[0].sum())})")
        print(f"{os.path.basename(sys.argv[2]):28s} " + "; ".join(out), flush=True)
    else:
        for lib in (sys.argv[1:] or ["product"]):
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", lib if lib == "product" else os.path.abspath(lib)])
