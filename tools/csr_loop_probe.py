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

# the same loop captured in a HIP graph (two iterations per graph: the engine double-buffers its scan
# scratch by launch parity, so an even number of iterations brings the handle back to the captured state)
for T in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096", "65536"])]:
    env = pkg.BatchedEnv(T, seed=0)
    env.reset()
    choice = torch.zeros(T, dtype=torch.int32, device="cuda")
    face = torch.empty((T, 6, 15, 4), dtype=torch.float32, device="cuda")

    def loop2(n, observe):
        for _ in range(n):
            env.legal()
            if observe:
                env.observe(3, out=face)
            env.step(choice, pkg.STEP_CHOICE, auto_reset=True)

    for observe in (False, True):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            loop2(4, observe)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            loop2(2, observe)
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = env.stats()
        print(f"T={T:6d} observe={observe} hipGraph: {dt / (2 * n) * 1e6:8.1f} us/iter  {T * 2 * n / dt / 1e6:8.1f} M steps/s  "
              f"status={env.status()} plies={st['plies']}", flush=True)
    del env

# slab API: one launch per iteration (step_slab applies the choices and writes the next lists)
for T in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096", "65536"])]:
    env = pkg.BatchedEnv(T, seed=0)
    env.reset()
    env.legal_slab()
    choice = torch.zeros(T, dtype=torch.int32, device="cuda")
    face = torch.empty((T, 6, 15, 4), dtype=torch.float32, device="cuda")
    for observe in (False, True):
        for _ in range(20):
            env.step_slab(choice, pkg.STEP_CHOICE, auto_reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            if observe:
                env.observe(3, out=face)
            env.step_slab(choice, pkg.STEP_CHOICE, auto_reset=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"T={T:6d} observe={observe} slab API: {dt / n * 1e6:8.1f} us/iter  {T * n / dt / 1e6:8.1f} M steps/s  status={env.status()}", flush=True)
    del env
