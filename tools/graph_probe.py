"""hipGraph capture of the stepping loop (torch.cuda.graphs around the C-ABI launches, which go to the current stream):
K lock-step iterations of the fused policy step (arg-max over q + apply + new lists + face) replayed as ONE graph launch,
against the same calls issued one by one from Python.  Bit-identical states; the graph removes the host cost per call.
Usage: python tools/graph_probe.py [T ...]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("doudizhu-rl_amd")
dev = torch.device("cuda:0")
K = 16

for T in [int(x) for x in sys.argv[1:]] or [256, 4096, 65536]:
    a = pkg.BatchedEnv(T, seed=5, device=dev, want_ids=False)
    b = pkg.BatchedEnv(T, seed=5, device=dev, want_ids=False)
    a.reset(); b.reset(); a.legal_slab(); b.legal_slab()
    q = torch.rand((T, a.slab_stride), dtype=torch.float32, device=dev)
    fa = torch.empty((T, 6, 15, 4), dtype=torch.float32, device=dev)
    fb = torch.empty_like(fa)

    def body(env, face):
        for _ in range(K):
            env.policy_step_slab(q, 0.0, face_variant=3, face_out=face)

    body(a, fa); body(b, fb)            # warm-up, both environments in step
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            body(a, fa)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    # capture does not execute: a is K iterations behind b until the first replay
    g.replay()
    body(b, fb)
    torch.cuda.synchronize()
    assert torch.equal(a.state, b.state) and torch.equal(fa, fb) and torch.equal(a.counts, b.counts)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / (reps * K)
    t0 = time.perf_counter()
    for _ in range(reps):
        body(b, fb)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / (reps * K)
    assert torch.equal(a.state, b.state)
    print(f"T={T:6d}: eager {te * 1e6:7.1f} us/iteration = {T / te / 1e6:8.1f} M steps/s | graph of {K} iterations "
          f"{tg * 1e6:7.1f} us/iteration = {T / tg / 1e6:8.1f} M steps/s", flush=True)
