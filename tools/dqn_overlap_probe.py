"""Does the Q-network loop (configs[2]) gain from running two half batches side by side on two streams -- the first layer of one
half (vector ALU) under the fc1 GEMM of the other (matrix pipes)?  One PolicyLoop of 65,536 tables against two PolicyLoops of
32,768 tables, each on its own stream, iterations issued alternately.
  python tools/dqn_overlap_probe.py [iters=20]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
pkg = importlib.import_module("doudizhu-rl_amd")
glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
net = glue.QNet(6).to(dev).eval()


def make(T, seed):
    env = pkg.BatchedEnv(T, seed=seed, device=dev)
    env.reset()
    loop = glue.PolicyLoop(env, net, face_variant=3, epsilon=0.0)
    loop.run(3)
    return env, loop


env, loop = make(65536, 0)
torch.cuda.synchronize()
t0 = time.perf_counter(); loop.run(N); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"one loop of 65,536 tables: {dt / N * 1e3:.3f} ms per iteration = {65536 * N / dt / 1e6:.1f} M env steps/s")
del env, loop
halves = [make(32768, s) for s in (1, 2)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _, lp in halves:
    lp.run(N)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"two loops of 32,768, one after the other: {dt / N * 1e3:.3f} ms per iteration pair = {65536 * N / dt / 1e6:.1f} M env steps/s")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for s in streams:
    s.wait_stream(torch.cuda.current_stream())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    for (_, lp), s in zip(halves, streams):
        with torch.cuda.stream(s):
            lp.run(1)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"two loops of 32,768 on two streams:       {dt / N * 1e3:.3f} ms per iteration pair = {65536 * N / dt / 1e6:.1f} M env steps/s")
print("status", [e.status() for e, _ in halves])
